"""Dev tool (GPU box): the stride-1 3x3 layers of the 128:3 step, forward and dgrad -- max error against a float64 reference
(small batch) and isolated time at the step's batches.  Run once per TG_CONV_WINO value (the knob is read once per process)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import torch.nn.functional as F
from tartangan_amd import backend
K = backend.get()
SHAPES = [(128, 128, 16), (64, 128, 16), (64, 64, 32), (32, 64, 32), (32, 32, 64), (16, 32, 64), (16, 16, 128), (128, 128, 8), (128, 128, 4)]
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
print('TG_CONV_WINO =', os.environ.get('TG_CONV_WINO', '(default)'))
torch.manual_seed(0)
for Cin, Cout, H in SHAPES:
    Bs = 5
    x = torch.randn(Bs, Cin, H, H); w = torch.randn(Cout, Cin, 3, 3) * 0.2; b = torch.randn(Cout); r = torch.randn(Bs, Cout, H, H)
    rlo = torch.randn(Bs, Cout, H // 2, H // 2); gy = torch.randn(Bs, Cout, H, H)
    want = F.conv2d(x.double(), w.double(), b.double(), padding=1) + r.double()
    y = torch.empty(Bs, Cout, H, H, device='cuda')
    K.conv2d_fwd(x.cuda(), w.cuda(), b.cuda(), r.cuda(), y, Bs, Cin, Cout, H, H, 3)
    e1 = float((y.cpu().double() - want).abs().max() / want.abs().max())
    want = F.conv2d(x.double(), w.double(), None, padding=1) + F.interpolate(rlo.double(), scale_factor=2, mode='nearest')
    K.conv2d_fwd_up2res(x.cuda(), w.cuda(), None, rlo.cuda(), y, Bs, Cin, Cout, H, H)
    e2 = float((y.cpu().double() - want).abs().max() / want.abs().max())
    want = F.conv_transpose2d(gy.double(), w.double(), padding=1)
    gx = torch.empty(Bs, Cin, H, H, device='cuda')
    K.conv2d_dgrad(gy.cuda(), w.cuda(), gx, Bs, Cin, Cout, H, H, 3)
    e3 = float((gx.cpu().double() - want).abs().max() / want.abs().max())
    line = f'{Cin:4d}->{Cout:4d} @{H:3d}^2  err fwd {e1:.1e} up2res {e2:.1e} dgrad {e3:.1e} |'
    for B in (64, 128):
        x = torch.randn(B, Cin, H, H, device='cuda'); gy = torch.randn(B, Cout, H, H, device='cuda')
        w_ = w.cuda(); b_ = b.cuda(); y = torch.empty(B, Cout, H, H, device='cuda'); gx = torch.empty_like(x)
        fl = 2.0 * B * Cin * Cout * H * H * 9
        t1 = timeit(lambda: K.conv2d_fwd(x, w_, b_, None, y, B, Cin, Cout, H, H, 3))
        t2 = timeit(lambda: K.conv2d_dgrad(gy, w_, gx, B, Cin, Cout, H, H, 3))
        line += f'  B{B}: fwd {t1*1e3:6.1f} us ({fl/t1/1e9:5.1f} TF) dgrad {t2*1e3:6.1f} us ({fl/t2/1e9:5.1f} TF)'
    print(line, flush=True)
