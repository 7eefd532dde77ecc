"""Dev tool (GPU box): isolated timing of the stride-2 family at the 128:3 discriminator / generator shapes."""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from tartangan_amd import backend
K = backend.get()
B = 64
ITERS = int(os.environ.get('ITERS', 20))

def timeit(fn, iters=ITERS):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3

only = sys.argv[1] if len(sys.argv) > 1 else ''
for (Cin, Cout, H) in [(16, 16, 64), (32, 32, 32), (64, 64, 16), (128, 128, 8)]:       # pooled conv: x (Cin, 2H) -> y (Cout, H)
    x = torch.randn(B, Cin, 2 * H, 2 * H, device='cuda'); gy = torch.randn(B, Cout, H, H, device='cuda')
    w = torch.randn(Cout, Cin, 3, 3, device='cuda'); bias = torch.randn(Cout, device='cuda')
    w4 = torch.empty(Cout, Cin, 4, 4, device='cuda'); wp = torch.empty(4, Cin, Cout, 2, 2, device='cuda')
    K.poolconv3x3_weights(w, w4, wp, Cout, Cin)
    y = torch.empty(B, Cout, H, H, device='cuda'); gx = torch.empty_like(x); gw = torch.empty_like(w)
    ws = torch.empty(K.poolconv3x3_wgrad_workspace(B, Cin, Cout, H, H) // 4 + 4, device='cuda')
    fl = 2.0 * B * Cin * Cout * H * H * 16
    t1 = timeit(lambda: K.poolconv3x3_fwd(x, w4, bias, None, y, B, Cin, Cout, H, H))
    if only == 'fwd': continue
    t2 = timeit(lambda: K.poolconv3x3_dgrad(gy, wp, gx, B, Cin, Cout, H, H))
    t3 = timeit(lambda: K.poolconv3x3_wgrad(x, gy, gw, ws, ws.numel() * 4, B, Cin, Cout, H, H, 0, None))
    tf = lambda t: fl / t / 1e6
    print(f'pool {Cin:3d}->{Cout:3d} out {H:3d}^2  {fl/1e9:5.2f} GF  fwd {t1:6.1f} us ({tf(t1):5.1f} TF)  dgrad {t2:6.1f} ({tf(t2):5.1f})  wgrad {t3:6.1f} ({tf(t3):5.1f})')
for (Cin, Cout, H) in [(128, 128, 8), (128, 64, 16), (64, 32, 32), (32, 16, 64)]:       # up-conv: a (Cin, H) -> y (Cout, 2H)
    a = torch.randn(B, Cin, H, H, device='cuda'); gy = torch.randn(B, Cout, 2 * H, 2 * H, device='cuda')
    w = torch.randn(Cout, Cin, 3, 3, device='cuda'); bias = torch.randn(Cout, device='cuda')
    wp = torch.empty(4, Cout, Cin, 2, 2, device='cuda'); w4t = torch.empty(Cin, Cout, 4, 4, device='cuda')
    K.upconv3x3_weights(w, wp, Cout, Cin); K.upconv3x3_weights_t(w, w4t, Cout, Cin)
    y = torch.empty(B, Cout, 2 * H, 2 * H, device='cuda'); ga = torch.empty_like(a)
    fl = 2.0 * B * Cin * Cout * H * H * 16
    t1 = timeit(lambda: K.upconv3x3_fwd(a, wp, bias, None, y, B, Cin, Cout, H, H))
    t2 = timeit(lambda: K.upconv3x3_dgrad(gy, w4t, ga, B, Cin, Cout, H, H)) if K.upconv3x3_dgrad_supported(B, Cin, Cout, H, H) else float('nan')
    tf = lambda t: fl / t / 1e6
    print(f'up   {Cin:3d}->{Cout:3d} in  {H:3d}^2  {fl/1e9:5.2f} GF  fwd {t1:6.1f} us ({tf(t1):5.1f} TF)  dgrad {t2:6.1f} ({tf(t2):5.1f})')
