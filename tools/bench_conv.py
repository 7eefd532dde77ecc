"""Dev tool (GPU box): isolated timing of every conv shape of the 128:3 step (fwd / dgrad / wgrad)."""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from tartangan_amd import backend
K = backend.get()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
SHAPES = [  # Cin, Cout, H, ks
    (128, 128, 8, 3), (128, 128, 16, 3), (128, 64, 32, 3), (64, 64, 32, 3), (64, 32, 64, 3), (32, 32, 64, 3),
    (32, 16, 128, 3), (16, 16, 128, 3), (16, 32, 64, 3), (32, 64, 32, 3), (64, 128, 16, 3), (128, 128, 4, 3), (4, 16, 128, 3),
    (3, 16, 128, 1), (16, 3, 128, 1), (32, 16, 128, 1), (64, 32, 64, 1), (128, 64, 32, 1), (16, 32, 64, 1), (32, 4, 64, 1), (32, 16, 64, 1), (16, 32, 64, 1),
]
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
print(f'B={B}   shape                 GFLOP   fwd us (TF)     dgrad us (TF)    wgrad us (TF)')
tot = [0, 0, 0]
for Cin, Cout, H, ks in SHAPES:
    x = torch.randn(B, Cin, H, H, device='cuda'); gy = torch.randn(B, Cout, H, H, device='cuda')
    w = torch.randn(Cout, Cin, ks, ks, device='cuda'); bias = torch.randn(Cout, device='cuda')
    y = torch.empty(B, Cout, H, H, device='cuda'); gx = torch.empty_like(x); gw = torch.empty_like(w); gb = torch.empty_like(bias)
    ws = torch.empty(K.conv2d_wgrad_workspace(B, Cin, Cout, H, H, ks) // 4 + 4, device='cuda')
    fl = 2.0 * B * Cin * Cout * H * H * ks * ks
    t1 = timeit(lambda: K.conv2d_fwd(x, w, bias, None, y, B, Cin, Cout, H, H, ks))
    t2 = timeit(lambda: K.conv2d_dgrad(gy, w, gx, B, Cin, Cout, H, H, ks))
    t3 = timeit(lambda: K.conv2d_wgrad(x, gy, gw, gb, ws, ws.numel() * 4, B, Cin, Cout, H, H, ks, 0))
    tf = lambda t: fl / (t * 1e-3) / 1e12
    print(f'{Cin:4d}->{Cout:4d} @{H:3d}^2 k{ks}   {fl/1e9:8.2f}   {t1*1e3:7.1f} ({tf(t1):5.1f})   {t2*1e3:7.1f} ({tf(t2):5.1f})   {t3*1e3:7.1f} ({tf(t3):5.1f})')
