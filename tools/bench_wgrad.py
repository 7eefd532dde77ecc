"""Dev tool (GPU box): isolated timing of the 3x3 weight-gradient layers of the 128:3 step (partials kernel), at the batch
sizes the step runs them (64: generator / R1 sweep; 128: the paired discriminator pass)."""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from tartangan_amd import backend
K = backend.get()
SHAPES = [  # B, Cin, Cout, H
    (64, 128, 128, 8), (64, 128, 128, 16), (64, 64, 64, 32), (64, 32, 32, 64), (64, 16, 16, 128),
    (128, 4, 16, 128), (128, 16, 32, 64), (128, 32, 64, 32), (128, 64, 128, 16), (128, 128, 128, 8),
    (64, 16, 32, 64), (64, 32, 64, 32), (64, 64, 128, 16),
]
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
tot = 0.0
for B, Cin, Cout, H in SHAPES:
    x = torch.randn(B, Cin, H, H, device='cuda'); gy = torch.randn(B, Cout, H, H, device='cuda')
    ws = torch.empty(K.conv2d_wgrad_workspace(B, Cin, Cout, H, H, 3) // 4 + 4, device='cuda')
    fl = 2.0 * B * Cin * Cout * H * H * 9
    t = timeit(lambda: K.conv2d_wgrad_partials(x, gy, ws, ws.numel() * 4, B, Cin, Cout, H, H, 3, 1))
    tot += t
    print(f'B{B:4d} {Cin:4d}->{Cout:4d} @{H:3d}^2   {fl/1e9:7.2f} GFLOP  partials {t*1e3:7.1f} us  {fl/(t*1e-3)/1e12:5.1f} TF', flush=True)
print(f'sum {tot*1e3:.1f} us')
