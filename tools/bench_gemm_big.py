"""Dev tool (GPU box): tg_gemm_big at the Newton-Schulz shape (2048^3 fp32) -- time and TFLOP/s."""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from tartangan_amd import backend
K = backend.get()
for (M, N, Kd, ta) in [(2048, 2048, 2048, 0), (2048, 2048, 1000, 1), (2048, 2048, 10000, 1), (4096, 4096, 4096, 0), (1024, 1024, 1024, 0)]:
    A = torch.randn((Kd, M) if ta else (M, Kd), device='cuda'); B = torch.randn(Kd, N, device='cuda'); C = torch.empty(M, N, device='cuda')
    f = lambda: K.gemm_big(A, B, C, M, N, Kd, A.shape[1], N, N, ta, 1.0, 0.0)
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): f()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    print(f'M{M} N{N} K{Kd} ta{ta}: {ms*1e3:8.1f} us  {2.0*M*N*Kd/ms/1e9:6.1f} TFLOP/s  ({2.0*M*N*Kd/ms/1e9/157.3*100:.0f}% of 157.3)', flush=True)
import time
from tartangan_amd import inception_utils as IU
S = torch.randn(2048, 2048, device='cuda'); S = (S @ S.t()) / 2048
torch.cuda.synchronize(); t0 = time.perf_counter()
r = IU.sqrt_newton_schulz(S.unsqueeze(0), 20); torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f'sqrt_newton_schulz 2048, 20 iters: {dt*1e3:.2f} ms  ({60*2*2048**3/dt/1e12:.1f} TFLOP/s incl. host)')
