"""Dev tool (GPU box): isolated timing of the fused attention kernels at the 128:3 shapes."""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from tartangan_amd import backend
K = backend.get()

def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3

for (B, D, DV, N, M) in [(64, 4, 16, 4096, 1024), (64, 4, 16, 1024, 256), (64, 16, 64, 256, 64), (16, 4, 16, 4096, 1024), (64, 8, 32, 1024, 256)]:
    th, ph, g = torch.randn(B, D, N, device='cuda'), torch.randn(B, D, M, device='cuda'), torch.randn(B, DV, M, device='cuda')
    o, lse = torch.empty(B, DV, N, device='cuda'), torch.empty(B, N, device='cuda')
    go = torch.randn(B, DV, N, device='cuda')
    dth, dph, dg = torch.empty_like(th), torch.empty_like(ph), torch.empty_like(g)
    ws = torch.empty(K.attn_bwd_workspace(B, D, DV, N, M) // 4 + 4, device='cuda')
    t1 = timeit(lambda: K.attn_fwd(th, ph, g, o, lse, B, D, DV, N, M))
    t2 = timeit(lambda: K.attn_bwd(go, th, ph, g, o, lse, dth, dph, dg, ws, B, D, DV, N, M))
    pairs = B * N * M
    print(f'B={B} D={D} DV={DV} N={N} M={M}: fwd {t1:7.1f} us ({2*pairs*(D+DV)/t1/1e6:6.2f} TF)   bwd {t2:7.1f} us ({2*pairs*(2*D+2*DV+D+DV)/t2/1e6:6.2f} TF)')

# second-order kernel (the discriminator's map under R1)
for (B, D, DV, N, M) in [(64, 4, 16, 1024, 256), (64, 8, 32, 256, 64)]:
    if not K.attn_dbwd_supported(D, DV, M):
        continue
    r = lambda *s: torch.randn(*s, device='cuda')
    go, th, ph, g, a, b, c = r(B, DV, N), r(B, D, N), r(B, D, M), r(B, DV, M), r(B, D, N), r(B, D, M), r(B, DV, M)
    lse = torch.logsumexp(torch.bmm(th.transpose(1, 2), ph), -1)
    outs = [torch.empty_like(t) for t in (go, th, ph, g)]
    ws = torch.empty(K.attn_dbwd_workspace(B, D, DV, N, M) // 4 + 4, device='cuda')
    t = timeit(lambda: K.attn_dbwd(go, th, ph, g, lse, a, b, c, *outs, ws, B, D, DV, N, M))
    print(f'B={B} D={D} DV={DV} N={N} M={M}: dbwd {t:7.1f} us ({2*B*N*M*5.4*(D+DV)/t/1e6:6.2f} TF vector)')
