"""Dev tool (GPU box): the LDS-DMA conv kernel against the register-staged one.
  TG_CONV_DMA=0 python tools/dma_check.py save   -> reference outputs + timings of the old kernel
  [TG_DMA_TILE=..] [TG_DMA_CK=..] python tools/dma_check.py check  -> bitwise / max-abs comparison + timings
"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from tartangan_amd import backend
K = backend.get()
mode = sys.argv[1]
B = int(os.environ.get('B', 64))
SHAPES = [(int(v) for v in os.environ['SHAPE'].split(','))] if os.environ.get('SHAPE') else [(16, 16, 128), (32, 16, 128), (4, 16, 128), (32, 32, 64), (16, 32, 64), (64, 32, 64), (64, 64, 32), (32, 64, 32), (128, 64, 32),
          (128, 128, 16), (64, 128, 16), (20, 24, 32), (16, 16, 16), (128, 128, 8), (64, 128, 8), (128, 128, 4), (128, 64, 16), (256, 256, 8)]
path = os.path.join(REPO, 'gpurun_out', 'dma_ref.pt')
ref = torch.load(path) if mode == 'check' else {}
def timeit(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
tot = 0
for Cin, Cout, H in SHAPES:
    g = torch.Generator(device='cuda').manual_seed(Cin * 1000 + Cout * 10 + H)
    x = torch.randn(B, Cin, H, H, device='cuda', generator=g); gy = torch.randn(B, Cout, H, H, device='cuda', generator=g)
    w = torch.randn(Cout, Cin, 3, 3, device='cuda', generator=g); bias = torch.randn(Cout, device='cuda', generator=g)
    res = torch.randn(B, Cout, H, H, device='cuda', generator=g)
    y = torch.empty(B, Cout, H, H, device='cuda'); y2 = torch.empty_like(y); gx = torch.empty_like(x)
    K.conv2d_fwd(x, w, bias, None, y, B, Cin, Cout, H, H, 3)
    K.conv2d_fwd(x, w, None, res, y2, B, Cin, Cout, H, H, 3)
    K.conv2d_dgrad(gy, w, gx, B, Cin, Cout, H, H, 3)
    gw = torch.empty_like(w); gb = torch.empty_like(bias)
    ws = torch.empty(K.conv2d_wgrad_workspace(B, Cin, Cout, H, H, 3) // 4 + 4, device='cuda')
    K.conv2d_wgrad(x, gy, gw, gb, ws, ws.numel() * 4, B, Cin, Cout, H, H, 3, 0)
    torch.cuda.synchronize()
    key = f'{Cin}-{Cout}-{H}'
    fl = 2.0 * B * Cin * Cout * H * H * 9
    t1 = timeit(lambda: K.conv2d_fwd(x, w, bias, None, y, B, Cin, Cout, H, H, 3))
    t2 = timeit(lambda: K.conv2d_dgrad(gy, w, gx, B, Cin, Cout, H, H, 3))
    t3 = timeit(lambda: K.conv2d_wgrad(x, gy, gw, gb, ws, ws.numel() * 4, B, Cin, Cout, H, H, 3, 0))
    tot += t1 + t2 + t3
    msg = ''
    if os.environ.get('CPUREF'):
        want = torch.nn.functional.conv2d(x.cpu(), w.cpu(), bias.cpu(), padding=1).cuda()
        nb = int(((y - want).abs() > 1e-3 * want.abs().max()).sum())
        msg += f' [fwd vs CPU conv: {nb} bad]'
        y3 = torch.empty_like(y)
        K.conv2d_fwd(x, w, bias, None, y3, B, Cin, Cout, H, H, 3); torch.cuda.synchronize()
        nb = int(((y3 - want).abs() > 1e-3 * want.abs().max()).sum())
        msg += f' [fresh launch vs CPU conv: {nb} bad]'
    if mode == 'save':
        ref[key] = (y.cpu(), y2.cpu(), gx.cpu(), gw.cpu(), gb.cpu())
    else:
        for name, got, want in zip(('fwd', 'fwd+res', 'dgrad', 'wgrad', 'bgrad'), (y, y2, gx, gw, gb), ref[key]):
            want = want.cuda()
            d = (got - want).abs().max().item()
            if d > 1e-2 * want.abs().max().item() and got.dim() == 4:
                bad = ((got - want).abs() > 1e-3 * want.abs().max()).nonzero()
                print(f'   {name}: {len(bad)} bad of {got.numel()}; images {bad[:, 0].unique().numel()} channels {bad[:, 1].unique().tolist()[:40]} '
                      f'rows {bad[:, 2].unique().tolist()} cols {bad[:, 3].unique().tolist()}')
                b0 = bad[0].tolist()
                print('   first bad', b0, 'got', got[tuple(b0)].item(), 'want', want[tuple(b0)].item())
            msg += f' {name}: {"bit-equal" if torch.equal(got, want) else f"max|d| {d:.2e} (ref max {want.abs().max().item():.1f})"};'
    print(f'{Cin:4d}->{Cout:4d} @{H:3d}^2  fwd {t1:7.1f} us ({fl / t1 / 1e6:5.1f} TF)  dgrad {t2:7.1f} us ({fl / t2 / 1e6:5.1f} TF)  wgrad {t3:7.1f} us ({fl / t3 / 1e6:5.1f} TF) {msg}', flush=True)
print(f'sum {tot:.1f} us')
if mode == 'save':
    torch.save(ref, path)
