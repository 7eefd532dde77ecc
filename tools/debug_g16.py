"""Dev tool: when does the G16 forward kernel go wrong?  (what ran before it / zero inputs)"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from tartangan_amd import backend
K = backend.get()
B, Cin, Cout, H = 64, int(os.environ.get('CIN', 128)), int(os.environ.get('COUT', 128)), int(os.environ.get('HW', 16))
g = torch.Generator(device='cuda').manual_seed(Cin * 1000 + Cout * 10 + H)
x = torch.randn(B, Cin, H, H, device='cuda', generator=g); gy = torch.randn(B, Cout, H, H, device='cuda', generator=g)
w = torch.randn(Cout, Cin, 3, 3, device='cuda', generator=g); bias = torch.randn(Cout, device='cuda', generator=g)
want0 = torch.nn.functional.conv2d(x.cpu(), w.cpu(), bias.cpu(), padding=1).cuda()
def check(y, want, tag):
    bad = ((y - want).abs() > 1e-3 * want0.abs().max()).nonzero()
    msg = f'{tag}: {len(bad)} bad'
    if len(bad):
        per_img = torch.bincount(bad[:, 0], minlength=B)
        per_co = torch.bincount(bad[:, 1], minlength=Cout)
        per_row = torch.bincount(bad[:, 2], minlength=H)
        per_col = torch.bincount(bad[:, 3], minlength=H)
        msg += f'; images with bad {int((per_img > 0).sum())}; per co (first 40) {per_co.tolist()[:40]}; per row {per_row.tolist()}; per col {per_col.tolist()}; max |y| {y.abs().max().item():.3g}'
    print(msg, flush=True)
def fwd(xx, ww, tag, want):
    y = torch.full((B, Cout, H, H), 7.0, device='cuda')
    K.conv2d_fwd(xx, ww, bias, None, y, B, Cin, Cout, H, H, 3); torch.cuda.synchronize()
    check(y, want, tag)
fwd(x, w, 'first thing in the process', want0)
gx = torch.empty_like(x)
K.conv2d_dgrad(gy, w, gx, B, Cin, Cout, H, H, 3); torch.cuda.synchronize()
fwd(x, w, 'after a dgrad', want0)
gw = torch.empty_like(w); gb = torch.empty_like(bias)
ws = torch.empty(K.conv2d_wgrad_workspace(B, Cin, Cout, H, H, 3) // 4 + 4, device='cuda')
K.conv2d_wgrad(x, gy, gw, gb, ws, ws.numel() * 4, B, Cin, Cout, H, H, 3, 0); torch.cuda.synchronize()
fwd(x, w, 'after a wgrad', want0)
zb = bias[None, :, None, None].expand(B, Cout, H, H)
fwd(torch.zeros_like(x), w, 'after a wgrad, x = 0', zb)
fwd(x, torch.zeros_like(w), 'after a wgrad, w = 0', zb)
big = torch.full((1 << 24,), 3.0, device='cuda'); big2 = big * 2 + 1; torch.cuda.synchronize()
fwd(x, w, 'after an elementwise torch kernel', want0)
