"""Dev tool / test helper (GPU box): the 3x3 and stride-2 kernels under the CURRENT environment's TG_* knobs against plain torch on the
CPU, at the benched batch (64), with NaN-poisoned LDS in front of every launch.  Exit code 0 = every output within tolerance."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import torch.nn.functional as F
from tartangan_amd import backend
K = backend.get()
B = 64
SHAPES = [(16, 16, 128), (32, 32, 64), (64, 64, 32), (128, 128, 16), (128, 128, 8)]
bad = 0


def poison():
    nan = torch.full((64, 32, 64, 64), float('nan'), device='cuda')
    w = torch.full((32, 32, 3, 3), float('nan'), device='cuda')
    K.conv2d_fwd(nan, w, None, None, torch.empty_like(nan), 64, 32, 32, 64, 64, 3)
    ws = torch.empty(K.conv2d_wgrad_workspace(64, 32, 32, 64, 64, 3) // 4 + 4, device='cuda')
    K.conv2d_wgrad(nan, nan, torch.empty_like(w), None, ws, ws.numel() * 4, 64, 32, 32, 64, 64, 3, 0)


def check(name, got, want, tol=1e-4):
    global bad
    err = float((got.cpu() - want).abs().max())
    lim = tol * max(float(want.abs().max()), 1e-6)
    ok = err <= lim and bool(torch.isfinite(got).all())
    if not ok:
        bad += 1
    print(f'  {name}: max err {err:.3e} (limit {lim:.3e}) {"ok" if ok else "FAIL"}', flush=True)


for Cin, Cout, H in SHAPES:
    g = torch.Generator().manual_seed(Cin * 1000 + Cout * 10 + H)
    x = torch.randn(B, Cin, H, H, generator=g); gy = torch.randn(B, Cout, H, H, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.2; b = torch.randn(Cout, generator=g)
    r = torch.randn(B, Cout, H // 2, H // 2, generator=g)
    xd, gyd, wd, bd, rd = x.cuda(), gy.cuda(), w.cuda(), b.cuda(), r.cuda()
    print(f'{Cin}->{Cout} @{H}^2')
    y = torch.empty(B, Cout, H, H, device='cuda')
    poison(); K.conv2d_fwd(xd, wd, bd, None, y, B, Cin, Cout, H, H, 3)
    want = F.conv2d(x, w, b, padding=1)
    check('fwd', y, want)
    poison(); K.conv2d_fwd_up2res(xd, wd, bd, rd, y, B, Cin, Cout, H, H)
    check('fwd + up2x(residual)', y, want + F.interpolate(r, scale_factor=2))
    gx = torch.empty(B, Cin, H, H, device='cuda')
    poison(); K.conv2d_dgrad(gyd, wd, gx, B, Cin, Cout, H, H, 3)
    check('dgrad', gx, F.conv_transpose2d(gy, w, padding=1))
    gw = torch.empty_like(wd); gb = torch.empty_like(bd)
    ws = torch.empty(K.conv2d_wgrad_workspace(B, Cin, Cout, H, H, 3) // 4 + 4, device='cuda')
    poison(); K.conv2d_wgrad(xd, gyd, gw, gb, ws, ws.numel() * 4, B, Cin, Cout, H, H, 3, 0)
    check('wgrad', gw, torch.nn.grad.conv2d_weight(x, w.shape, gy, padding=1), tol=2e-4)
    check('bgrad', gb, gy.sum((0, 2, 3)), tol=2e-4)
    if H >= 16:                                             # stride-2 forms: H is the HIGH resolution here
        Hl = H // 2
        a = torch.randn(B, Cin, Hl, Hl, generator=g)
        wp = torch.empty(4, Cout, Cin, 2, 2, device='cuda')
        K.upconv3x3_weights(wd, wp, Cout, Cin)
        poison(); K.upconv3x3_fwd(a.cuda(), wp, bd, None, y, B, Cin, Cout, Hl, Hl)
        check('upconv fwd', y, F.conv2d(F.interpolate(a, scale_factor=2), w, b, padding=1))
        if K.poolconv3x3_supported(B, Cin, Cout, Hl, Hl):
            w4 = torch.empty(Cout, Cin, 4, 4, device='cuda'); wpp = torch.empty(4, Cin, Cout, 2, 2, device='cuda')
            K.poolconv3x3_weights(wd, w4, wpp, Cout, Cin)
            yl = torch.empty(B, Cout, Hl, Hl, device='cuda')
            poison(); K.poolconv3x3_fwd(xd, w4, bd, None, yl, B, Cin, Cout, Hl, Hl)
            check('poolconv fwd', yl, F.avg_pool2d(want, 2))
print('knob check:', 'FAILED' if bad else 'ok', {k: v for k, v in os.environ.items() if k.startswith('TG_')})
sys.exit(1 if bad else 0)
