import os, sys
sys.path.insert(0, os.getcwd())
import torch, torch.nn.functional as F
from tartangan_amd import backend
K = backend.get()
torch.manual_seed(0)
print('TG_CONV_WINO', os.environ.get('TG_CONV_WINO'))
for Cin, Cout, H, B in [(512, 512, 16, 8), (256, 256, 32, 8), (128, 256, 32, 8), (64, 64, 128, 8), (128, 128, 64, 8)]:
    x = torch.randn(B, Cin, H, H); w = torch.randn(Cout, Cin, 3, 3) * (2.0 / (9 * Cin)) ** 0.5; gy = torch.randn(B, Cout, H, H)
    want = F.conv2d(x.double(), w.double(), None, padding=1)
    y = torch.empty(B, Cout, H, H, device='cuda')
    K.conv2d_fwd(x.cuda(), w.cuda(), None, None, y, B, Cin, Cout, H, H, 3)
    e = (y.cpu().double() - want)
    wantg = F.conv_transpose2d(gy.double(), w.double(), padding=1)
    gx = torch.empty(B, Cin, H, H, device='cuda')
    K.conv2d_dgrad(gy.cuda(), w.cuda(), gx, B, Cin, Cout, H, H, 3)
    eg = (gx.cpu().double() - wantg)
    print(f'{Cin}->{Cout}@{H} fwd max {float(e.abs().max()/want.abs().max()):.2e} rms {float(e.pow(2).mean().sqrt()/want.pow(2).mean().sqrt()):.2e} | dgrad max {float(eg.abs().max()/wantg.abs().max()):.2e} rms {float(eg.pow(2).mean().sqrt()/wantg.pow(2).mean().sqrt()):.2e}')
