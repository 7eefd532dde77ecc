import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch, numpy as np
from tartangan_amd import backend
K = backend.get()
torch.manual_seed(0)
x = torch.randn(1, 128, 128)
y = torch.zeros(1, 64, 64).cuda()
K.bilinear_half_fwd(x.cuda(), y, 1, 128, 128)
ref = torch.nn.functional.interpolate(x[None], scale_factor=0.5, mode='bilinear', align_corners=True)[0]
d = (y.cpu() - ref)[0].abs()
print('max', float(d.max()))
print('per-row max:', [f'{float(v):.1e}' for v in d.max(1).values[:64:4]])
print('per-col max:', [f'{float(v):.1e}' for v in d.max(0).values[:64:4]])
# probe lambdas through a linear ramp image: x[h][w] = h  -> y[oh][ow] = l0*i0 + l1*i1 = r (interpolated row coordinate)
ramp = torch.arange(128).float().view(1, 128, 1).expand(1, 128, 128).contiguous()
yr = torch.zeros(1, 64, 64).cuda(); K.bilinear_half_fwd(ramp.cuda(), yr, 1, 128, 128)
refr = torch.nn.functional.interpolate(ramp[None], scale_factor=0.5, mode='bilinear', align_corners=True)[0]
print('ramp rows hip :', [f'{float(v):.6f}' for v in yr.cpu()[0, 58:64, 0]])
print('ramp rows torch:', [f'{float(v):.6f}' for v in refr[0, 58:64, 0]])
print('ramp rows exact:', [f'{127/63*o:.6f}' for o in range(58, 64)])
for o, i0 in ((59, 118), (61, 122), (37, 74), (5, 10)):
    r2 = (torch.arange(128).float() - i0).view(1, 128, 1).expand(1, 128, 128).contiguous()
    y2 = torch.zeros(1, 64, 64).cuda(); K.bilinear_half_fwd(r2.cuda(), y2, 1, 128, 128)
    t2 = torch.nn.functional.interpolate(r2[None], scale_factor=0.5, mode='bilinear', align_corners=True)[0]
    print(f'lambda(o={o}): hip {float(y2[0, o, 0]):.9f} torch {float(t2[0, o, 0]):.9f} exact {127/63*o - i0:.9f}')
