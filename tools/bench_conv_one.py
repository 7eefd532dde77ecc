"""Dev tool: run one conv shape N times (for rocprofv3 --pmc)."""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from tartangan_amd import backend
K = backend.get()
B = 64
for spec in sys.argv[1:]:
    Cin, Cout, H, ks = map(int, spec.split(','))
    x = torch.randn(B, Cin, H, H, device='cuda'); gy = torch.randn(B, Cout, H, H, device='cuda')
    w = torch.randn(Cout, Cin, ks, ks, device='cuda'); bias = torch.randn(Cout, device='cuda')
    y = torch.empty(B, Cout, H, H, device='cuda'); gw = torch.empty_like(w); gb = torch.empty_like(bias)
    ws = torch.empty(K.conv2d_wgrad_workspace(B, Cin, Cout, H, H, ks) // 4 + 4, device='cuda')
    gx = torch.empty_like(x)
    for _ in range(5):
        K.conv2d_fwd(x, w, bias, None, y, B, Cin, Cout, H, H, ks)
        K.conv2d_dgrad(gy, w, gx, B, Cin, Cout, H, H, ks)
        K.conv2d_wgrad(x, gy, gw, gb, ws, ws.numel() * 4, B, Cin, Cout, H, H, ks, 0)
    torch.cuda.synchronize()
