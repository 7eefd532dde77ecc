"""SelfAttention2d (reference models/blocks/attention.py:6-35) on HIP kernels.

theta/phi/g/o are bias-free 1x1 convolutions, phi and g are 2x2 max-pooled, the
(N x N/4) attention map is a row softmax of theta^T phi and the block returns
gamma * o(g beta^T) + x with a learnable scalar gamma initialised to 0.

The attention core runs in fused kernels that never materialise the (N x N/4) map: forward,
first-order backward and -- under the discriminator's R1 penalty, where the backward is itself
differentiated -- the hand-derived derivative of that backward (functional._AttnBwd).
"""
import torch
from torch import nn

from ... import functional as TF
from ..layers import Conv2d


class SelfAttention2d(nn.Module):
    def __init__(self, in_dims, attention_dims=None):
        super().__init__()
        self.in_dims = in_dims
        # registration order theta, phi, g, o, gamma fixes both the init RNG order and the state_dict keys
        self.theta = Conv2d(in_dims, in_dims // 8, 1, bias=False)
        self.phi = Conv2d(in_dims, in_dims // 8, 1, bias=False)
        self.g = Conv2d(in_dims, in_dims // 2, 1, bias=False)
        self.o = Conv2d(in_dims // 2, in_dims, 1, bias=False)
        self.gamma = nn.Parameter(torch.tensor(0.), requires_grad=True)

    def forward(self, x, y=None):
        b, c, h, w = x.shape
        n = h * w
        if x.requires_grad and torch.is_grad_enabled():
            xt, xp, xg, xr = TF.fork(x, 4)       # four consumers: their gradients meet in one kernel, not three autograd adds
        else:
            xt = xp = xg = xr = x
        theta = self.theta(xt).view(b, c // 8, n)
        phi = TF.max_pool2(self.phi(xp)).view(b, c // 8, n // 4)
        g = TF.max_pool2(self.g(xg)).view(b, c // 2, n // 4)
        o = TF.attention_core(theta, phi, g).view(b, c // 2, h, w)          # g softmax(theta^T phi)^T
        return TF.scale_add(self.gamma, self.o(o), xr)
