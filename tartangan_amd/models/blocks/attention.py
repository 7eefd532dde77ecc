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

    fuse_projections = True

    def _projections_fuse(self, x):
        plain = all(type(m) is Conv2d and m.bias is None and m.kernel_size == (1, 1) for m in (self.theta, self.phi, self.g))
        return (self.fuse_projections and plain and x.dim() == 4
                and TF.qkv_supported(x, self.theta.weight, self.phi.weight, self.g.weight))

    def forward(self, x, y=None):
        b, c, h, w = x.shape
        n = h * w
        tracked = x.requires_grad and torch.is_grad_enabled()
        if self._projections_fuse(x):
            # theta, phi, g read the same x: one pass for the three maps (and, backwards, for their joint input gradient
            # and their filter gradients) instead of three -- x has two consumers left
            xq, xr = TF.fork(x, 2) if tracked else (x, x)
            theta, phi, g = TF.qkv_projections(xq, self.theta.weight, self.phi.weight, self.g.weight)
        else:
            # four consumers: their gradients meet in one kernel, not three autograd adds
            xt, xp, xg, xr = TF.fork(x, 4) if tracked else (x, x, x, x)
            theta, phi, g = self.theta(xt), self.phi(xp), self.g(xg)
        theta = theta.view(b, c // 8, n)
        phi = TF.max_pool2(phi).view(b, c // 8, n // 4)
        g = TF.max_pool2(g).view(b, c // 2, n // 4)
        o = TF.attention_core(theta, phi, g).view(b, c // 2, h, w)          # g softmax(theta^T phi)^T
        return TF.scale_add(self.gamma, self.o(o), xr)
