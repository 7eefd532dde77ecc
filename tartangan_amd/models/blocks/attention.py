"""SelfAttention2d (reference models/blocks/attention.py:6-35) on HIP kernels.

theta/phi/g/o are bias-free 1x1 convolutions, phi and g are 2x2 max-pooled, the
(N x N/4) attention map is a row softmax of theta^T phi and the block returns
gamma * o(g beta^T) + x with a learnable scalar gamma initialised to 0.

This composition is differentiable twice (every primitive is a transpose pair or
has an explicit second-order kernel), which the discriminator's R1 penalty needs.
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
        theta = self.theta(x).view(b, c // 8, n)
        phi = TF.max_pool2(self.phi(x)).view(b, c // 8, n // 4)
        g = TF.max_pool2(self.g(x)).view(b, c // 2, n // 4)
        beta = TF.softmax_lastdim(TF.matmul(theta, phi, transA=True))       # (b, n, n/4)
        o = TF.matmul(g, beta, transB=True).view(b, c // 2, h, w)           # g beta^T
        return TF.scale_add(self.gamma, self.o(o), x)
