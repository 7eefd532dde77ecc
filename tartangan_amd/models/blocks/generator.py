"""Generator blocks of the reference (models/blocks/generator.py) on HIP kernels.

Only the classes trainers.cnn / trainers.iqn instantiate are implemented:
``ResidualGeneratorBlock`` (generator.py:32-62), ``GeneratorInputMLP`` (:65-80),
``TiledZGeneratorInput`` (:101-112, the --g-base tiledz variant) and
``GeneratorOutput`` (:115-129).  Constructor signatures, sub-module names and
registration order follow the reference so state_dicts interchange.
"""
import functools

import torch
from torch import nn

from ... import functional as TF
from ..layers import BatchNorm2d, Conv2d, LeakyReLU, Linear, Tanh, run_layers

_lrelu = functools.partial(LeakyReLU, 0.2)


class ResidualGeneratorBlock(nn.Module):
    """up x2 (nearest) -> [norm, act,] conv3x3, norm, act, conv3x3, plus the upsampled
    input (through a 1x1 projection when the width changes)."""

    def __init__(self, in_dims, out_dims, upsample=True, first_block=False,
                 norm_factory=BatchNorm2d, conv_factory=Conv2d, activation_factory=_lrelu):
        super().__init__()
        body = []
        if not first_block:
            body += [norm_factory(in_dims), activation_factory()]
        body += [conv_factory(in_dims, out_dims, 3, padding=1),
                 norm_factory(out_dims), activation_factory(),
                 conv_factory(out_dims, out_dims, 3, padding=1)]
        self.upsample = upsample
        self.project_input = None
        if in_dims != out_dims:
            self.project_input = nn.Sequential(conv_factory(in_dims, out_dims, 1))
        self.convs = nn.Sequential(*body)

    def forward(self, x):
        """reference generator.py:52-62: x = up(x); h = convs(x); return h + project(x)"""
        if not self.upsample:
            shortcut = x if self.project_input is None else run_layers(self.project_input, x)
            return run_layers(self.convs, x, residual=shortcut)      # x + h, the add fused into the last conv
        mods = list(self.convs)
        low_res_norm = (len(mods) > 2 and isinstance(mods[0], BatchNorm2d) and isinstance(mods[1], LeakyReLU))
        if low_res_norm and (self.project_input is None or all(type(m) is Conv2d for m in self.project_input)):
            # Everything between the block input and the first 3x3 conv commutes with nearest-neighbour upsampling:
            # BatchNorm statistics of an upsampled tensor are those of its source (each element 4 times), LeakyReLU is
            # elementwise, and the 1x1 projection acts per pixel.  So norm + activation and the projection run at
            # the LOW resolution (a quarter of the traffic / FLOPs) and only their results are upsampled.
            # Where the low-resolution plane is large enough for the one-kernel form (TF.upconv3x3_pays), the first 3x3
            # conv is evaluated at the low resolution too: conv3x3(up2x(a)) = four 2x2-tap phase convs, 2.25x fewer FLOPs.
            if x.requires_grad and torch.is_grad_enabled():
                xa, xs = TF.fork(x, 2)           # two consumers: their gradients meet in one kernel instead of an autograd add
            else:
                xa = xs = x
            a = mods[0].forward_act(xa, mods[1].negative_slope, replicate=4)
            # the shortcut stays at the low resolution too: the block's last 3x3 conv adds it upsampled in its epilogue
            # (tg_conv2d_fwd_up2res), no high-resolution copy is written or read
            shortcut = xs if self.project_input is None else run_layers(self.project_input, xs)
            conv = mods[2]
            if type(conv) is Conv2d and conv.kernel_size == (3, 3) and len(mods) > 3 and TF.upconv3x3_pays(a, conv.weight):
                return run_layers(self.convs[3:], TF.upconv3x3(a, conv.weight, conv.bias), residual=shortcut, residual_up=True)
            return run_layers(self.convs[2:], TF.upsample_nearest2x(a), residual=shortcut, residual_up=True)
        xs, xu = TF.fork_upsample_nearest2x(x)           # one graph node for both uses (functional._ForkUp2x)
        shortcut = xs if self.project_input is None else run_layers(self.project_input, xs)
        return run_layers(self.convs, xu, residual=shortcut)


class GeneratorInputMLP(nn.Module):
    """z -> Linear -> act -> (B, C0, size, size)"""

    def __init__(self, latent_dims, output_dims, size=4, norm_factory=None, activation_factory=_lrelu):
        super().__init__()
        self.base_img = nn.Sequential(Linear(latent_dims, size ** 2 * output_dims), activation_factory())
        self.latent_dims = latent_dims
        self.output_dims = output_dims
        self.size = size

    def forward(self, z):
        return run_layers(self.base_img, z).view(-1, self.output_dims, self.size, self.size)


class TiledZGeneratorInput(nn.Module):
    """z tiled over a size x size grid (needs latent_dims == output_dims)."""

    def __init__(self, latent_dims, output_dims, size=4, norm_factory=None, **_):
        super().__init__()
        self.size = size
        assert latent_dims == output_dims

    def forward(self, z):
        return TF._RowBcast.apply(z, (self.size, self.size), 1.0)


class GeneratorOutput(nn.Module):
    """norm -> act -> conv1x1 -> tanh"""

    def __init__(self, in_dims, out_dims, norm_factory=BatchNorm2d, conv_factory=Conv2d,
                 activation_factory=_lrelu, output_activation_factory=Tanh):
        super().__init__()
        self.convs = nn.Sequential(
            norm_factory(in_dims), activation_factory(),
            conv_factory(in_dims, out_dims, 1, padding=0, bias=True),
            output_activation_factory())

    def forward(self, x):
        return run_layers(self.convs, x)
