"""Discriminator blocks of the reference (models/blocks/discriminator.py) on HIP kernels.

Implemented: ``DiscriminatorInput`` (discriminator.py:11-22),
``ResidualDiscriminatorBlock`` (:49-95), ``DiscriminatorOutput`` (:126-146),
``IQNDiscriminatorOutput`` (:149-178).  Every op here is differentiable twice
(the real-image branch sits under the R1 penalty).
"""
import functools

import torch
from torch import nn

from ... import functional as TF
from ..iqn import IQN, iqn_loss
from ..layers import AvgPool2d, BatchNorm2d, Conv2d, LeakyReLU, Linear, interpolate, run_layers

_lrelu = functools.partial(LeakyReLU, 0.2)
_half = functools.partial(interpolate, scale_factor=0.5, mode='bilinear', align_corners=True)


class DiscriminatorInput(nn.Module):
    """from-RGB 1x1 convolution"""

    def __init__(self, in_dims, out_dims, conv_factory=Conv2d, activation_factory=_lrelu):
        super().__init__()
        self.convs = nn.Sequential(conv_factory(in_dims, out_dims, 1, padding=0, bias=True))

    def forward(self, img):
        return run_layers(self.convs, img)


class ResidualDiscriminatorBlock(nn.Module):
    """[norm, act,] conv3x3, norm, act, conv3x3, avgpool2  +  bilinear-half(x) [-> conv1x1]"""

    def __init__(self, in_dims, out_dims, first_block=False, norm_factory=BatchNorm2d,
                 conv_factory=Conv2d, avg_pool_factory=AvgPool2d, activation_factory=_lrelu,
                 interpolate=_half):
        super().__init__()
        body = []
        if not first_block:
            body += [norm_factory(in_dims), activation_factory()]
        body += [conv_factory(in_dims, out_dims, 3, padding=1, bias=True),
                 norm_factory(out_dims), activation_factory(),
                 conv_factory(out_dims, out_dims, 3, padding=1, bias=True),
                 avg_pool_factory(2)]
        self.convs = nn.Sequential(*body)
        self.in_dims = in_dims
        self.out_dims = out_dims
        self.project_input = None
        if in_dims != out_dims:
            self.project_input = nn.Sequential(conv_factory(in_dims, out_dims, 1))
        self.interpolate = interpolate

    def forward(self, x):
        if self.interpolate is _half:
            shortcut, x = TF.fork_bilinear_half(x)      # one graph node for both uses of x (see functional._ForkBilinearHalf)
        else:
            shortcut = self.interpolate(x)
        if self.project_input is not None:
            shortcut = run_layers(self.project_input, shortcut)
        return run_layers(self.convs, x, residual=shortcut)          # x + h, the add fused into the avg-pool

    def default_resampling(self):
        return self.interpolate is _half

    # ---- the first block fused with the from-RGB 1x1 convolution in front of it
    def fuses_with(self, rgb):
        """True if ``rgb`` (a DiscriminatorInput) followed by this block can run as ``forward_from_rgb``: the block starts
        with a plain 3x3 convolution (first_block: no norm / activation in between), keeps its width and uses the
        default resampling."""
        mods = list(self.convs)
        rc = list(getattr(rgb, 'convs', []))
        return (isinstance(rgb, DiscriminatorInput) and len(rc) == 1 and type(rc[0]) is Conv2d and rc[0].kernel_size == (1, 1)
                and rc[0].bias is not None and len(mods) > 1 and type(mods[0]) is Conv2d and mods[0].kernel_size == (3, 3)
                and self.project_input is None and self.interpolate is _half)

    def forward_from_rgb(self, img, rgb):
        """block(rgb(img)) with the two linear maps in front composed (discriminator.py:11-22 + :60-61: a 1x1 convolution
        directly followed by a 3x3 one).  W'[co][c][tap] = sum_m W3[co][m][tap] W1[m][c]; the from-RGB bias rides on an
        extra all-ones input channel so that zero padding treats it exactly like the reference does at the borders.
        One 4->C 3x3 convolution replaces a 3->C 1x1 and a C->C 3x3 one (C = 16 at 128 px: 4x fewer multiply-adds at
        the full resolution), and the shortcut's bilinear-1/2 runs on the 3-channel image (it commutes with the 1x1)."""
        rgb_conv, conv1 = rgb.convs[0], self.convs[0]
        C, Cimg = rgb_conv.weight.shape[:2]
        Cout = conv1.weight.shape[0]
        wc = TF.compose_rgb_filter(rgb_conv.weight, rgb_conv.bias, conv1.weight)                      # one launch (and one backwards)
        img_a = img_b = img
        if img.requires_grad and torch.is_grad_enabled():
            img_a, img_b = TF.fork(img, 2)         # two consumers (R1 differentiates w.r.t. the images): one fan-in kernel
        h = TF.conv2d(TF.copy_channels(img_a, Cimg + 1, 1.0), wc.view(Cout, Cimg + 1, 3, 3), conv1.bias)     # [img, 1] in one pass
        shortcut = run_layers(rgb.convs, self.interpolate(img_b))
        return run_layers(self.convs[1:], h, residual=shortcut)


class DiscriminatorOutput(nn.Module):
    """norm -> act -> sum over (H, W) -> Linear"""

    def __init__(self, in_dims, out_dims, norm_factory=BatchNorm2d, activation_factory=_lrelu,
                 output_activation_factory=nn.Identity):
        super().__init__()
        self.activation = nn.Sequential(norm_factory(in_dims), activation_factory())
        self.to_output = nn.Sequential(Linear(in_dims, out_dims), output_activation_factory())

    def forward(self, feats):
        feats = TF.sum_hw(run_layers(self.activation, feats))
        return run_layers(self.to_output, feats)


class IQNDiscriminatorOutput(nn.Module):
    """norm -> act -> sum pool -> IQN quantile mixing -> Linear; returns the quantile
    mean and, when targets are given, the quantile Huber loss."""

    def __init__(self, in_dims, out_dims, norm_factory=BatchNorm2d, activation_factory=_lrelu):
        super().__init__()
        self.activation = nn.Sequential(norm_factory(in_dims), activation_factory())
        self.to_output = nn.Sequential(Linear(in_dims, out_dims))
        self.iqn = IQN(in_dims)
        self.out_dims = out_dims

    def forward(self, feats, targets=None):
        feats = TF.sum_hw(run_layers(self.activation, feats))
        feats_tau, taus = self.iqn(feats)
        p_target_tau = run_layers(self.to_output, feats_tau)
        loss = None
        if targets is not None:
            if self.out_dims != 1:
                raise NotImplementedError('IQN loss kernel covers out_dims == 1 (the trainers\' case)')
            loss = iqn_loss(p_target_tau, targets, taus)
        p_target = TF.mean_reps(p_target_tau, self.iqn.num_quantiles)
        if targets is not None:
            return p_target, loss
        return p_target
