"""HIP-backed layer modules.

The reference's ``models/layers.py:6-22`` holds ``Interpolate`` and
``PixelNorm``; the reference blocks otherwise take stock ``torch.nn`` layers
through their ``norm_factory`` / ``conv_factory`` / ``activation_factory`` /
``avg_pool_factory`` hooks (models/blocks/discriminator.py:50-58,
generator.py:33-35).  The classes here are what this package plugs into those
hooks: same constructor signatures, same parameters / buffers / state_dict
keys and the same default initialisation (they subclass the torch modules and
only replace ``forward``), but the arithmetic runs in libtartangan_amd.so.
"""
import functools

import torch
from torch import nn

from .. import functional as TF


class Conv2d(nn.Conv2d):
    """nn.Conv2d restricted to what the hot path uses: 3x3/pad 1 or 1x1/pad 0, stride 1."""

    def __init__(self, in_channels, out_channels, kernel_size, padding=0, bias=True, **kw):
        super().__init__(in_channels, out_channels, kernel_size, padding=padding, bias=bias, **kw)
        ks = self.kernel_size
        ok = (ks in ((1, 1), (3, 3)) and self.padding == (ks[0] // 2, ks[0] // 2)
              and self.stride == (1, 1) and self.dilation == (1, 1) and self.groups == 1)
        if not ok:
            raise ValueError(f'tartangan_amd.Conv2d supports 3x3/pad1 and 1x1/pad0 stride-1 only, got {self}')

    def forward(self, x):
        return TF.conv2d(x, self.weight, self.bias)


class SpectralNormConv2d(Conv2d):
    """``conv_factory`` with spectral normalisation (what ``torch.nn.utils.spectral_norm(nn.Conv2d(...))`` gives):
    parameters ``weight_orig`` / ``bias``, buffers ``weight_u`` / ``weight_v``, one power iteration per training
    forward, ``weight = weight_orig / sigma``.  Off by default: the reference's trainers never apply it."""

    def __init__(self, *args, n_power_iterations=1, eps=1e-12, **kw):
        super().__init__(*args, **kw)
        weight = self._parameters.pop('weight')
        self.register_parameter('weight_orig', weight)
        with torch.no_grad():
            h, w = weight.shape[0], weight[0].numel()
            u = nn.functional.normalize(weight.new_empty(h).normal_(0, 1), dim=0, eps=eps)
            v = nn.functional.normalize(weight.new_empty(w).normal_(0, 1), dim=0, eps=eps)
        self.register_buffer('weight_u', u)
        self.register_buffer('weight_v', v)
        self.n_power_iterations, self.sn_eps = n_power_iterations, eps

    def forward(self, x):
        w = TF.spectral_normalize(self.weight_orig, self.weight_u, self.weight_v, self.training,
                                  self.n_power_iterations, self.sn_eps)
        return TF.conv2d(x, w, self.bias)


class Linear(nn.Linear):
    def forward(self, x):
        return TF.linear(x, self.weight, self.bias)


class LeakyReLU(nn.LeakyReLU):
    def forward(self, x):
        return TF.leaky_relu(x, self.negative_slope)


class ELU(nn.ELU):
    def forward(self, x):
        return TF.elu(x, self.alpha)


class SELU(nn.SELU):
    def forward(self, x):
        return TF.selu(x)


class Tanh(nn.Tanh):
    def forward(self, x):
        return TF.tanh(x)


class AvgPool2d(nn.AvgPool2d):
    def __init__(self, kernel_size, **kw):
        super().__init__(kernel_size, **kw)
        if kernel_size not in (2, (2, 2)):
            raise ValueError('tartangan_amd.AvgPool2d supports kernel_size=2 only')

    def forward(self, x):
        return TF.avg_pool2(x)


class BatchNorm2d(nn.BatchNorm2d):
    """Train-mode batch statistics or eval-mode running statistics; ``forward_act``
    fuses the LeakyReLU that follows it in every reference block.  ``sync_group`` (a ``functional.SyncGroup``, set by
    ``parallel.DataParallel(sync_bn=True)``; never pickled) makes the training statistics those of the GLOBAL batch."""
    sync_group = None

    def __getstate__(self):
        state = dict(self.__dict__)
        state.pop('sync_group', None)          # process-group handles do not travel with a checkpoint
        return state

    def _run(self, x, slope, replicate=1):
        if self.momentum is None or not self.affine or not self.track_running_stats:
            raise NotImplementedError('tartangan_amd.BatchNorm2d: default nn.BatchNorm2d options only')
        return TF.batch_norm_act(x, self.weight, self.bias, self.running_mean, self.running_var,
                                 self.training, self.momentum, self.eps, slope,
                                 self.num_batches_tracked if self.training else None, replicate, self.sync_group)

    def forward(self, x):
        return self._run(x, 1.0)

    def forward_act(self, x, slope, replicate=1):
        """BatchNorm + LeakyReLU(slope); ``replicate`` = 4 when ``x`` is the low-resolution source of a nearest x2
        upsampled tensor that the reference would have normalised (identical statistics, a quarter of the traffic)."""
        return self._run(x, slope, replicate)


def run_layers(seq, x, residual=None, residual_up=False):
    """Apply an ``nn.Sequential`` the way ``Sequential.forward`` would, fusing each
    (BatchNorm2d, LeakyReLU) pair into one kernel pass.  ``residual`` (the block's shortcut) is added to the
    result; when the last layer is a Conv2d or an AvgPool2d the add rides in that kernel's epilogue.
    ``residual_up``: the shortcut is given at half the resolution of the result (generator blocks); a final 3x3 Conv2d adds it
    upsampled on the fly, anything else gets it upsampled first."""
    mods = list(seq)
    if residual_up and not (mods and type(mods[-1]) is Conv2d and mods[-1].kernel_size == (3, 3)):
        residual, residual_up = TF.upsample_nearest2x(residual), False
    i = 0
    while i < len(mods):
        m = mods[i]
        last = i == len(mods) - 1
        if (isinstance(m, BatchNorm2d) and i + 1 < len(mods) and isinstance(mods[i + 1], LeakyReLU)):
            x = m.forward_act(x, mods[i + 1].negative_slope)
            i += 2
            continue
        if (i == len(mods) - 2 and type(m) is Conv2d and type(mods[i + 1]) is AvgPool2d and m.kernel_size == (3, 3)
                and x.dim() == 4 and TF.pool_conv3x3_supported(x, m.weight)):
            # conv3x3 -> AvgPool2d(2) [+ shortcut]: one 4x4-tap stride-2 convolution (2.25x fewer FLOPs, no full-resolution
            # conv output, no pool pass)
            x, residual = TF.pool_conv3x3(x, m.weight, m.bias, residual), None
            i += 2
            continue
        if last and residual is not None and type(m) is Conv2d:
            x, residual = TF.conv2d(x, m.weight, m.bias, residual, residual_up), None
        elif last and residual is not None and type(m) is AvgPool2d:
            x, residual = TF.avg_pool2(x, residual), None
        else:
            x = m(x)
        i += 1
    if residual is not None:
        x = TF.add(residual, x)
    return x


class Interpolate(nn.Module):
    """models/layers.py:6-14 for the two resamplings the hot path uses."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        self.args = args
        self.kwargs = kwargs

    def forward(self, x):
        return interpolate(x, *self.args, **self.kwargs)


def interpolate(x, size=None, scale_factor=None, mode='nearest', align_corners=None):
    """F.interpolate for (scale 2, nearest) and (scale .5, bilinear, align_corners=True)."""
    if size is None and scale_factor == 2 and mode == 'nearest':
        return TF.upsample_nearest2x(x)
    if size is None and scale_factor == 0.5 and mode == 'bilinear' and align_corners:
        return TF.bilinear_half(x)
    raise NotImplementedError(f'interpolate(scale_factor={scale_factor}, mode={mode}, '
                              f'align_corners={align_corners}) is outside the hot path')


class PixelNorm(nn.Module):
    """models/layers.py:17-22 -- not on the cnn/iqn hot path; kept for API parity."""

    def __init__(self, eps=1e-8):
        super().__init__()
        self.eps = eps

    def forward(self, x):
        raise NotImplementedError('PixelNorm is not used by trainers.cnn / trainers.iqn')


default_activation = functools.partial(LeakyReLU, 0.2)
