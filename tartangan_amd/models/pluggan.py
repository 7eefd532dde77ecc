"""Factory-assembled Generator / Discriminator / IQNDiscriminator.

Mirror of the reference's plug-in API (models/pluggan.py:18-132, configs
:199-406): a model is ``BlockModel(config, input_factory, block_factory,
output_factory)`` and the factories are called with exactly the reference's
arguments, so ``functools.partial``-bound block factories written for tartangan
work unchanged.  The default factories here are the HIP-backed blocks of
``tartangan_amd.models.blocks``.
"""
from collections import namedtuple

from torch import nn

from .. import functional as TF
from .blocks import (
    DiscriminatorInput, DiscriminatorOutput, GeneratorInputMLP, GeneratorOutput,
    ResidualDiscriminatorBlock, ResidualGeneratorBlock, SelfAttention2d,
)

_FIELDS = 'base_size, latent_dims, data_dims, blocks, num_blocks_per_scale, attention'


class GANConfig(namedtuple('GANConfig', _FIELDS)):
    def scale_model(self, scale):
        """Multiply every block width by ``scale`` (int() truncation, pluggan.py:24-28)."""
        return self._replace(blocks=[int(c * scale) for c in self.blocks])


class BlockModel(nn.Module):
    default_input = default_block = default_output = None

    def __init__(self, config, input_factory=None, block_factory=None, output_factory=None):
        super().__init__()
        self.config = config
        self.input_factory = input_factory or self.default_input
        self.block_factory = block_factory or self.default_block
        self.output_factory = output_factory or self.default_output
        self.build()

    def build(self):
        raise NotImplementedError

    def forward(self, x):
        for block in self.blocks:
            x = block(x)
        return x

    @property
    def max_size(self):
        return self.config.base_size * 2 ** len(self.config.blocks)

    def _wants_attention(self, block_i):
        return bool(self.config.attention) and block_i in self.config.attention


class Generator(BlockModel):
    default_input = GeneratorInputMLP
    default_block = ResidualGeneratorBlock
    default_output = GeneratorOutput

    def build(self):
        cfg = self.config
        width = cfg.blocks[0]
        stages = [self.input_factory(cfg.latent_dims, width, cfg.base_size)]
        for block_i, out_dims in enumerate(cfg.blocks):
            stages.append(self.block_factory(width, out_dims, first_block=(block_i == 0)))
            for _ in range(cfg.num_blocks_per_scale - 1):
                stages.append(self.block_factory(out_dims, out_dims, upsample=False))
            if self._wants_attention(block_i):
                stages.append(SelfAttention2d(out_dims))
            width = out_dims
        stages.append(self.output_factory(width, cfg.data_dims))
        self.blocks = nn.Sequential(*stages)


class Discriminator(BlockModel):
    default_input = DiscriminatorInput
    default_block = ResidualDiscriminatorBlock
    default_output = DiscriminatorOutput

    def build(self):
        cfg = self.config
        width = cfg.blocks[-1]
        stages = [self.input_factory(cfg.data_dims, width)]
        first = True
        for block_i in reversed(range(len(cfg.blocks))):
            out_dims = cfg.blocks[block_i]
            stages.append(self.block_factory(width, out_dims, first_block=first))
            if self._wants_attention(block_i):
                stages.append(SelfAttention2d(out_dims))
            width, first = out_dims, False
        stages.append(self.output_factory(width, 1))
        self.blocks = nn.Sequential(*stages)

    def forward(self, x):
        blocks = list(self.blocks)
        if len(blocks) > 1 and getattr(blocks[1], 'fuses_with', lambda _: False)(blocks[0]):
            x = blocks[1].forward_from_rgb(x, blocks[0])        # from-RGB 1x1 composed into the first block's 3x3
            blocks = blocks[2:]
        for block in blocks:
            x = block(x)
        return x


class IQNDiscriminator(Discriminator):
    """No from-RGB conv: the first residual block runs on the 3 image channels; the
    head takes ``targets`` and returns ``(p_target, loss)`` (pluggan.py:114-132)."""

    def build(self):
        cfg = self.config
        width = cfg.data_dims
        stages = []
        for block_i in reversed(range(len(cfg.blocks))):
            out_dims = cfg.blocks[block_i]
            stages.append(self.block_factory(width, out_dims))
            if self._wants_attention(block_i):
                stages.append(SelfAttention2d(out_dims))
            width = out_dims
        self.to_output = self.output_factory(width, 1)      # registered before .blocks, like the reference
        self.blocks = nn.Sequential(*stages)

    def forward(self, x, targets=None):
        # a Pair (real | fake as one batch): blocks and head alike on 2B images; the head keeps each half's rows quantile-major
        # on their own and draws each half's taus separately, real first (iqn.py:118-119); with targets (2B, 1) its loss is
        # loss_real + loss_fake
        return self.to_output(self.blocks(x), targets=targets)


def _cfg(latent, blocks, attention=()):
    return GANConfig(base_size=4, data_dims=3, latent_dims=latent, attention=attention,
                     num_blocks_per_scale=1, blocks=tuple(blocks))


# widths per scale (8, 16, 32, ... px), pluggan.py:199-406
GAN_CONFIGS = {
    '16': _cfg(100, (64, 32)),
    '32': _cfg(128, (128, 64, 32)),
    '64': _cfg(128, (128, 128, 64, 32)),
    '128': _cfg(256, (128, 128, 64, 32, 16)),
    '128big': _cfg(256, (1024, 1024, 512, 256, 128)),
    '256': _cfg(256, (256, 256, 128, 64, 32, 16)),
    '256big': _cfg(256, (1024, 1024, 512, 256, 128, 64)),
    '512': _cfg(512, (256, 256, 256, 128, 64, 32, 16)),
    '512thin': _cfg(256, (128, 128, 128, 64, 32, 16, 8), (3,)),
    '512thin-test': _cfg(128, (128, 120, 100, 64, 32, 16, 8), (3,)),
    '1024': _cfg(512, (512, 512, 512, 256, 128, 64, 32, 16), (3,)),
    '1024thin': _cfg(256, (256, 256, 256, 128, 64, 32, 16, 8), (3,)),
    'test128': _cfg(64, (64, 32, 16, 8, 4), (3,)),
    'test256': _cfg(256, (200, 180, 128, 64, 32, 16), (3,)),
}
