"""IQN pieces of the reference (models/iqn.py) on HIP kernels.

``CosineQuantileEmbedding`` (iqn.py:27-46), ``IQN`` (:76-108) and ``iqn_loss``
(:111-130).  Quantile fractions tau are drawn from the CPU default generator and
then moved to the device exactly like the reference (:105-108), so a seed gives
the same taus on any device; rows are quantile-major (row = q * B + b).
"""
import torch
from torch import nn

from .. import functional as TF
from .layers import Linear, Tanh, run_layers


class CosineQuantileEmbedding(nn.Module):
    def __init__(self, state_dims, embedding_dims=64, activation=Tanh, norm_factory=None):
        super().__init__()
        self.embedding_dims = embedding_dims
        self.to_state = nn.Sequential(Linear(embedding_dims, state_dims), activation())
        self.register_buffer('embedding_range', torch.arange(1, embedding_dims + 1).float())

    def forward(self, quantiles):
        return run_layers(self.to_state, TF.iqn_cos_embed(quantiles, self.embedding_range))


class IQN(nn.Module):
    def __init__(self, feature_dims, quantile_dims=20, num_quantiles=8, mix='mult',
                 quantile_embedding_factory=CosineQuantileEmbedding, norm_factory=None):
        super().__init__()
        self.quantile_embedding = quantile_embedding_factory(feature_dims, quantile_dims,
                                                             norm_factory=norm_factory)
        self.feature_dims = feature_dims
        self.num_quantiles = num_quantiles
        self.mix = mix
        self._device = None
        self.tau_source = None      # optional callable(rows, num_quantiles) -> (rows, 1) device tensor

    def __getstate__(self):
        state = dict(self.__dict__)
        state['tau_source'] = None          # a trainer's hook (bound to its RNG feed): not part of a pickled model
        return state

    def forward(self, x):
        if isinstance(x, TF.Pair):
            # the real and the fake batch of a discriminator step as one tensor: rows [half][q][b]; each evaluation draws its
            # own taus, real first (trainers/iqn.py:118-119 are two calls)
            batch_size = x.r.shape[0]
            x = TF.repeat_rows(x, self.num_quantiles)
            q_real = self.sample_quantiles(batch_size)
            quantiles = TF.Pair(q_real, self.sample_quantiles(batch_size))
        else:
            batch_size = x.shape[0]
            x = TF.repeat_rows(x, self.num_quantiles)               # (Q*B, C), row = q*B + b
            quantiles = self.sample_quantiles(batch_size)           # (Q*B, 1)
        emb = self.quantile_embedding(quantiles)
        if self.mix == 'add':
            return TF.add(x, emb), quantiles
        elif self.mix.startswith('mult'):
            return TF.mul(x, emb), quantiles
        raise ValueError(f'Unknown mix method {self.mix}')

    def sample_quantiles(self, n=1):
        if self.tau_source is not None:
            return self.tau_source(n * self.num_quantiles, self.num_quantiles)
        if self._device is None:
            self._device = next(self.parameters()).device
        return torch.rand(n * self.num_quantiles, 1).to(self._device)


def iqn_loss(preds, target, taus, k=1.):
    """A Pair of predictions (with the (2B, .) targets of both halves): the sum of the two evaluations' losses."""
    assert not target.requires_grad
    batch_size = target.shape[0]
    num_quantiles = preds.shape[0] // batch_size           # (Pair.shape counts both halves, like target)
    return TF.iqn_quantile_huber_loss(preds, target.reshape(batch_size, -1), taus, num_quantiles, k)
