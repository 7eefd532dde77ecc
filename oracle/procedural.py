"""Deterministic procedural weights + synthetic inputs shared by the golden
generator, the CPU oracle tests and the GPU parity tests.

TEST INFRASTRUCTURE ONLY -- nothing under ``tartangan_amd/`` may import this.

Weights are never stored in fixtures (SURVEY.md §8c "What travels"): they are
regenerated from (seed, tensor name, tensor shape) so that the reference
modules, the oracle and the HIP modules can all be loaded with the *same*
values through their shared ``state_dict`` keys.
"""
import zlib

import torch


def _gen(seed, name):
    g = torch.Generator()
    g.manual_seed((seed * 1000003 + zlib.crc32(name.encode())) % (2 ** 31 - 1))
    return g


def procedural_state(template, seed):
    """Return {name: tensor} with the same keys/shapes/dtypes as ``template``.

    ``template`` is any mapping name -> tensor (e.g. ``module.state_dict()``).
    """
    out = {}
    for name, t in template.items():
        leaf = name.rsplit('.', 1)[-1]
        g = _gen(seed, name)
        if leaf == 'num_batches_tracked':
            v = torch.zeros_like(t)
        elif leaf == 'embedding_range':
            v = t.clone()
        elif leaf == 'running_mean':
            v = 0.05 * torch.randn(t.shape, generator=g)
        elif leaf == 'running_var':
            v = 1.0 + 0.1 * torch.rand(t.shape, generator=g)
        elif leaf == 'gamma':
            v = torch.full(t.shape, 0.7)
        elif leaf == 'bias':
            v = 0.1 * torch.randn(t.shape, generator=g)
        elif leaf == 'weight' and t.dim() == 1:
            v = 1.0 + 0.1 * torch.randn(t.shape, generator=g)      # norm scale
        elif leaf == 'weight':
            fan_in = t[0].numel()
            v = torch.randn(t.shape, generator=g) * (1.6 / fan_in) ** 0.5
        else:
            raise KeyError(f'no procedural rule for {name}')
        out[name] = v.to(t.dtype)
    return out


def synthetic_images(batch, size, seed=1234):
    """U[-1,1] fp32 images, the range Normalize(.5,.5) produces."""
    g = torch.Generator()
    g.manual_seed(seed)
    return torch.rand(batch, 3, size, size, generator=g) * 2 - 1


def summarize(t, n_samples=8):
    """Compact, layout-independent description of a tensor for fixtures."""
    t = t.detach().to(torch.float64).reshape(-1).cpu()
    n = t.numel()
    idx = [(i * 2654435761 + 12345) % n for i in range(n_samples)] if n else []
    return dict(
        numel=n,
        sum=float(t.sum()),
        abs_sum=float(t.abs().sum()),
        l2=float(t.pow(2).sum().sqrt()),
        max_abs=float(t.abs().max()) if n else 0.0,
        idx=idx,
        samples=[float(t[i]) for i in idx],
    )
