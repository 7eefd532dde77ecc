"""Procedural stand-ins for Inception activations (TEST INFRASTRUCTURE ONLY): the pretrained network is unreachable
offline, so the FID / IS math is pinned on synthetic pooled features and softmax outputs that are
regenerated from a seed wherever they are needed (the golden generator, the oracle tests, the GPU parity tests)."""
import torch


def procedural_features(n, d, seed, shift=0.0):
    """(n, d) non-negative, correlated features (a ReLU of a low-rank mix plus noise, like pooled activations)."""
    g = torch.Generator().manual_seed(seed)
    rank = max(8, d // 16)
    mix = torch.randn(rank, d, generator=g) / rank ** 0.5
    z = torch.randn(n, rank, generator=g)
    return torch.relu(z @ mix + 0.3 * torch.randn(n, d, generator=g) + 0.5 + shift)


def procedural_probs(n, classes, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.softmax(2.0 * torch.randn(n, classes, generator=g), 1)
