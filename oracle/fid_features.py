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


def tiny_inception(features=64, classes=10, seed=5):
    """A stand-in with torchvision Inception3's ATTRIBUTE NAMES (what ``WrapInception`` calls, reference
    inception_utils.py:51-95) and a few procedural weights: enough structure to run the reference's own WrapInception /
    accumulate_inception_activations and this package's on the same inputs.  Not an Inception network."""
    from torch import nn
    g = torch.Generator().manual_seed(seed)
    net = nn.Module()
    for name in ('Conv2d_2a_3x3', 'Conv2d_2b_3x3', 'Conv2d_4a_3x3', 'Mixed_5b', 'Mixed_5c', 'Mixed_5d', 'Mixed_6a', 'Mixed_6b',
                 'Mixed_6c', 'Mixed_6d', 'Mixed_6e', 'Mixed_7a', 'Mixed_7b'):
        setattr(net, name, nn.Identity())
    net.Conv2d_1a_3x3 = nn.AvgPool2d(13, 13)                     # 299 -> 23 (then the two 3/2 max-pools: 11, 5)
    net.Conv2d_3b_1x1 = nn.Conv2d(3, 16, 1)
    net.Mixed_7c = nn.Sequential(nn.Conv2d(16, features, 1), nn.ReLU())
    net.fc = nn.Linear(features, classes)
    with torch.no_grad():
        for name, p in net.named_parameters():
            scale = 0.02 if name.startswith('fc.') else (0.5 if p.dim() > 1 else 0.1)       # (logits of order 1: a soft softmax)
            p.copy_(torch.randn(p.shape, generator=g) * scale)
    return net.eval()


def blocky_images(batch, size, seed):
    """(batch, 3, size, size) in [-1, 1] with low-frequency content (4 x 4 random blocks + a little noise), so that pooled
    features differ from image to image (plain uniform noise averages out under the stand-in's pooling)."""
    g = torch.Generator().manual_seed(seed)
    coarse = torch.rand(batch, 3, 4, 4, generator=g) * 2 - 1
    img = coarse.repeat_interleave(-(-size // 4), 2).repeat_interleave(-(-size // 4), 3)[:, :, :size, :size]
    return (0.85 * img + 0.15 * (torch.rand(batch, 3, size, size, generator=g) * 2 - 1)).contiguous()
