"""FID / Inception-score pipeline of ``tartangan.inception_utils`` on the HIP engine.

Both sides of the Inception network are here with the reference's names, arguments and semantics.
In front of it: ``accumulate_inception_activations`` (inception_utils.py:249-268: call ``sample()`` until enough images
exist, ``(s + 1) / 2`` and the VGG mean / std, softmax of the logits) and ``WrapInception`` (:34-95: the same
normalisation once more, bilinear resize to 299 x 299 with align_corners=True, then the wrapped network's layers)
-- normalisation(s) and resize run as ONE kernel (``tg_inception_preprocess``).  Behind it: ``torch_cov`` (:97-124,
including its in-place centring of the argument), ``sqrt_newton_schulz`` (:129-144),
``torch_calculate_frechet_distance`` (:205-235), ``calculate_inception_score`` (:239-246),
``prepare_inception_metrics`` / ``get_inception_metrics`` (:285-328) and, for a data-parallel run, the rank-sharded
form of the moments (``sharded_moments``: all-reduce of the column sums, then of the centred X^T X).
The Inception-v3 network itself is the caller's: its pretrained weights come from the network (``inception_v3(
pretrained=True)``, :273) and cannot be fetched offline -- ``WrapInception`` takes any module with
torchvision's Inception3 attribute names, ``prepare_inception_metrics`` any ``net`` returning ``(pool, logits)``.

The products run on ``tg_gemm_big`` (fp32 MFMA, 128 x 128 tiles, LDS-DMA): one Newton-Schulz
evaluation at 2048 features is 61 products of 2048^3 = 1.05 TFLOP.  No CPU fallback: device tensors only.
"""
import math

import numpy as np
import torch
from torch import nn

from . import backend as _be

VGG_MEAN = torch.tensor([0.485, 0.456, 0.406])[..., None, None]       # inception_utils.py:29-30
VGG_STD = torch.tensor([0.229, 0.224, 0.225])[..., None, None]


def K():
    return _be.get()


def _ws(like, nbytes):
    return torch.empty(max(1, (int(nbytes) + 3) // 4), dtype=torch.float32, device=like.device)


def _matmul(A, B, alpha=1.0, diag=0.0, trans_a=False, out=None):
    """alpha * op(A) @ B + diag * I, 2-D row-major fp32 device tensors."""
    A, B = A.contiguous(), B.contiguous()
    Kd, M = (A.shape[0], A.shape[1]) if trans_a else (A.shape[1], A.shape[0])
    N = B.shape[1]
    assert B.shape[0] == Kd, (A.shape, B.shape, trans_a)
    C = A.new_empty(M, N) if out is None else out
    if K().gemm_big_supported(M, N, Kd, A.shape[1], N, int(trans_a)):
        K().gemm_big(A, B, C, M, N, Kd, A.shape[1], N, N, int(trans_a), float(alpha), float(diag))
        return C
    if diag != 0.0:
        raise NotImplementedError('identity term needs the tiled kernel (dimensions must be multiples of 4)')
    K().gemm(A, B, C, None, M, N, Kd, A.shape[1], N, N, int(trans_a), 0, 1, 0, 0, 0, 0.0)
    if alpha != 1.0:
        K().scale(C, float(alpha), C, C.numel())
    return C


def column_mean(x):
    """torch.mean(x, 0) for a (N, D) tensor."""
    x = x.contiguous()
    N, D = x.shape
    out = x.new_empty(D)
    K().channel_sum(x, out, _ws(x, K().bn_workspace(N, D, 1)), N, D, 1, 0)
    K().scale(out, 1.0 / N, out, D)
    return out


def torch_cov(m, rowvar=False):
    """inception_utils.py:97-124.  Like the reference, the argument is centred IN PLACE (``m -= mean``)."""
    if m.dim() > 2:
        raise ValueError('m has more than 2 dimensions')
    if m.dim() < 2:
        m = m.view(1, -1)
    if not m.is_contiguous():
        raise ValueError('torch_cov needs a contiguous tensor (it centres its argument in place)')
    if not rowvar and m.size(0) != 1:
        N, D = m.shape                       # rows are observations
        fact = 1.0 / (N - 1)
        K().center_rows(m, column_mean(m), N, D)
        return _matmul(m, m, alpha=fact, trans_a=True).squeeze()
    D, N = m.shape                           # rows are variables
    fact = 1.0 / (N - 1)
    mean = m.new_empty(D)
    K().row_sum(m, mean, 1.0 / N, D, N)
    mt = m.t().contiguous()                  # (N, D): centre the transposed copy column-wise, then write it back
    K().center_rows(mt, mean, N, D)
    m.copy_(mt.t())
    return _matmul(mt, mt, alpha=fact, trans_a=True).squeeze()


def sqrt_newton_schulz(A, numIters, dtype=None):
    """inception_utils.py:129-144: A (batch, D, D) -> its matrix square root by ``numIters`` Newton-Schulz steps
    (T = 0.5 (3 I - Z Y); Y <- Y T; Z <- T Z), each step three D^3 products on the matrix cores."""
    with torch.no_grad():
        batch, dim = A.shape[0], A.shape[1]
        out = torch.empty_like(A)
        for b in range(batch):
            a = A[b].contiguous()
            ss = a.new_empty(())
            K().sumsq(a, 1.0, ss, _ws(a, K().reduce_workspace(a.numel())), a.numel())
            norm = math.sqrt(float(ss))                       # ||A||_F (one host read per matrix)
            Y = torch.empty_like(a)
            K().scale(a, 1.0 / norm, Y, a.numel())
            Z = torch.eye(dim, device=a.device, dtype=a.dtype)
            T, Y2, Z2 = torch.empty_like(a), torch.empty_like(a), torch.empty_like(a)
            for _ in range(numIters):
                _matmul(Z, Y, alpha=-0.5, diag=1.5, out=T)    # 0.5 * (3 I - Z Y)
                _matmul(Y, T, out=Y2)
                _matmul(T, Z, out=Z2)
                Y, Y2 = Y2, Y
                Z, Z2 = Z2, Z
            K().scale(Y, math.sqrt(norm), out[b], a.numel())
        return out


def _trace(a):
    out = a.new_empty(())
    K().trace(a.contiguous(), out, a.shape[0], a.shape[1])
    return out


def torch_calculate_frechet_distance(mu1, sigma1, mu2, sigma2, eps=1e-6):
    """inception_utils.py:205-235: ||mu1 - mu2||^2 + Tr(s1) + Tr(s2) - 2 Tr(sqrt(s1 s2)), 20 Newton-Schulz steps.
    Returns a 0-d device tensor like the reference."""
    assert mu1.shape == mu2.shape, 'Training and test mean vectors have different lengths'
    assert sigma1.shape == sigma2.shape, 'Training and test covariances have different dimensions'
    diff = torch.empty_like(mu1)
    K().scale(mu2.contiguous(), -1.0, diff, diff.numel())
    K().add(mu1.contiguous(), diff, diff, diff.numel())
    dd = diff.new_empty(())
    K().sumsq(diff, 1.0, dd, _ws(diff, K().reduce_workspace(diff.numel())), diff.numel())
    covmean = sqrt_newton_schulz(_matmul(sigma1, sigma2).unsqueeze(0), 20).squeeze(0)
    parts = torch.stack([dd, _trace(sigma1), _trace(sigma2), _trace(covmean)]).tolist()   # one host read
    return torch.tensor((parts[0] + parts[1] + parts[2]) - 2 * parts[3], device=mu1.device)


def calculate_inception_score(pred, num_splits=10):
    """inception_utils.py:239-246 on a (N, classes) device tensor of softmax outputs -> (mean, std) floats."""
    pred = pred.contiguous()
    n, classes = pred.shape
    per = n // num_splits
    scores = []
    for index in range(num_splits):
        chunk = pred[index * per:(index + 1) * per]
        rows = chunk.new_empty(per)
        K().is_kl_rows(chunk, column_mean(chunk), rows, per, classes)
        kl = chunk.new_empty(())
        K().row_sum(rows.view(1, per), kl.view(1), 1.0 / per, 1, per)
        scores.append(kl)
    scores = torch.exp(torch.stack(scores).double().cpu())
    return float(scores.mean()), float(scores.std(unbiased=False))


def inception_metrics_from_activations(pool, probs, data_mu, data_sigma, num_splits=10):
    """The tail of ``get_inception_metrics`` (inception_utils.py:307-327) once the pooled features (N, 2048) and the
    softmax outputs (N, 1000) exist: -> (IS_mean, IS_std, FID).  ``pool`` is centred in place, as there."""
    is_mean, is_std = calculate_inception_score(probs, num_splits)
    mu = column_mean(pool)
    sigma = torch_cov(pool, rowvar=False)
    fid = torch_calculate_frechet_distance(mu, sigma, data_mu.float().to(pool.device), data_sigma.float().to(pool.device))
    return is_mean, is_std, float(fid)


# ---------------------------------------------------------------------------------------------- in front of the network
def inception_preprocess(x, size=(299, 299), stages=1, mean=VGG_MEAN, std=VGG_STD):
    """``stages`` times ``x <- ((x + 1) / 2 - mean) / std`` (per channel), then ``F.interpolate(x, size=size, mode='bilinear',
    align_corners=True)`` when the size differs (inception_utils.py:44-50), in one pass over the output."""
    x = x.contiguous().float()
    B, C, H, W = x.shape
    OH, OW = (H, W) if size is None else size
    out = x.new_empty(B, C, OH, OW)
    m = mean.reshape(-1).to(x.device, torch.float32).contiguous()
    sd = std.reshape(-1).to(x.device, torch.float32).contiguous()
    if m.numel() != C or sd.numel() != C:
        raise ValueError(f'{C} channels, but mean / std have {m.numel()} / {sd.numel()} entries')
    K().inception_preprocess(x, m, sd, out, B, C, H, W, OH, OW, int(stages))
    return out


class WrapInception(nn.Module):
    """inception_utils.py:34-95 around a torchvision ``Inception3`` (or anything with its attribute names): normalise, resize
    to 299 x 299, run the stem / Mixed blocks, return ``(pool, logits)``.  ``stages=2`` folds the transform that
    ``accumulate_inception_activations`` applies to generator samples into the same kernel (see ``forward_samples``)."""
    LAYERS = ('Conv2d_1a_3x3', 'Conv2d_2a_3x3', 'Conv2d_2b_3x3', 'pool', 'Conv2d_3b_1x1', 'Conv2d_4a_3x3', 'pool',
              'Mixed_5b', 'Mixed_5c', 'Mixed_5d', 'Mixed_6a', 'Mixed_6b', 'Mixed_6c', 'Mixed_6d', 'Mixed_6e',
              'Mixed_7a', 'Mixed_7b', 'Mixed_7c')

    def __init__(self, net):
        super().__init__()
        self.net = net
        self.mean = nn.Parameter(torch.tensor([0.485, 0.456, 0.406]).view(1, -1, 1, 1), requires_grad=False)
        self.std = nn.Parameter(torch.tensor([0.229, 0.224, 0.225]).view(1, -1, 1, 1), requires_grad=False)

    def _features(self, x):
        for name in self.LAYERS:
            x = nn.functional.max_pool2d(x, kernel_size=3, stride=2) if name == 'pool' else getattr(self.net, name)(x)
        pool = torch.mean(x.view(x.size(0), x.size(1), -1), 2)
        logits = self.net.fc(nn.functional.dropout(pool, training=False).view(pool.size(0), -1))
        return pool, logits

    def forward(self, x):
        return self._features(inception_preprocess(x, (299, 299), 1, self.mean, self.std))

    def forward_samples(self, s):
        """``self(transform(s))`` for generator samples in [-1, 1], ``transform`` being accumulate_inception_activations'
        (inception_utils.py:254-258): both normalisations and the resize in one kernel."""
        return self._features(inception_preprocess(s, (299, 299), 2, self.mean, self.std))


def _softmax_rows(logits):
    logits = logits.contiguous().float()
    out = torch.empty_like(logits)
    K().softmax_fwd(logits, out, logits.shape[0], logits.shape[1])
    return out


def accumulate_inception_activations(sample, net, num_inception_images=50000):
    """inception_utils.py:249-268: run ``sample()`` and ``net`` until ``num_inception_images`` activations exist ->
    (pool (N', D), softmax(logits) (N', classes)), N' the first multiple of the sample batch >= the request."""
    pool, probs, have = [], [], 0
    while have < num_inception_images:
        with torch.no_grad():
            images = sample()
            if hasattr(net, 'forward_samples'):
                pool_val, logits_val = net.forward_samples(images)
            else:
                pool_val, logits_val = net(inception_preprocess(images, None, 1))      # the transform alone (:254-258)
            pool.append(pool_val)
            probs.append(_softmax_rows(logits_val))
            have += logits_val.shape[0]
    return torch.cat(pool, 0), torch.cat(probs, 0)


def load_inception_net(parallel=False, net=None):
    """inception_utils.py:271-278.  The pretrained Inception-v3 weights come from the network; offline the caller supplies
    the loaded torchvision model (``net``)."""
    if net is None:
        try:
            from torchvision.models.inception import inception_v3
            net = inception_v3(pretrained=True, transform_input=False)
        except Exception as exc:         # no torchvision / no network
            raise RuntimeError('load_inception_net: pass a loaded torchvision Inception3 as `net` (the pretrained weights '
                               f'cannot be fetched here: {exc!r})') from exc
    return WrapInception(net.eval())


def sharded_moments(pool, group=None):
    """Mean and covariance (``torch.mean(pool, 0)``, ``torch_cov(pool)``) of the union of every rank's ``pool`` rows, on every
    rank: all-reduce of the column sums and the row count, local centring with the GLOBAL mean, all-reduce of the centred
    X^T X (D x D: 16 MB at 2048 features, against N x 2048 features to gather).  ``pool`` is centred in place, like
    ``torch_cov`` does.  One rank / no process group: exactly ``column_mean`` + ``torch_cov``."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        mu = column_mean(pool)
        return mu, torch_cov(pool, rowvar=False)
    pool = pool.contiguous()
    n_local, D = pool.shape
    sums = pool.new_empty(D)
    K().channel_sum(pool, sums, _ws(pool, K().bn_workspace(n_local, D, 1)), n_local, D, 1, 0)
    head = torch.cat([sums.double(), torch.tensor([float(n_local)], dtype=torch.float64, device=pool.device)])
    dist.all_reduce(head, group=group)
    n = int(head[-1].item())
    mu = (head[:-1] / n).float().contiguous()
    K().center_rows(pool, mu, n_local, D)
    xtx = _matmul(pool, pool, trans_a=True)
    dist.all_reduce(xtx, group=group)
    K().scale(xtx, 1.0 / (n - 1), xtx, xtx.numel())
    return mu, xtx


def prepare_inception_metrics(moments_path, device, parallel=False, no_fid=False, net=None, group=None):
    """inception_utils.py:285-328 -> ``get_inception_metrics(sample, num_inception_images, num_splits=10, prints=True,
    use_torch=True)`` returning ``(IS_mean, IS_std, FID)``.  ``moments_path``: the .npz with ``mu`` / ``sigma`` written by
    calculate_inception_moments.  ``net``: a loaded network, see ``load_inception_net``.  Under data parallelism
    (``group`` / an initialised default group) every rank samples its own share and the moments are combined with
    ``sharded_moments``; the per-sample class probabilities are gathered for the Inception score."""
    data = np.load(moments_path)
    data_mu = torch.tensor(data['mu']).float().to(device)
    data_sigma = torch.tensor(data['sigma']).float().to(device)
    net = net if (net is not None and not isinstance(net, type)) else load_inception_net(parallel)
    if isinstance(net, nn.Module):
        net = net.to(device)

    def get_inception_metrics(sample, num_inception_images, num_splits=10, prints=True, use_torch=True):
        import torch.distributed as dist
        world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        share = -(-num_inception_images // world)
        pool, probs = accumulate_inception_activations(sample, net, share)
        if world > 1:
            gathered = [torch.empty_like(probs) for _ in range(world)]
            dist.all_gather(gathered, probs.contiguous(), group=group)
            probs = torch.cat(gathered, 0)
        IS_mean, IS_std = calculate_inception_score(probs, num_splits)
        if no_fid:
            return IS_mean, IS_std, 9999.0
        mu, sigma = sharded_moments(pool, group)
        FID = float(torch_calculate_frechet_distance(mu, sigma, data_mu, data_sigma))
        return IS_mean, IS_std, FID
    return get_inception_metrics
