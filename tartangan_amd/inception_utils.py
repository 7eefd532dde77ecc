"""FID / Inception-score math of ``tartangan.inception_utils`` on the HIP engine.

Everything downstream of the Inception network is here with the reference's names, arguments and
semantics: ``torch_cov`` (inception_utils.py:97-124, including its in-place centring of the argument),
``sqrt_newton_schulz`` (:129-144), ``torch_calculate_frechet_distance`` (:205-235),
``calculate_inception_score`` (:239-246) and ``inception_metrics_from_activations`` = the tail of
``get_inception_metrics`` (:301-328) once ``pool`` / ``logits`` exist.  The Inception-v3 forward itself needs
pretrained weights from the network and is out of reach offline; a caller that has the
reference's ``WrapInception`` feeds its outputs in.

The products run on ``tg_gemm_big`` (fp32 MFMA, 128 x 128 tiles, LDS-DMA): one Newton-Schulz
evaluation at 2048 features is 61 products of 2048^3 = 1.05 TFLOP.  No CPU fallback: device tensors only.
"""
import math

import torch

from . import backend as _be


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
