"""FID / Inception-score math (tartangan_amd.inception_utils) against golden vectors produced by the reference's own
functions (tests/golden/make_fid_golden.py -> fid_math.json) on procedural features.

CPU (not gpu): the host logic over the emulator.  GPU: the HIP kernels (tg_gemm_big etc.) through the C ABI, incl. the
2048-feature case of BASELINE.json's config 5; tolerance 1e-4 relative on FID, as for the losses."""
import json
import os

import pytest
import torch

from conftest import GOLDEN_DIR
from emulator import Emulator
from oracle.fid_features import procedural_features, procedural_probs
from oracle.procedural import summarize

with open(os.path.join(GOLDEN_DIR, 'fid_math.json')) as f:
    FIX = json.load(f)


def _close(a, b, rel, abs_=0.0):
    return abs(a - b) <= abs_ + rel * max(abs(a), abs(b))


def _check_summary(t, ref, rel, what):
    got = summarize(t, len(ref['idx']))
    assert got['numel'] == ref['numel'], what
    assert _close(got['l2'], ref['l2'], rel), (what, got['l2'], ref['l2'])
    for g, r in zip(got['samples'], ref['samples']):
        assert abs(g - r) <= 4 * rel * ref['max_abs'] + 1e-9, (what, g, r)


def _run(name, device, rel):
    from tartangan_amd import inception_utils as IU
    fx = FIX[name]
    D, N = fx['D'], fx['N']
    gen = procedural_features(N, D, 11).to(device)
    data = procedural_features(N, D, 12, shift=0.25).to(device)
    mu1, mu2 = IU.column_mean(gen), IU.column_mean(data)
    g2 = gen.clone()
    s1 = IU.torch_cov(g2, rowvar=False)
    s2 = IU.torch_cov(data.clone(), rowvar=False)
    _check_summary(s1, fx['cov'], rel, 'cov')
    _check_summary(g2, fx['centred_in_place'], rel, 'in-place centring')       # the reference's side effect
    root = IU.sqrt_newton_schulz(IU._matmul(s1, s2).unsqueeze(0), 20).squeeze(0)
    assert _close(float(IU._trace(root)), fx['sqrt_trace'], rel), (float(IU._trace(root)), fx['sqrt_trace'])
    _check_summary(root, fx['sqrt'], 10 * rel, 'sqrt')
    fid = float(IU.torch_calculate_frechet_distance(mu1, s1, mu2, s2))
    assert _close(fid, fx['fid'], rel), (fid, fx['fid'])
    probs = procedural_probs(N, fx['classes'], 13).to(device)
    m, s = IU.calculate_inception_score(probs, fx['splits'])
    assert _close(m, fx['is_mean'], 1e-5) and _close(s, fx['is_std'], 1e-3, 1e-6), (m, s, fx['is_mean'], fx['is_std'])
    # the whole tail of get_inception_metrics in one call
    pool = procedural_features(N, D, 11).to(device)
    mu_d, sig_d = mu2, s2
    im, istd, f2 = IU.inception_metrics_from_activations(pool, probs, mu_d, sig_d, fx['splits'])
    assert _close(f2, fx['fid'], rel) and _close(im, fx['is_mean'], 1e-5)


@pytest.mark.parametrize('name', ['d64_n200', 'd256_n1000'])
def test_fid_math_host_logic_matches_reference(name, single_thread):
    from tartangan_amd import backend
    prev = backend._set_backend_for_testing(Emulator())
    try:
        _run(name, 'cpu', 2e-5)
    finally:
        backend._set_backend_for_testing(prev)


@pytest.mark.gpu
@pytest.mark.parametrize('name', sorted(FIX))
def test_fid_math_hip_matches_reference(name):
    from tartangan_amd import backend
    backend._set_backend_for_testing(None)
    _run(name, 'cuda', 1e-4)


@pytest.mark.gpu
def test_gemm_big_against_fp64_and_ragged_shapes():
    """tg_gemm_big on shapes that exercise the bounds-checked edges (M, N not multiples of 128, K tails), both operand
    forms, and the alpha / diagonal epilogue; reference = fp64 matmul of the same fp32 inputs."""
    from tartangan_amd import backend
    backend._set_backend_for_testing(None)
    K = backend.get()
    g = torch.Generator().manual_seed(5)
    for (M, N, Kd, ta) in [(128, 128, 32, 0), (256, 384, 100, 0), (200, 136, 68, 0), (200, 136, 70, 1), (2048, 2048, 999, 1), (520, 520, 520, 0)]:
        A = torch.randn((Kd, M) if ta else (M, Kd), generator=g)
        B = torch.randn(Kd, N, generator=g)
        want = (-0.5 * ((A.t() if ta else A).double() @ B.double()) + 1.5 * torch.eye(M, N, dtype=torch.float64))
        C = torch.full((M, N), float('nan'), device='cuda')
        assert K.gemm_big_supported(M, N, Kd, A.shape[1], N, ta)
        K.gemm_big(A.cuda(), B.cuda(), C, M, N, Kd, A.shape[1], N, N, ta, -0.5, 1.5)
        err = (C.cpu().double() - want).abs().max().item()
        assert err <= 2e-6 * Kd ** 0.5 * max(1.0, want.abs().max().item()), (M, N, Kd, ta, err)
