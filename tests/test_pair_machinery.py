"""functional.Pair / _Paired, op by op (CPU, over the reference-semantics emulator): for every Function that has a paired
form, the outputs of ONE pass over the two halves equal two separate passes, and so do the gradients -- when both halves
receive one (the joined backward), when only the first does (the R1 penalty's case, incl. a second derivative through it) and
when only the second does (the generator's shared pass)."""
import pytest
import torch

from emulator import Emulator
from tartangan_amd import backend, functional as TF


@pytest.fixture(autouse=True)
def emulated_backend():
    prev = backend._set_backend_for_testing(Emulator())
    yield
    backend._set_backend_for_testing(prev)


def _t(*shape, seed=0, grad=True):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return torch.randn(*shape, generator=g).requires_grad_(grad)


B, C, H = 3, 8, 8
W3 = _t(C, C, 3, 3, seed=1)
W1 = _t(C, C, 1, 1, seed=2)
BIAS = _t(C, seed=3)
GAMMA, BETA = _t(C, seed=4), _t(C, seed=5)
WL, BL = _t(5, C, seed=6), _t(5, seed=7)


def _bn(x):
    return TF.batch_norm_act(x, GAMMA, BETA, torch.zeros(C), torch.ones(C), True, 0.1, 1e-5, 0.2, torch.zeros((), dtype=torch.int64))


# name -> (callable taking one tensor-or-Pair (B, C, H, H) and returning a tensor-or-Pair, parameters it uses)
OPS = {
    'conv3x3': (lambda x: TF.conv2d(x, W3, BIAS), [W3, BIAS]),
    'conv3x3+residual': (lambda x: TF.conv2d(x, W3, BIAS, x), [W3, BIAS]),
    'conv1x1': (lambda x: TF.conv2d(x, W1, None), [W1]),
    'pool_conv3x3': (lambda x: TF.pool_conv3x3(x, W3, BIAS), [W3, BIAS]),
    'upconv3x3': (lambda x: TF.upconv3x3(x, W3, BIAS), [W3, BIAS]),
    'bn_lrelu': (_bn, [GAMMA, BETA]),
    'fork_bilinear_half': (lambda x: TF.fork_bilinear_half(x)[0], []),
    'fork_bilinear_half:pass': (lambda x: TF.fork_bilinear_half(x)[1], []),
    'bilinear_half': (TF.bilinear_half, []),
    'avg_pool2': (TF.avg_pool2, []),
    'max_pool2': (TF.max_pool2, []),
    'upsample_nearest2x': (TF.upsample_nearest2x, []),
    'fork_upsample_nearest2x': (lambda x: TF.fork_upsample_nearest2x(x)[1], []),
    'fork3': (lambda x: TF.add(TF.fork(x, 3)[0], TF.fork(x, 3)[2]), []),
    'sum_hw': (TF.sum_hw, []),
    'sum_hw+linear': (lambda x: TF.linear(TF.sum_hw(x), WL, BL), [WL, BL]),
    'leaky_relu': (TF.leaky_relu, []),
    'elu': (TF.elu, []),
    'selu': (TF.selu, []),
    'tanh': (TF.tanh, []),
    'add': (lambda x: TF.add(x, x), []),
    'scale_add': (lambda x: TF.scale_add(GAMMA[0], x, x), [GAMMA]),
    'copy_channels': (lambda x: TF.copy_channels(x, C + 1, 1.0), []),
    'qkv': (lambda x: TF.qkv_projections(x, W1[:2], W1[2:4], W1[4:])[2], [W1]),
}


def _heads(x, d, dv):
    """theta (B, d, N), phi (B, d, N/4), g (B, dv, N/4) carved out of x by differentiable ops that exist in paired form."""
    b = x.shape[0]
    pooled = TF.max_pool2(x)
    theta = TF.conv2d(x, W1[:d], None).view(b, d, H * H)
    phi = TF.conv2d(pooled, W1[1:1 + d], None).view(b, d, H * H // 4)
    g = TF.conv2d(pooled, W1[8 - dv:], None).view(b, dv, H * H // 4)
    return theta, phi, g


OPS['attention_core'] = (lambda x: TF.attention_core(*_heads(x, 1, 4)), [W1])          # head dims with a fused kernel
OPS['attention_composed'] = (lambda x: TF.attention_core(*_heads(x, 3, 3)), [W1])      # without: per-half composition
NOT_JOINED = {'attention_composed'}


def _grads(out, wrt):
    gs = torch.autograd.grad(out, wrt, allow_unused=True)
    return [torch.zeros_like(w) if g is None else g for g, w in zip(gs, wrt)]


@pytest.mark.parametrize('name', sorted(OPS))
def test_paired_op_equals_two_separate_passes(name):
    fn, params = OPS[name]
    xr, xf = _t(B, C, H, H, seed=11), _t(B, C, H, H, seed=12)
    out = fn(TF.Pair(xr, xf))
    sep_r, sep_f = fn(xr), fn(xf)
    assert isinstance(out, TF.Pair)
    assert torch.allclose(out.r, sep_r, rtol=1e-5, atol=1e-6) and torch.allclose(out.f, sep_f, rtol=1e-5, atol=1e-6)
    if name not in NOT_JOINED:
        assert out.f.data_ptr() == out.r.data_ptr() + out.r.numel() * 4      # halves back to back: the next op joins for free
    wr, wf = torch.randn_like(sep_r), torch.randn_like(sep_f)
    wrt = [xr, xf] + params
    # both halves (joined backward), the first half alone, the second half alone
    for cr, cf in ((1.0, 1.0), (1.0, 0.0), (0.0, 1.0)):
        terms_p = [(o * w).sum() for o, w, c in ((out.r, wr, cr), (out.f, wf, cf)) if c]
        terms_s = [(o * w).sum() for o, w, c in ((sep_r, wr, cr), (sep_f, wf, cf)) if c]
        got = torch.autograd.grad(sum(terms_p), wrt, retain_graph=True, allow_unused=True)
        want = torch.autograd.grad(sum(terms_s), wrt, retain_graph=True, allow_unused=True)
        for k, (a, b) in enumerate(zip(got, want)):
            if b is None or a is None:
                assert (a is None or float(a.abs().max()) == 0.0) and (b is None or float(b.abs().max()) == 0.0), (name, k, cr, cf)
                continue
            assert torch.allclose(a, b, rtol=2e-4, atol=2e-5), (name, k, cr, cf, float((a - b).abs().max()))


@pytest.mark.parametrize('name', ['conv3x3', 'pool_conv3x3', 'bn_lrelu', 'conv1x1', 'qkv', 'bilinear_half', 'avg_pool2', 'max_pool2', 'leaky_relu',
                                  'attention_core', 'scale_add', 'copy_channels', 'sum_hw', 'add', 'sum_hw+linear', 'elu', 'selu',
                                  'fork3', 'upconv3x3', 'tanh'])
def test_second_derivative_through_the_first_half_of_a_paired_op(name):
    """R1: d/d(params) of || d out.r / d x_r ||^2 through the paired node equals the same through a separate pass."""
    fn, params = OPS[name]
    xr, xf = _t(B, C, H, H, seed=21), _t(B, C, H, H, seed=22)
    vals = []
    for paired in (True, False):
        # tanh in front: the linear ops get a second derivative to carry
        o = fn(TF.tanh(TF.Pair(xr, xf))).r if paired else fn(TF.tanh(xr))
        w = torch.ones_like(o)
        g, = torch.autograd.grad((o * w).sum(), xr, create_graph=True)
        penalty = (g * g).sum()
        wrt = [xr] + params
        vals.append([None if t is None else t.clone() for t in torch.autograd.grad(penalty, wrt, allow_unused=True)])
    for k, (a, b) in enumerate(zip(*vals)):
        if a is None or b is None:
            assert (a is None or float(a.abs().max()) < 1e-12) and (b is None or float(b.abs().max()) < 1e-12), (name, k)
            continue
        assert torch.allclose(a, b, rtol=3e-4, atol=3e-5), (name, k, float((a - b).abs().max()))


def test_bce_over_a_pair_is_the_loss_over_the_concatenation():
    """trainers/cnn.py:124-131: BCE(cat(p_real, p_fake), cat(ones, zeros)), mean over 2B."""
    pr, pf = _t(B, 1, seed=31), _t(B, 1, seed=32)
    targets = torch.cat([torch.ones(B, 1), torch.zeros(B, 1)])
    got = TF.bce_with_logits(TF.Pair(pr, pf), targets)
    want = torch.nn.functional.binary_cross_entropy_with_logits(torch.cat([pr, pf]), targets)
    assert torch.allclose(got, want, rtol=1e-6, atol=1e-7)
    for a, b in zip(torch.autograd.grad(got, [pr, pf]), torch.autograd.grad(want, [pr, pf])):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-7)


# ---- the IQN head's row ops: a Pair keeps each half's rows quantile-major on their own ([half][q][b])
def _rows(seed):
    return _t(B, 5, seed=seed)


@pytest.mark.parametrize('name,fn', [('repeat_rows', lambda x: TF.repeat_rows(x, 3)),
                                     ('repeat+mean', lambda x: TF.mean_reps(TF.tanh(TF.repeat_rows(x, 3)), 3)),
                                     ('mul', lambda x: TF.mul(TF.repeat_rows(x, 2), TF.tanh(TF.repeat_rows(x, 2))))])
def test_paired_row_ops_of_the_iqn_head(name, fn):
    xr, xf = _rows(41), _rows(42)
    out, sep_r, sep_f = fn(TF.Pair(xr, xf)), fn(xr), fn(xf)
    assert torch.allclose(out.r, sep_r, rtol=1e-6, atol=1e-7) and torch.allclose(out.f, sep_f, rtol=1e-6, atol=1e-7)
    assert out.f.data_ptr() == out.r.data_ptr() + out.r.numel() * 4
    wr, wf = torch.randn_like(sep_r), torch.randn_like(sep_f)
    for cr, cf in ((1.0, 1.0), (1.0, 0.0), (0.0, 1.0)):
        got = torch.autograd.grad(sum((o * w).sum() for o, w, c in ((out.r, wr, cr), (out.f, wf, cf)) if c), [xr, xf],
                                  retain_graph=True, allow_unused=True)
        want = torch.autograd.grad(sum((o * w).sum() for o, w, c in ((sep_r, wr, cr), (sep_f, wf, cf)) if c), [xr, xf],
                                   retain_graph=True, allow_unused=True)
        for a, b in zip(got, want):
            assert (a is None and b is None) or torch.allclose(a, b, rtol=1e-5, atol=1e-6), (name, cr, cf)
    # second derivative through the first half (the R1 penalty differentiates the quantile-mean prediction)
    vals = []
    for paired in (True, False):
        o = fn(TF.Pair(xr, xf)).r if paired else fn(xr)
        g, = torch.autograd.grad((o * o).sum(), xr, create_graph=True)
        vals.append(torch.autograd.grad((g * g).sum(), xr)[0])
    assert torch.allclose(vals[0], vals[1], rtol=1e-4, atol=1e-6), name


def test_iqn_loss_over_a_pair_is_the_sum_of_the_two_evaluations():
    """trainers/iqn.py:118-120: loss_real + loss_fake, each over its own quantile-major rows and its own taus."""
    Q = 4
    pr, pf = _t(Q * B, 1, seed=51), _t(Q * B, 1, seed=52)
    tr, tf = torch.rand(Q * B, 1), torch.rand(Q * B, 1)
    targets = torch.cat([torch.ones(B, 1), torch.zeros(B, 1)])
    got = TF.iqn_quantile_huber_loss(TF.Pair(pr, pf), targets, TF.Pair(tr, tf), Q)
    want = TF.iqn_quantile_huber_loss(pr, targets[:B], tr, Q) + TF.iqn_quantile_huber_loss(pf, targets[B:], tf, Q)
    assert torch.allclose(got, want, rtol=1e-6, atol=1e-7)
    for a, b in zip(torch.autograd.grad(got, [pr, pf]), torch.autograd.grad(want, [pr, pf])):
        assert torch.allclose(a, b, rtol=1e-6, atol=1e-8)
    emb = TF.iqn_cos_embed(TF.Pair(tr, tf), torch.arange(1, 7).float())
    assert torch.equal(emb.r, TF.iqn_cos_embed(tr, torch.arange(1, 7).float()))
    assert torch.equal(emb.f, TF.iqn_cos_embed(tf, torch.arange(1, 7).float()))
