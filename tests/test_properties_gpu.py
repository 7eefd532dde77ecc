"""Size-independent properties of the HIP path at the FULL benchmark size (128:3 SA-GAN, 128x128, batch 64).

The CPU oracle needs minutes per step at this size, so here the kernels are checked through identities that hold at
any size, on exactly the operand shapes one full-size training step launches (recorded from a real step):

* every convolution form is bilinear, so forward, input gradient and weight gradient are three views of one trilinear
  form:  <conv(x, w), g> = <x, dgrad(g, w)> = <w, wgrad(x, g)>   (regular 3x3 / 1x1, conv3x3 o up2x, avgpool2 o conv3x3);
* the attention core is linear in the values and its map rows sum to one:  <attn(th, ph, v), g> = <v, d_v>,
  attn(th, ph, 1) = 1;
* training BatchNorm removes the batch mean and variance, and its input gradient is orthogonal to 1 and to x_hat;
* a replayed step is a pure function of (state, inputs): two trainers from one seed stay bit-identical, losses finite;
* averaging the gradients of the two half batches equals the full-batch gradient once the only batch-coupled layer
  (BatchNorm) is the identity -- the property the data-parallel path rests on.
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

FULL = dict(config='128:3', batch=64)


@pytest.fixture(scope='module')
def K():
    from tartangan_amd import backend
    return backend.get()


def _trainer(kind='cnn', batch=FULL['batch'], seed=1234, norm=None):
    from tartangan_amd.models.pluggan import GAN_CONFIGS
    from tartangan_amd.trainers.cnn import CNNTrainer
    from tartangan_amd.trainers.iqn import IQNTrainer
    name, att = FULL['config'].split(':')
    cfg = GAN_CONFIGS[name]._replace(attention=(int(att),))
    cls = {'cnn': CNNTrainer, 'iqn': IQNTrainer}[kind]
    kw = dict(config=cfg, batch_size=batch, device='cuda')
    if norm is not None:
        kw['norm'] = norm
    tr = cls(cls.default_args(**kw))
    torch.manual_seed(seed)
    tr.build_models()
    return tr


def _images(batch, seed=7):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(batch, 3, 128, 128, generator=g) * 2 - 1).cuda()


class _Recorder:
    """Shapes of every launch of the named entry points during one step."""

    def __init__(self, K, names):
        self.K, self.names, self.seen, self._saved = K, names, {}, {}

    def __enter__(self):
        for n in self.names:
            fn = getattr(self.K, n)
            self._saved[n] = fn
            setattr(self.K, n, self._wrap(n, fn))
        return self

    def _wrap(self, name, fn):
        def rec(*args):
            self.seen.setdefault(name, set()).add(tuple(a for a in args if isinstance(a, int)))
            return fn(*args)
        return rec

    def __exit__(self, *exc):
        for n, fn in self._saved.items():
            setattr(self.K, n, fn)


@pytest.fixture(scope='module')
def step_shapes(K):
    tr = _trainer()
    names = ('conv2d_fwd', 'upconv3x3_fwd', 'poolconv3x3_fwd', 'attn_fwd', 'bn_train_fwd')
    with _Recorder(K, names) as r:
        tr.train_batch(_images(FULL['batch']))
    torch.cuda.synchronize()
    assert all(n in r.seen for n in names), sorted(r.seen)
    return r.seen


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator(device='cuda').manual_seed(seed)
    return torch.randn(*shape, generator=g, device='cuda') * scale


def _dot(a, b):
    return float((a.double() * b.double()).sum())


def _same(a, b, tol):
    return abs(a - b) <= tol * max(abs(a), abs(b), 1e-30)


def test_regular_convs_are_one_trilinear_form_at_full_size(K, step_shapes):
    shapes = sorted(step_shapes['conv2d_fwd'])
    # (the discriminator's real | fake pass runs its layers on 2 x batch images)
    assert len(shapes) >= 8 and max(s[0] for s in shapes) == 2 * FULL['batch']
    for (B, Cin, Cout, H, W, ks) in shapes:
        x, w, g = _rand(B, Cin, H, W), _rand(Cout, Cin, ks, ks, seed=1, scale=0.1), _rand(B, Cout, H, W, seed=2)
        y, gx, gw = torch.empty_like(g), torch.empty_like(x), torch.empty_like(w)
        K.conv2d_fwd(x, w, None, None, y, B, Cin, Cout, H, W, ks)
        K.conv2d_dgrad(g, w, gx, B, Cin, Cout, H, W, ks)
        ws = torch.empty(K.conv2d_wgrad_workspace(B, Cin, Cout, H, W, ks) // 4 + 4, device='cuda')
        K.conv2d_wgrad(x, g, gw, None, ws, ws.numel() * 4, B, Cin, Cout, H, W, ks, 0)
        a, b, c = _dot(y, g), _dot(x, gx), _dot(w, gw)
        assert _same(a, b, 2e-5) and _same(a, c, 2e-5), ((B, Cin, Cout, H, W, ks), a, b, c)


@pytest.mark.parametrize('form', ['upconv3x3', 'poolconv3x3'])
def test_stride2_convs_are_one_trilinear_form_at_full_size(K, step_shapes, form):
    shapes = sorted(step_shapes[form + '_fwd'])
    # (both stride-2 forms run on 2 x batch images: the discriminator's real | fake pass, the generator's two forwards)
    assert shapes and max(s[0] for s in shapes) == 2 * FULL['batch']
    F = torch.nn.functional
    for (B, Cin, Cout, H, W) in shapes:          # H x W: the low-resolution plane
        w = _rand(Cout, Cin, 3, 3, seed=1, scale=0.1)
        gw = torch.empty_like(w)
        if form == 'upconv3x3':
            x, g = _rand(B, Cin, H, W), _rand(B, Cout, 2 * H, 2 * W, seed=2)
            y, gx = torch.empty_like(g), torch.empty_like(x)
            wp, w4t = torch.empty(4, Cout, Cin, 2, 2, device='cuda'), torch.empty(Cin, Cout, 4, 4, device='cuda')
            K.upconv3x3_weights(w, wp, Cout, Cin)
            K.upconv3x3_fwd(x, wp, None, None, y, B, Cin, Cout, H, W)
            assert K.upconv3x3_dgrad_supported(B, Cin, Cout, H, W)
            K.upconv3x3_weights_t(w, w4t, Cout, Cin)
            K.upconv3x3_dgrad(g, w4t, gx, B, Cin, Cout, H, W)
            ws = torch.empty(K.upconv3x3_wgrad_workspace(B, Cin, Cout, H, W) // 4 + 4, device='cuda')
            K.upconv3x3_wgrad(x, g, gw, ws, ws.numel() * 4, B, Cin, Cout, H, W, 0, None)
            # and the form itself, on one image, against the textbook composition
            want = F.conv2d(F.interpolate(x[:1], scale_factor=2), w, None, padding=1)
        else:
            x, g = _rand(B, Cin, 2 * H, 2 * W), _rand(B, Cout, H, W, seed=2)
            y, gx = torch.empty_like(g), torch.empty_like(x)
            w4, wp = torch.empty(Cout, Cin, 4, 4, device='cuda'), torch.empty(4, Cin, Cout, 2, 2, device='cuda')
            K.poolconv3x3_weights(w, w4, wp, Cout, Cin)
            K.poolconv3x3_fwd(x, w4, None, None, y, B, Cin, Cout, H, W)
            K.poolconv3x3_dgrad(g, wp, gx, B, Cin, Cout, H, W)
            ws = torch.empty(K.poolconv3x3_wgrad_workspace(B, Cin, Cout, H, W) // 4 + 4, device='cuda')
            K.poolconv3x3_wgrad(x, g, gw, ws, ws.numel() * 4, B, Cin, Cout, H, W, 0, None)
            want = F.avg_pool2d(F.conv2d(x[:1], w, None, padding=1), 2)
        a, b, c = _dot(y, g), _dot(x, gx), _dot(w, gw)
        assert _same(a, b, 2e-5) and _same(a, c, 2e-5), (form, (B, Cin, Cout, H, W), a, b, c)
        assert float((y[:1] - want).abs().max()) <= 2e-4 * float(want.abs().max())


def test_attention_is_linear_in_values_and_rows_sum_to_one_at_full_size(K, step_shapes):
    shapes = sorted(step_shapes['attn_fwd'])
    assert max(s[3] for s in shapes) == 64 * 64          # the generator's 4096 x 1024 map is among them
    for (B, D, DV, N, M) in shapes:
        th, ph, v, g = _rand(B, D, N), _rand(B, D, M, seed=1), _rand(B, DV, M, seed=2), _rand(B, DV, N, seed=3)
        o, lse = torch.empty(B, DV, N, device='cuda'), torch.empty(B, N, device='cuda')
        K.attn_fwd(th, ph, v, o, lse, B, D, DV, N, M)
        dth, dph, dv = torch.empty_like(th), torch.empty_like(ph), torch.empty_like(v)
        ws = torch.empty(K.attn_bwd_workspace(B, D, DV, N, M) // 4 + 4, device='cuda')
        K.attn_bwd(g, th, ph, v, o, lse, dth, dph, dv, ws, B, D, DV, N, M)
        assert _same(_dot(o, g), _dot(v, dv), 2e-5)
        # shifting every score of a row leaves the map unchanged: the query gradient has no component along phi's row sums ...
        # ... and with constant values the output is that constant, the score gradients vanish
        ones = torch.ones_like(v)
        K.attn_fwd(th, ph, ones, o, lse, B, D, DV, N, M)
        assert float((o - 1).abs().max()) <= 1e-5
        K.attn_bwd(g, th, ph, ones, o, lse, dth, dph, dv, ws, B, D, DV, N, M)
        assert float(dth.abs().max()) <= 1e-4 * float(g.abs().max()) and float(dph.abs().max()) <= 1e-3 * float(g.abs().max())


def test_batchnorm_statistics_and_gradient_orthogonality_at_full_size(step_shapes):
    from tartangan_amd import functional as TF
    shapes = sorted(step_shapes['bn_train_fwd'])
    assert max(s[0] * s[2] for s in shapes) >= 64 * 64 * 64
    for dims in shapes:
        B, C, HW = dims[:3]
        side = int(math.isqrt(HW))
        x = (_rand(B, C, side, HW // side) * 3 + 1.5).requires_grad_()
        gamma, beta = torch.ones(C, device='cuda', requires_grad=True), torch.zeros(C, device='cuda', requires_grad=True)
        y = TF.batch_norm_act(x, gamma, beta, None, None, True, 0.1, 1e-5, 1.0)      # slope 1: the affine map alone
        m, v = y.detach().mean((0, 2, 3)), y.detach().var((0, 2, 3), unbiased=False)
        assert float(m.abs().max()) <= 1e-5 and float((v - 1).abs().max()) <= 1e-4
        g = _rand(*y.shape, seed=5)
        gx, = torch.autograd.grad(y, x, g)
        xh = y.detach()
        scale = float(g.abs().mean()) * B * HW
        assert float(gx.sum((0, 2, 3)).abs().max()) <= 2e-5 * scale
        assert float((gx * xh).sum((0, 2, 3)).abs().max()) <= 2e-5 * scale


@pytest.mark.parametrize('kind', ['cnn', 'iqn'])
def test_full_size_step_is_deterministic_and_finite(kind):
    logs = []
    for _ in range(2):
        tr = _trainer(kind)
        tr.enable_graphs()
        torch.manual_seed(99)
        logs.append([tr.train_batch(_images(FULL['batch'], seed=k)) for k in range(3)])
    for a, b in zip(*logs):
        assert a == b, (a, b)
        assert all(math.isfinite(v) for v in a.values()), a


def test_half_batch_gradients_average_to_the_full_batch_gradient():
    """Data parallelism shards the batch and averages gradients: exact (up to summation order) when no layer couples the
    samples, i.e. with the identity norm; checked on the discriminator's hinge-free part of the step at full width."""
    from tartangan_amd import functional as TF
    tr = _trainer(norm='id')
    d = tr.d
    imgs = _images(FULL['batch'])

    def grads(batch):
        for p in d.parameters():
            p.grad = None
        out = d(batch)
        out = out[0] if isinstance(out, (tuple, list)) else out
        with TF.deferred_wgrad():
            out.mean().backward()
        return torch.cat([p.grad.flatten() for p in d.parameters() if p.grad is not None]).clone()

    full = grads(imgs)
    halves = 0.5 * (grads(imgs[:32]) + grads(imgs[32:]))
    assert float((full - halves).abs().max()) <= 2e-5 * float(full.abs().max())
