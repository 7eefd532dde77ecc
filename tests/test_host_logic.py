"""Host logic on CPU: the autograd composition (incl. double backward), the
drop-in modules / state_dict keys / default init, the optimiser and the trainers,
run over the reference-semantics emulator (tests/emulator.py) and compared with
the golden fixtures produced by the reference and with the oracle.

These tests say nothing about the HIP kernels (that is tests/test_*_gpu.py); they
pin everything between the C ABI and the reference's Python API."""
import contextlib

import pytest
import torch

from conftest import golden_cases, load_golden, trainer_from_fixture
from emulator import Emulator
from oracle import sagan_cpu as O
from oracle.procedural import procedural_state, summarize, synthetic_images
from tartangan_amd import backend, functional as TF
from tartangan_amd.models.pluggan import GAN_CONFIGS
from tartangan_amd.trainers.cnn import CNNTrainer
from tartangan_amd.trainers.iqn import IQNTrainer


@pytest.fixture(autouse=True)
def emulated_backend():
    prev = backend._set_backend_for_testing(Emulator())
    yield
    backend._set_backend_for_testing(prev)


def _close(a, b, rel=1e-4, abs_=1e-6):
    return abs(a - b) <= abs_ + rel * max(abs(a), abs(b))


def make_trainer(fx, seed=0):
    return trainer_from_fixture(fx, 'cpu', seed)


def _total_l2(module, grads=False):
    s = 0.
    for p in module.parameters():
        t = p.grad if grads else p
        s += float(t.detach().double().pow(2).sum())
    return s ** 0.5


FAST = [c for c in golden_cases() if not c.endswith(('b64', 'b256')) and not c.startswith('c128') and c != 'c256a3_cnn_b8']


@pytest.mark.parametrize('case', FAST)
def test_trainer_matches_reference_fixture(case, single_thread):
    fx = load_golden(case)
    tr = make_trainer(fx)
    assert list(tr.g.state_dict().keys()) == fx['state_keys']['g']
    assert list(tr.d.state_dict().keys()) == fx['state_keys']['d']
    di = fx['default_init']
    assert _close(_total_l2(tr.g), di['g_l2'], 1e-6)
    assert _close(_total_l2(tr.target_g), di['target_g_l2'], 1e-6)
    assert _close(_total_l2(tr.d), di['d_l2'], 1e-6)
    tr.g.load_state_dict(procedural_state(tr.g.state_dict(), fx['weight_seed']))
    tr.target_g.load_state_dict(procedural_state(tr.target_g.state_dict(), fx['weight_seed'] + 1))
    tr.d.load_state_dict(procedural_state(tr.d.state_dict(), fx['weight_seed'] + 2))
    torch.manual_seed(fx['rng_seed'])
    for k, ref in enumerate(fx['steps']):
        logs = tr.train_batch(synthetic_images(fx['batch'], fx['size'], fx['img_seed'] + k))
        # Tolerances: step 1 from identical state is the parity statement (1e-4 rel).
        # Later steps amplify rounding chaotically -- the reference itself, re-run with 8 instead of
        # 1 CPU threads, moves d_loss by up to 4.5e-3 and the G-grad norm by up to 6e-2 by steps 2-3
        # (DESIGN.md "Chaotic divergence") -- so they are only sanity-bounded.
        loss_tol, grad_tol = (1e-4, 1e-3) if k == 0 else (1e-1, 1.0)
        for name in ('g_loss', 'd_loss', 'gp'):
            assert _close(logs[name], ref[name], loss_tol), (case, k, name, logs[name], ref[name])
        assert _close(_total_l2(tr.g), ref['g_l2'], 1e-4)
        assert _close(_total_l2(tr.d), ref['d_l2'], 1e-4)
        assert _close(_total_l2(tr.target_g), ref['target_g_l2'], 1e-4)
        assert _close(_total_l2(tr.g, True), ref['g_grad_l2'], grad_tol), (case, k)
        assert _close(_total_l2(tr.d, True), ref['d_grad_l2'], grad_tol), (case, k)
        if k == 0:
            for name, p in tr.d.named_parameters():
                ref_s = fx['after_step1']['d_grad'][name]
                got = summarize(p.grad, len(ref_s['idx']))
                # conv biases in front of a BatchNorm have an exactly-zero true gradient: only rounding noise there
                assert _close(got['l2'], ref_s['l2'], 5e-4, 2e-5 * ref['d_grad_l2']), ('d_grad', name, got['l2'], ref_s['l2'])
    assert float(torch.rand(1)) == fx['rng_after']        # same RNG consumption as the reference


def test_flat_parameter_views_survive_load_state_dict():
    from tartangan_amd.optim import param_offsets
    fx = load_golden('c32a2_cnn_b8')
    tr = make_trainer(fx)
    flat = tr.optimizer_d.flat
    tr.d.load_state_dict(procedural_state(tr.d.state_dict(), 3))
    params = list(tr.d.parameters())
    offs, n = param_offsets(params)
    assert n == flat.numel() and any(p.numel() % 4 for p in params)       # (the attention gamma: 1 element)
    for p, off in zip(params, offs):
        assert off % 4 == 0 and p.data_ptr() % 16 == 0                   # every tensor on a 16-byte boundary
        assert p.data_ptr() == flat[off:].data_ptr()
        assert p.grad.data_ptr() == tr.optimizer_d.grads[off:].data_ptr()
    # the padding stays exactly zero through optimiser steps and the EMA
    used = torch.zeros(n, dtype=torch.bool)
    for p, off in zip(params, offs):
        used[off:off + p.numel()] = True
    tr.train_batch(synthetic_images(fx['batch'], fx['size'], 1))
    assert float(tr.optimizer_d.flat[~used].abs().max()) == 0.0 and float(tr.optimizer_d.exp_avg_sq[~used].abs().max()) == 0.0


def test_deferred_wgrad_never_reduces_one_gradient_twice_in_a_launch():
    """The batched reduce runs its items concurrently: the contributions a discriminator weight collects in one backward
    (real batch, fake batch, and the R1 penalty's second-order term) must land in different launches, and the result must
    equal the eager path."""
    fx = load_golden('c32_cnn_b16')

    class Recording(Emulator):
        calls = []

        def conv2d_wgrad_reduce_batch(self, items, n):
            Recording.calls.append(items.clone())
            return super().conv2d_wgrad_reduce_batch(items, n)

    backend._set_backend_for_testing(Recording())
    tr = make_trainer(fx)
    imgs = synthetic_images(fx['batch'], 32, 7)
    torch.manual_seed(1)
    tr._d_phase(imgs)
    # real | fake travel as one batch: a weight collects TWO contributions (the paired pass, the R1 second-order term);
    # with the reference's separate passes it would be three
    assert len(Recording.calls) == 2
    tr3 = make_trainer(fx)
    tr3.args.pair_d = False
    Recording.calls.clear()
    torch.manual_seed(1)
    tr3._d_phase(imgs)
    assert len(Recording.calls) == 3
    for t in Recording.calls:
        dst = t[:, 1].tolist()
        assert len(dst) == len(set(dst))
    deferred = tr.optimizer_d.grads.clone()

    backend._set_backend_for_testing(Emulator())
    tr2 = make_trainer(fx)
    torch.manual_seed(1)
    import contextlib
    orig, TF.deferred_wgrad = TF.deferred_wgrad, contextlib.nullcontext
    try:
        tr2._d_phase(imgs)
    finally:
        TF.deferred_wgrad = orig
    # (on the GPU the two paths are bit-identical, tests/test_kernels_gpu.py; the CPU stand-in's sums differ in the last
    # bit with buffer alignment)
    assert torch.allclose(deferred, tr2.optimizer_d.grads, rtol=1e-6, atol=1e-6)


def test_r1_first_order_pass_skips_parameter_gradients_without_changing_results():
    """models.losses.gradient_penalty runs its autograd.grad under TF.input_grads_only(): same gp, same D gradients,
    but no weight-gradient kernels in that pass."""
    fx = load_golden('c32a2_cnn_b8')
    imgs = synthetic_images(fx['batch'], 32, 7)

    class Counting(Emulator):
        wgrads = 0

        def conv2d_wgrad(self, *a):
            Counting.wgrads += 1
            return super().conv2d_wgrad(*a)

        def conv2d_wgrad_partials(self, *a):
            Counting.wgrads += 1
            return super().conv2d_wgrad_partials(*a)

    def run(hint):
        backend._set_backend_for_testing(Counting())
        Counting.wgrads = 0
        tr = make_trainer(fx)
        torch.manual_seed(1)
        import contextlib
        orig = TF.input_grads_only
        if not hint:
            TF.input_grads_only = contextlib.nullcontext
        try:
            d_loss, gp = tr._d_phase(imgs)
        finally:
            TF.input_grads_only = orig
        return float(d_loss), float(gp), tr.optimizer_d.grads.clone(), Counting.wgrads

    a, b = run(True), run(False)
    assert a[0] == b[0] and a[1] == b[1]
    assert torch.allclose(a[2], b[2], rtol=1e-6, atol=1e-7)
    n_convs = sum(1 for m in make_trainer(fx).d.modules() if type(m).__name__ == 'Conv2d')
    assert b[3] - a[3] == n_convs           # exactly one wasted weight gradient per discriminator conv is gone


@pytest.mark.parametrize('op', ['pool_conv', 'up_conv'])
def test_stride2_functions_match_torch_including_double_backward(op):
    """pool_conv3x3 / upconv3x3 against the reference formulation: values, first-order gradients and, for the pooled conv
    (it sits under the R1 penalty), the gradient of a gradient-norm penalty."""
    torch.manual_seed(3)
    B, Cin, Cout = 2, 5, 7
    x = torch.randn(B, Cin, 16, 16, requires_grad=True)
    w = torch.nn.Parameter(torch.randn(Cout, Cin, 3, 3) * 0.3)
    b = torch.nn.Parameter(torch.randn(Cout) * 0.1)
    F = torch.nn.functional
    if op == 'pool_conv':
        ours = lambda: TF.pool_conv3x3(x, w, b)
        ref = lambda: F.avg_pool2d(F.conv2d(x, w, b, padding=1), 2)
    else:
        ours = lambda: TF.upconv3x3(x, w, b)
        ref = lambda: F.conv2d(F.interpolate(x, scale_factor=2), w, b, padding=1)

    def run(f, second_order):
        for t in (x, w, b):
            t.grad = None
        y = f()
        if second_order:
            g, = torch.autograd.grad(y.tanh().sum(), x, create_graph=True)
            (g.pow(2).sum() + y.sum()).backward()
        else:
            y.pow(2).sum().backward()
        return y.detach(), x.grad.clone(), w.grad.clone(), b.grad.clone()

    for second_order in ((False, True) if op == 'pool_conv' else (False,)):
        got, want = run(ours, second_order), run(ref, second_order)
        for a_, b_ in zip(got, want):
            assert torch.allclose(a_, b_, rtol=2e-4, atol=2e-4), (op, second_order, float((a_ - b_).abs().max()))


def test_attention_core_matches_torch_including_double_backward():
    """The fused attention core under a gradient-norm penalty (its backward is differentiated again, as under R1):
    the hand-derived second-order formulas against autograd on the textbook formulation."""
    torch.manual_seed(5)
    B, D, DV, N, M = 2, 4, 16, 24, 6
    theta, phi, g = (torch.randn(B, d, n, requires_grad=True) for d, n in ((D, N), (D, M), (DV, M)))
    ours = lambda: TF.attention_core(theta, phi, g)
    ref = lambda: torch.bmm(g, torch.softmax(torch.bmm(theta.transpose(1, 2), phi), -1).transpose(1, 2))

    def run(f, wrt):
        for t in (theta, phi, g):
            t.grad = None
        y = f()
        grads = torch.autograd.grad(y.tanh().sum(), wrt, create_graph=True)
        (sum(q.pow(2).sum() for q in grads) + y.pow(2).sum()).backward()
        return [y.detach()] + [t.grad.clone() for t in (theta, phi, g)]

    for wrt in ((theta, phi, g), (theta,), (g,)):          # the last two leave some adjoints of the backward undefined
        for a_, b_ in zip(run(ours, wrt), run(ref, wrt)):
            assert torch.allclose(a_, b_, rtol=2e-4, atol=2e-5), float((a_ - b_).abs().max())


def test_filter_forms_scope_derives_each_filter_once_and_never_serves_a_stale_one():
    calls = []

    class Counting(Emulator):
        def poolconv3x3_weights(self, w, w4, wp, Cout, Cin):
            calls.append(w.data_ptr())
            return super().poolconv3x3_weights(w, w4, wp, Cout, Cin)

    backend._set_backend_for_testing(Counting())
    torch.manual_seed(0)
    x = torch.randn(2, 4, 16, 16)
    w = torch.nn.Parameter(torch.randn(6, 4, 3, 3) * 0.2)
    ref = lambda: torch.nn.functional.avg_pool2d(torch.nn.functional.conv2d(x, w, None, padding=1), 2)
    TF.pool_conv3x3(x, w); TF.pool_conv3x3(x, w)
    assert len(calls) == 2                                   # no scope: derived per use
    with TF.filter_forms():
        a = TF.pool_conv3x3(x, w)
        b = TF.pool_conv3x3(x, w)
        assert len(calls) == 3 and torch.equal(a, b)         # once per scope
        with torch.no_grad():
            w.mul_(2.0)                                      # an in-place update inside the scope: new version, new forms
        c = TF.pool_conv3x3(x, w)
        assert len(calls) == 4
        assert torch.allclose(c, ref(), rtol=1e-4, atol=1e-5)
    TF.pool_conv3x3(x, w)
    assert len(calls) == 5                                   # nothing survives the scope


def test_tall_linear_weight_gradient_is_row_sliced_and_equals_torch():
    """quantiles x batch rows into a small Linear (the IQN head): the weight gradient runs as one batched GEMM over row
    slices + a sum; same numbers as the plain product, with and without an existing .grad to accumulate into."""
    torch.manual_seed(2)
    x = torch.randn(512, 48)
    lin = torch.nn.Linear(48, 24)
    w, b = torch.nn.Parameter(lin.weight.detach().clone()), torch.nn.Parameter(lin.bias.detach().clone())
    for k in range(2):                              # second pass: .grad exists -> accumulate-in-place path (opt-in)
        with (TF.grads_into_buckets() if k else contextlib.nullcontext()):
            TF.linear(x, w, b).tanh().sum().backward()
        torch.nn.functional.linear(x, lin.weight, lin.bias).tanh().sum().backward()
        assert torch.allclose(w.grad, lin.weight.grad, rtol=1e-4, atol=1e-4)
        assert torch.allclose(b.grad, lin.bias.grad, rtol=1e-4, atol=1e-4)


def test_foreign_training_loop_sees_stock_autograd_semantics():
    """INTEGRATION.md section 2 invites foreign training loops on these modules: outside the trainers' own
    ``grads_into_buckets()`` scope a backward must behave like stock modules -- ``torch.autograd.grad`` returns every
    parameter gradient and leaves ``.grad`` alone, ``backward(inputs=subset)`` touches only the subset, tensor hooks
    fire -- even though every parameter already owns a ``.grad`` view into the optimiser's flat bucket."""
    fx = load_golden('c32a2_cnn_b8')
    tr = make_trainer(fx)
    tr.d.load_state_dict(procedural_state(tr.d.state_dict(), 5))
    imgs = synthetic_images(4, 32, 3)
    params = list(tr.d.parameters())
    assert all(p.grad is not None for p in params)           # FusedAdam bound the flat bucket
    tr.optimizer_d.zero_grad()
    loss = TF.bce_with_logits(tr.d(imgs), torch.ones(4, 1))
    got = torch.autograd.grad(loss, params, retain_graph=True)
    assert all(g is not None for g in got)
    assert float(tr.optimizer_d.grads.abs().max()) == 0.0    # .grad untouched
    fired = []
    handle = params[0].register_hook(lambda g: fired.append(g.clone()))
    loss.backward(inputs=params[:3], retain_graph=True)
    handle.remove()
    assert len(fired) == 1 and torch.allclose(fired[0], got[0], rtol=1e-5, atol=1e-7)
    for p, g in zip(params[:3], got[:3]):
        assert torch.allclose(p.grad, g, rtol=1e-5, atol=1e-7)
    for p in params[3:]:
        assert float(p.grad.abs().max()) == 0.0              # not in `inputs`: not written
    tr.optimizer_d.zero_grad()
    with TF.deferred_wgrad():                                 # the trainers' path: straight into the bucket
        loss.backward()
    for p, g in zip(params, got):
        assert torch.allclose(p.grad, g, rtol=1e-4, atol=1e-6), (p.shape,)


def test_hinge_losses_match_the_reference_formulas():
    from tartangan_amd.models.losses import discriminator_hinge_loss, generator_hinge_loss
    torch.manual_seed(4)
    real, fake = torch.randn(9, 1, requires_grad=True), torch.randn(9, 1, requires_grad=True)
    lr, lf = discriminator_hinge_loss(real, fake)
    lg = generator_hinge_loss(fake)
    want = (torch.relu(1. - real).mean(), torch.relu(1. + fake).mean(), -fake.mean())       # losses.py:7-14
    for got, w in zip((lr, lf, lg), want):
        assert torch.allclose(got, w.detach(), rtol=1e-6, atol=1e-7)
    (lr + lf + lg).backward()
    g_real, g_fake = real.grad.clone(), fake.grad.clone()
    real.grad = fake.grad = None
    (want[0] + want[1] + want[2]).backward()
    assert torch.allclose(g_real, real.grad, atol=1e-7) and torch.allclose(g_fake, fake.grad, atol=1e-7)


def test_product_has_no_cpu_fallback():
    backend._set_backend_for_testing(None)
    x = torch.zeros(1, 4, 4, 4)
    w = torch.zeros(4, 4, 3, 3)
    with pytest.raises(RuntimeError):
        TF.conv2d(x, w, None)


@pytest.mark.parametrize('kind,flags', [('cnn', {}), ('iqn', {}), ('cnn', {'norm': 'id'}), ('cnn', {'activation': 'selu'})])
def test_paired_discriminator_pass_equals_two_separate_passes(kind, flags, single_thread):
    """D(real) || D(fake) as ONE pass over 2B images (functional.Pair, grouped BatchNorm) against the reference's two
    forwards: same losses over two steps, same parameters and BatchNorm buffers after the first (running statistics updated
    real-then-fake, num_batches_tracked += 2 per layer and D phase), same RNG consumption."""
    cls = {'cnn': CNNTrainer, 'iqn': IQNTrainer}[kind]
    attention = () if flags.get('activation') == 'selu' else (1,)       # (the reference's selu init rejects the 0-d gamma)
    cfg = GAN_CONFIGS['32']._replace(attention=attention)
    results = []
    for pair in (True, False):
        tr = cls(cls.default_args(config=cfg, batch_size=6, device='cpu', pair_d=pair, **flags))
        torch.manual_seed(0)
        tr.build_models()
        tr.g.load_state_dict(procedural_state(tr.g.state_dict(), 7))
        tr.d.load_state_dict(procedural_state(tr.d.state_dict(), 9))
        assert tr._d_pairable() == pair
        imgs = synthetic_images(6, 32, 99)
        torch.manual_seed(5)
        logs = [tr.train_batch(imgs)]
        snap = ({k: v.clone() for k, v in tr.d.state_dict().items()}, {k: v.clone() for k, v in tr.g.state_dict().items()})
        logs.append(tr.train_batch(imgs))           # (step 2: losses only -- the noise biases below feed the next statistics)
        results.append((logs, snap[0], snap[1], float(torch.rand(1))))
    (logs_p, d_p, g_p, rng_p), (logs_s, d_s, g_s, rng_s) = results
    assert rng_p == rng_s
    for step, (a, b) in enumerate(zip(logs_p, logs_s)):
        for k in a:          # (step 2 starts from parameters that already differ by Adam's sign noise, see below)
            assert _close(a[k], b[k], 2e-5 if step == 0 else 2e-3), (step, k, a, b)
    # Adam(beta1=0) moves every element by ~lr * sign(gradient) at step 1: an element whose (tiny) gradient flips sign in
    # the rounding noise differs by 2 lr -- the bound here; the gradients themselves are compared tightly below
    for name, lr, (p, q) in [('d', 4e-4, (d_p, d_s)), ('g', 1e-4, (g_p, g_s))]:
        for k in p:
            if k.endswith('num_batches_tracked'):
                assert int(p[k]) == int(q[k]), k
            else:
                assert float((p[k] - q[k]).abs().max()) <= 2.1 * lr + 1e-6, (name, k, float((p[k] - q[k]).abs().max()))


def test_paired_d_phase_leaves_the_batchnorm_buffers_of_two_forwards():
    """After the D phase alone (no optimiser step in between): the parameter gradients, and running statistics and
    num_batches_tracked of every BatchNorm of D, equal those of the reference's real-then-fake forwards, tightly."""
    cfg = GAN_CONFIGS['32']._replace(attention=(1,))
    states, grads = [], []
    for pair in (True, False):
        tr = CNNTrainer(CNNTrainer.default_args(config=cfg, batch_size=6, device='cpu', pair_d=pair))
        torch.manual_seed(0)
        tr.build_models()
        tr.d.load_state_dict(procedural_state(tr.d.state_dict(), 9))
        torch.manual_seed(5)
        tr.g.train(); tr.d.train()
        tr._d_phase(synthetic_images(6, 32, 99))
        states.append({k: v.clone() for k, v in tr.d.state_dict().items() if 'running' in k or 'num_batches' in k})
        grads.append({n: p.grad.clone() for n, p in tr.d.named_parameters()})
    for n in grads[0]:          # ... and the same parameter gradients (first order of both halves + the R1 second-order terms)
        scale = max(float(grads[1][n].abs().max()), 1e-3)
        assert float((grads[0][n] - grads[1][n]).abs().max()) <= 2e-5 * scale + 2e-5, n
    assert states[0].keys() == states[1].keys() and len(states[0]) > 10
    for k in states[0]:
        if k.endswith('num_batches_tracked'):
            assert int(states[0][k]) == int(states[1][k]) == 2, k
        else:
            assert torch.allclose(states[0][k], states[1][k], rtol=1e-6, atol=1e-7), k


def test_paired_pass_runs_the_r1_penalty_on_the_real_half_only():
    """Launch accounting on the emulator: with the pair, a D phase runs each forward conv ONCE (on 2B images), the R1
    first-order pass and its second-order sweep on B images, and the final backward once on 2B."""
    cfg = GAN_CONFIGS['32']._replace(attention=())
    calls = []

    class Counting(Emulator):
        def conv2d_fwd(self, x, w, bias, residual, y, B, *rest):
            calls.append(('fwd', B))
            return super().conv2d_fwd(x, w, bias, residual, y, B, *rest)

        def conv2d_dgrad(self, gy, w, gx, B, *rest):
            calls.append(('dgrad', B))
            return super().conv2d_dgrad(gy, w, gx, B, *rest)

    prev = backend._set_backend_for_testing(Counting())
    try:
        tr = CNNTrainer(CNNTrainer.default_args(config=cfg, batch_size=4, device='cpu'))
        torch.manual_seed(0)
        tr.build_models()
        calls.clear()
        tr._d_phase(synthetic_images(4, 32, 1))
    finally:
        backend._set_backend_for_testing(prev)
    sizes = {k: sorted({b for kk, b in calls if kk == k}) for k in ('fwd', 'dgrad')}
    assert sizes['fwd'] == [4, 8] and sizes['dgrad'] == [4, 8], sizes        # 4: generator + R1 passes; 8: the paired D passes


def test_final_d_backward_skips_the_input_gradient_of_the_first_layers():
    """``params_only``: the final backward of the D phase asks for parameter gradients only; the from-RGB layers (whose
    input is pure data) must not run their input-gradient convolutions there -- the R1 first-order pass (needs d/d real) and
    the generator phase (needs d/d fake) still do.  Same gradients either way."""
    cfg = GAN_CONFIGS['32']._replace(attention=())
    calls = []

    class Counting(Emulator):
        def conv2d_dgrad(self, gy, w, gx, B, Cin, Cout, H, W, ks):
            calls.append((B, Cin, Cout, H, ks))
            return super().conv2d_dgrad(gy, w, gx, B, Cin, Cout, H, W, ks)

    prev = backend._set_backend_for_testing(Counting())
    try:
        grads = []
        for hint in (True, False):
            tr = CNNTrainer(CNNTrainer.default_args(config=cfg, batch_size=4, device='cpu'))
            torch.manual_seed(0)
            tr.build_models()
            calls.clear()
            orig = TF.params_only
            if not hint:
                TF.params_only = contextlib.nullcontext
            try:
                torch.manual_seed(1)
                tr._d_phase(synthetic_images(4, 32, 1))
            finally:
                TF.params_only = orig
            first = [c for c in calls if c[3] == 32 and c[1] in (3, 4)]       # input gradients w.r.t. [img, 1] / img at 32 x 32
            assert sorted(c[0] for c in first) == ([4] if hint else [4, 8]), first     # R1 pass only / + the wasted 2B one
            grads.append(tr.optimizer_d.grads.clone())
            if hint:
                calls.clear()
                tr._g_phase(4)
                assert any(c[3] == 32 and c[1] in (3, 4) for c in calls)          # the G phase needs d/d fake
        assert torch.equal(grads[0], grads[1])
    finally:
        backend._set_backend_for_testing(prev)


@pytest.mark.parametrize('kind', ['cnn', 'iqn'])
def test_generator_forwards_of_both_phases_share_one_pass(kind, single_thread):
    """sample_g of the D phase and of the G phase (same weights, different latents) as ONE pass over 2B latents: same
    losses / gradients / RNG consumption as two passes, and the generator's BatchNorm buffers see the D-phase half first."""
    cls = {'cnn': CNNTrainer, 'iqn': IQNTrainer}[kind]
    cfg = GAN_CONFIGS['32']._replace(attention=(1,))
    calls = []

    class Counting(Emulator):
        def conv2d_fwd(self, x, w, bias, residual, y, B, Cin, Cout, H, W, ks):
            if Cout == 3:
                calls.append(B)                         # the to-RGB convolution: once per generator pass
            return super().conv2d_fwd(x, w, bias, residual, y, B, Cin, Cout, H, W, ks)

    prev = backend._set_backend_for_testing(Counting())
    try:
        res = []
        for pair in (True, False):
            tr = cls(cls.default_args(config=cfg, batch_size=6, device='cpu', pair_g=pair))
            torch.manual_seed(0)
            tr.build_models()
            tr.g.load_state_dict(procedural_state(tr.g.state_dict(), 7))
            tr.d.load_state_dict(procedural_state(tr.d.state_dict(), 9))
            calls.clear()
            torch.manual_seed(5)
            logs = tr.train_batch(synthetic_images(6, 32, 99))
            assert calls == ([12] if pair else [6, 6]), calls
            res.append((logs, {k: v.clone() for k, v in tr.g.state_dict().items()}, float(torch.rand(1))))
    finally:
        backend._set_backend_for_testing(prev)
    (lp, gp_, rp), (ls, gs, rs) = res
    assert rp == rs
    for k in lp:
        assert _close(lp[k], ls[k], 2e-5), (k, lp, ls)
    for k in gp_:
        if k.endswith('num_batches_tracked'):
            assert int(gp_[k]) == int(gs[k]) == 2, k
        elif k.endswith(('running_mean', 'running_var')):
            assert torch.allclose(gp_[k], gs[k], rtol=1e-5, atol=1e-6), k
        else:
            assert float((gp_[k] - gs[k]).abs().max()) <= 2.1 * 1e-4 + 1e-6, k          # Adam's +-lr sign noise (see above)


@pytest.mark.parametrize('kind', ['cnn', 'iqn'])
def test_fused_attention_projections_equal_three_separate_convolutions(kind, single_thread):
    """theta | phi | g of SelfAttention2d as one pass (forward, joint input gradient, filter gradients, and the R1 second-order
    terms through them) against the three separate 1x1 convolutions: same losses, same parameter gradients."""
    from tartangan_amd.models.blocks import SelfAttention2d
    cls = {'cnn': CNNTrainer, 'iqn': IQNTrainer}[kind]
    cfg = GAN_CONFIGS['32']._replace(attention=(1,))
    res = []
    try:
        for fused in (True, False):
            SelfAttention2d.fuse_projections = fused
            tr = cls(cls.default_args(config=cfg, batch_size=6, device='cpu'))
            torch.manual_seed(0)
            tr.build_models()
            tr.g.load_state_dict(procedural_state(tr.g.state_dict(), 7))
            tr.d.load_state_dict(procedural_state(tr.d.state_dict(), 9))
            with torch.no_grad():                                   # gamma = 0 would switch the attention branch off
                for net in (tr.g, tr.d):
                    for m in net.modules():
                        if isinstance(m, SelfAttention2d):
                            m.gamma.fill_(0.7)
            tr.g.train(); tr.d.train()
            torch.manual_seed(5)
            d_loss, gp = tr._d_phase(synthetic_images(6, 32, 99))
            d_grads = {n: p.grad.clone() for n, p in tr.d.named_parameters()}
            g_loss = tr._g_phase(6)
            g_grads = {n: p.grad.clone() for n, p in tr.g.named_parameters()}
            res.append((float(d_loss), float(gp), float(g_loss), d_grads, g_grads))
    finally:
        SelfAttention2d.fuse_projections = True
    a, b = res
    for k in range(3):
        assert _close(a[k], b[k], 2e-5), (k, a[k], b[k])
    for grads_a, grads_b in ((a[3], b[3]), (a[4], b[4])):
        for n in grads_a:
            scale = max(float(grads_b[n].abs().max()), 1e-3)
            assert float((grads_a[n] - grads_b[n]).abs().max()) <= 3e-5 * scale + 2e-5, n


def test_speculative_rng_prefetch_is_invisible():
    """RngFeed.prefetch draws the next step's inputs early and puts the generator back: a run whose steps are separated by
    nothing, by a foreign draw, or by a re-seed consumes the default generator exactly like a run that never speculates."""
    cfg = GAN_CONFIGS['32']._replace(attention=())
    imgs = synthetic_images(4, 32, 3)

    def run(prefetch):
        tr = IQNTrainer(IQNTrainer.default_args(config=cfg, batch_size=4, device='cpu', prefetch_rng=prefetch))
        torch.manual_seed(0)
        tr.build_models()
        torch.manual_seed(21)
        out = [tr.train_batch(imgs), tr.train_batch(imgs)]        # back to back: the speculation is adopted
        assert (tr.rng_feed._spec is not None) == prefetch
        out.append(float(torch.randn(3).sum()))                    # a foreign draw between steps sees the untouched stream ...
        out.append(tr.train_batch(imgs))                           # ... and the step after it draws afresh
        torch.manual_seed(77)                                      # a re-seed between steps
        out.append(tr.train_batch(imgs))
        out.append(tr.train_batch(imgs))
        out.append(float(torch.rand(1)))
        return out

    assert run(True) == run(False)


def test_adopted_rng_plan_lives_in_one_arena_with_the_latents_back_to_back():
    """RngFeed.adopt: one device arena behind the whole plan (a single upload per step), the latents adjacent whatever is
    drawn between them (the shared generator pass joins them without a copy), and the values still the reference's stream in
    draw order (trainer.py:153-156, iqn.py:105-108)."""
    from tartangan_amd.trainers.trainer import RngFeed
    feed = RngFeed('cpu')
    plan = [('z', 6, 8), ('tau', 6 * 3, 3), ('tau', 6 * 3, 3), ('z', 6, 8), ('tau', 6 * 3, 3)]
    feed.adopt(plan)
    z_d, z_g = feed.static[0], feed.static[3]
    joined = TF._join(z_d, z_g)
    assert joined.data_ptr() == z_d.data_ptr() and joined.shape == (12, 8)
    assert all(t.untyped_storage().data_ptr() == feed._arena.untyped_storage().data_ptr() for t in feed.static)
    assert all(t.data_ptr() % 16 == 0 for t in feed.static)
    torch.manual_seed(11)
    feed.refill()
    torch.manual_seed(11)
    for (kind, rows, cols), got in zip(plan, feed.static):
        want = torch.randn(rows, cols) if kind == 'z' else torch.rand(rows, 1)
        assert torch.equal(got, want)
    # slices do not overlap: writing one leaves the others alone
    before = [t.clone() for t in feed.static]
    feed.static[1].fill_(7.)
    assert all(torch.equal(a, b) for k, (a, b) in enumerate(zip(before, feed.static)) if k != 1)


def test_filter_forms_scope_for_a_module_derives_its_known_filters_in_one_launch():
    """filter_forms(owner): the first scope learns which filters the module's pass derives; every later scope derives them all up
    front with ONE batched call (fresh values each time: the parameters moved in between), and the results equal the per-layer path."""
    single, batch = [], []

    class Counting(Emulator):
        def poolconv3x3_weights(self, w, w4, wp, Cout, Cin):
            single.append(w.data_ptr())
            return super().poolconv3x3_weights(w, w4, wp, Cout, Cin)

        def poolconv3x3_weights_batch(self, items, n_items):
            batch.append(n_items)
            n = len(single)
            rc = super().poolconv3x3_weights_batch(items, n_items)
            del single[n:]                                   # (the emulator's batch entry goes through the single one)
            return rc

    backend._set_backend_for_testing(Counting())
    torch.manual_seed(0)
    owner = torch.nn.Module()
    ws = [torch.nn.Parameter(torch.randn(6, 4, 3, 3) * 0.2), torch.nn.Parameter(torch.randn(5, 4, 3, 3) * 0.2),
          torch.nn.Parameter(torch.randn(8, 4, 3, 3) * 0.2)]
    x = torch.randn(2, 4, 16, 16)
    ref = lambda w: torch.nn.functional.avg_pool2d(torch.nn.functional.conv2d(x, w, None, padding=1), 2)
    with TF.filter_forms(owner):                              # learning pass: per layer, as they come
        for w in ws:
            TF.pool_conv3x3(x, w)
    assert len(single) == 3 and batch == []
    for step in range(2):
        with torch.no_grad():
            for w in ws:
                w.add_(0.01)                                  # (an optimiser step between two passes)
        with TF.filter_forms(owner):
            outs = [TF.pool_conv3x3(x, w) for w in ws]
            outs2 = [TF.pool_conv3x3(x, w) for w in ws]
        assert len(single) == 3 and batch == [3] * (step + 1)  # one batched derivation, nothing per layer
        for o, o2, w in zip(outs, outs2, ws):
            assert torch.equal(o, o2) and torch.allclose(o, ref(w), rtol=1e-4, atol=1e-5)
    del ws[1], w, outs, outs2, o, o2                          # a filter that no longer exists drops out of the list
    import gc
    gc.collect()
    with TF.filter_forms(owner):
        pass
    assert batch[-1] == 2


def test_composed_from_rgb_filter_and_its_gradients():
    """conv3x3(conv1x1(img) + b1) == conv3x3([img, 1], compose_rgb_filter(...)) with zero padding, and the three parameter gradients
    equal autograd's through the two-layer formulation -- returned as tensors, or accumulated straight into ``.grad``."""
    torch.manual_seed(0)
    C, Cimg, Cout = 6, 3, 5
    w1 = torch.nn.Parameter(torch.randn(C, Cimg, 1, 1) * 0.5)
    b1 = torch.nn.Parameter(torch.randn(C) * 0.5)
    w3 = torch.nn.Parameter(torch.randn(Cout, C, 3, 3) * 0.3)
    img = torch.randn(2, Cimg, 8, 8)
    F = torch.nn.functional
    ref = F.conv2d(F.conv2d(img, w1, b1), w3, None, padding=1)
    wc = TF.compose_rgb_filter(w1, b1, w3)
    got = F.conv2d(torch.cat([img, torch.ones(2, 1, 8, 8)], 1), wc, None, padding=1)
    assert torch.allclose(got, ref, rtol=1e-5, atol=1e-5)
    cot = torch.randn_like(ref)
    want = torch.autograd.grad((ref * cot).sum(), [w1, b1, w3])
    have = torch.autograd.grad((got * cot).sum(), [w1, b1, w3])
    for a, b in zip(have, want):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-5)
    # accumulated in place inside the trainers' backward context
    for p in (w1, b1, w3):
        p.grad = torch.ones_like(p)
    wc = TF.compose_rgb_filter(w1, b1, w3)
    loss = (F.conv2d(torch.cat([img, torch.ones(2, 1, 8, 8)], 1), wc, None, padding=1) * cot).sum()
    with TF.grads_into_buckets():
        loss.backward()
    for p, b in zip((w1, b1, w3), want):
        assert torch.allclose(p.grad, 1 + b, rtol=1e-4, atol=1e-5)
