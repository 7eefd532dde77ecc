"""Autograd glue over the C ABI (include/tartangan_amd.h).

Every differentiable op of the SA-GAN / SA-GAN-IQN step is a
``torch.autograd.Function`` whose ``backward`` is itself written with
``Function.apply`` calls, so the R1 penalty's ``autograd.grad(...,
create_graph=True)`` followed by ``d_loss.backward()`` (reference
models/losses.py:17-30, trainers/cnn.py:133-136) differentiates twice through
hand-written HIP kernels only.  PyTorch supplies tensor storage and graph
bookkeeping, nothing else.

Linear ops come as transpose pairs (each is the other's backward):
  conv2d fwd <-> dgrad, (wgrad bilinear);  up2x <-> pool2;  bilinear_half <-> its
  transpose;  maxpool scatter <-> gather;  row_sum <-> row_bcast;
  repeat_rows <-> sum_reps;  channel_sum <-> channel_bcast.
"""
import contextlib
import weakref

import torch
from torch.autograd.function import once_differentiable

from . import backend as _be


class Function(torch.autograd.Function):
    """``torch.autograd.Function`` that never works on gradients nobody sent.

    By default autograd materialises a missing output gradient as zeros and runs ``backward`` on it.  With the real and
    the fake batch of the discriminator in one graph (``Pair``), the R1 penalty's ``autograd.grad(p_real.sum(), real)``
    reaches every node of the FAKE half that feeds a shared node with an undefined gradient -- zeros would then be pushed
    through its kernels and arrive at the shared nodes looking like a gradient for both halves.  Every Function of this
    module therefore opts out of materialisation: ``backward`` is skipped (all-None result) when no output received a
    gradient, and a partially missing set is zero-filled only for the Functions that do not handle None themselves
    (``handles_none_grads``)."""
    handles_none_grads = False

    def __init_subclass__(cls, **kw):
        super().__init_subclass__(**kw)
        fwd, bwd = cls.__dict__.get('forward'), cls.__dict__.get('backward')
        if isinstance(fwd, staticmethod) and isinstance(bwd, staticmethod):
            cls.forward = staticmethod(_wrap_forward(cls, fwd.__func__))
            cls.backward = staticmethod(_wrap_backward(cls, bwd.__func__))


def _wrap_forward(cls, fwd):
    def forward(ctx, *args):
        out = fwd(ctx, *args)
        ctx.set_materialize_grads(False)
        ctx._n_inputs = len(args)
        if not cls.handles_none_grads:
            outs = out if isinstance(out, tuple) else (out,)
            ctx._out_meta = [(tuple(o.shape), o.dtype, o.device) if torch.is_tensor(o) else None for o in outs]
        return out
    forward.__doc__, forward.__name__ = fwd.__doc__, fwd.__name__
    return forward


def _wrap_backward(cls, bwd):
    def backward(ctx, *grads):
        if all(g is None for g in grads):
            return (None,) * ctx._n_inputs
        if not cls.handles_none_grads and any(g is None for g in grads):
            grads = tuple(torch.zeros(m[0], dtype=m[1], device=m[2]) if (g is None and m is not None) else g
                          for g, m in zip(grads, ctx._out_meta))
        return bwd(ctx, *grads)
    backward.__doc__, backward.__name__ = bwd.__doc__, bwd.__name__
    return backward


def K():
    return _be.get()


def _ws(like, nbytes):
    return torch.empty(max(1, (int(nbytes) + 3) // 4), dtype=torch.float32, device=like.device)


class _GradSinks:
    active = False


@contextlib.contextmanager
def grads_into_buckets():
    """Opt-in for the direct parameter-gradient path (see ``_grad_sink``), entered by the trainers around their own
    ``backward`` calls only.  Outside it every Function returns its parameter gradients to autograd like any other op,
    so ``torch.autograd.grad(loss, params)``, ``backward(inputs=subset)``, tensor hooks and AccumulateGrad post-hooks
    (DDP, gradient clipping hooks) of a foreign training loop behave as with stock modules."""
    prev = _GradSinks.active
    _GradSinks.active = True
    try:
        yield
    finally:
        _GradSinks.active = prev


def _grad_sink(p):
    """Where a parameter gradient can be accumulated in place, or None.

    Inside ``grads_into_buckets()``, in a plain ``.backward()`` (no create_graph), the gradient of a leaf
    parameter that already owns a ``.grad`` buffer (tartangan_amd.optim keeps them as views of one flat
    bucket per network) is written by the producing kernel straight into that buffer with accumulate
    semantics, and the Function returns ``None`` for it.  That is exactly what autograd's AccumulateGrad
    would do with a returned tensor, minus one temporary and one elementwise-add launch per parameter.
    """
    if not _GradSinks.active:
        return None
    if p is None or torch.is_grad_enabled() or not (p.is_leaf and p.requires_grad):
        return None
    g = p.grad
    if g is None or not g.is_contiguous() or g.dtype != torch.float32:
        return None
    return g


class _InputGradsOnly:
    active = False


@contextlib.contextmanager
def input_grads_only():
    """Tell the layer Functions that the enclosing ``torch.autograd.grad`` call asks for gradients of DATA only.

    A custom Function cannot see which of its inputs a particular backward call actually needs
    (``ctx.needs_input_grad`` is fixed at forward time), so without this hint the R1 penalty's first-order pass
    (``autograd.grad(outputs=d_real.sum(), inputs=real)``, reference models/losses.py:23-26) would also compute --
    and throw away -- every weight and bias gradient of the discriminator: a fifth of all weight-gradient work
    of a training step.  Used by ``models.losses.gradient_penalty`` only.
    """
    prev = _InputGradsOnly.active
    _InputGradsOnly.active = True
    try:
        yield
    finally:
        _InputGradsOnly.active = prev


def _param_grads_wanted():
    return not _InputGradsOnly.active


class _ParamsOnly:
    active = False


@contextlib.contextmanager
def params_only():
    """The mirror image of ``input_grads_only``: the enclosing backward asks for PARAMETER gradients only
    (``torch.autograd.backward(d_loss, inputs=d.parameters())``).  ``needs_input_grad`` of the first layers still says
    "the images want a gradient" (they did, for the R1 pass), so without this hint the final backward of the D phase would
    push the gradient all the way down to the images -- an input-gradient convolution at the full resolution for
    both halves of the batch, thrown away.  Layers whose input is pure DATA (a leaf that is not a parameter, or something
    computed from such leaves without any parameter: ``mark_data``) skip their input gradient inside this context."""
    prev = _ParamsOnly.active
    _ParamsOnly.active = True
    try:
        yield
    finally:
        _ParamsOnly.active = prev


def is_data(x):
    """Is ``x`` (tensor or Pair) free of parameters upstream, as far as this module can tell?"""
    if isinstance(x, Pair):
        return is_data(x.r) and is_data(x.f)
    if x is None:
        return True
    return bool(getattr(x, '_tg_data', False)) or (x.is_leaf and not isinstance(x, torch.nn.Parameter))


def mark_data(out, *sources):
    """Tag ``out`` (tensor / Pair / tuple of them) as data when every source is."""
    if all(is_data(s) for s in sources):
        for o in (out if isinstance(out, tuple) else (out,)):
            for t in ((o.r, o.f) if isinstance(o, Pair) else (o,)):
                t._tg_data = True
    return out


def _skip_input_grad(ctx):
    return _ParamsOnly.active and getattr(ctx, 'data_input', False)


# =========================================================================== real | fake pairs
class Pair:
    """The real and the fake batch of a discriminator step travelling through the network as ONE tensor of 2B images.

    The reference evaluates D twice with the same weights (trainers/cnn.py:122-123, iqn.py:118-119).  Convolutions,
    resampling, attention and the heads are per-image maps, so both evaluations can share every kernel launch -- twice the
    grid on the small-plane layers that underfill the chip, half the launches on the launch-bound ones; BatchNorm keeps
    per-half statistics (tg_bn_*_groups).  For autograd the halves stay SEPARATE tensors (``r``, ``f``: views of one
    (2B, ...) buffer): the R1 penalty differentiates the real half alone (its first-order pass and the second-order sweep
    run on B images exactly as before), while the final backward, which reaches a layer with both halves' gradients, runs
    each backward kernel once on 2B.  See ``_Paired``."""
    __slots__ = ('r', 'f')

    def __init__(self, r, f):
        if r.shape != f.shape or r.dtype != f.dtype or r.device != f.device:
            raise ValueError(f'Pair: halves differ ({tuple(r.shape)} {r.dtype} vs {tuple(f.shape)} {f.dtype})')
        self.r, self.f = r, f

    @property
    def shape(self):
        return torch.Size((2 * self.r.shape[0],) + tuple(self.r.shape[1:]))

    def size(self, i=None):
        return self.shape if i is None else self.shape[i]

    def dim(self):
        return self.r.dim()

    def __len__(self):
        return 2 * self.r.shape[0]

    @property
    def requires_grad(self):
        return self.r.requires_grad or self.f.requires_grad

    @property
    def device(self):
        return self.r.device

    def view(self, *shape):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)):
            shape = tuple(shape[0])
        if shape[0] != -1:
            if shape[0] != 2 * self.r.shape[0]:
                raise ValueError('Pair.view: the leading (image) dimension must stay')
            shape = (self.r.shape[0],) + tuple(shape[1:])
        return Pair(self.r.view(*shape), self.f.view(*shape))


def _join(r, f):
    """One (2B, ...) tensor over both halves: an alias when they lie back to back in one storage (what every paired op
    returns), a copy otherwise (the network input; a gradient autograd had to sum out of place)."""
    if (r.numel() and r.is_contiguous() and f.is_contiguous()
            and r.untyped_storage().data_ptr() == f.untyped_storage().data_ptr()
            and f.storage_offset() == r.storage_offset() + r.numel()):
        return r.as_strided((2 * r.shape[0],) + tuple(r.shape[1:]), r.stride(), r.storage_offset())
    out = r.new_empty((2 * r.shape[0],) + tuple(r.shape[1:]))
    out[:r.shape[0]].copy_(r)
    out[r.shape[0]:].copy_(f)
    return out


class _PairCtx:
    """What a Function's ``forward`` / ``backward`` see in place of the autograd ctx when ``_Paired`` runs them."""

    def __init__(self):
        self.saved_tensors = ()
        self.materialize = True
        self.needs_input_grad = ()

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors

    def set_materialize_grads(self, flag):
        self.materialize = bool(flag)

    def mark_non_differentiable(self, *tensors):
        pass


# Function -> (forward arguments: 'b' a batch tensor, '-' anything else;
#              saved tensors:     'h' halve along dim 0 for the real-half view, '-' keep;
#              outputs:           'b' batch tensor, '-' other; '*' = all batch)
PAIR_SPECS = {}


class _Paired(Function):
    """Run Function ``F`` on the two halves of a ``Pair`` as one batch.

    forward : the halves are joined (an alias, see ``_join``), ``F.forward`` runs ONCE on 2B images, the outputs are handed
              back as two views.  Autograd sees one node with inputs (.., x_r, x_f, ..) and outputs (y_r, y_f).
    backward: gradients for BOTH halves (the final backward of the D phase) -> joined, ``F.backward`` runs once on 2B;
              a gradient for ONE half only (the R1 penalty's ``autograd.grad(p_real.sum(), real, create_graph=True)``
              reaches the real half alone; of the generator's two forwards of a step only the G phase's is
              differentiated) -> ``F.backward`` runs on that half's views of the saved tensors, building exactly the
              graph the unpaired pass would build (B images; the R1 second-order sweep is untouched).
    ``F``'s own forward / backward are reused unchanged through a stand-in ctx (``_PairCtx``)."""
    handles_none_grads = True

    @staticmethod
    def forward(ctx, F, *args):
        layout, _, out_spec = PAIR_SPECS[F]
        fargs, pos, reals, fakes, joined = [], 0, [], [], {}
        for kind in layout:
            if kind == 'b':
                r, f = args[pos], args[pos + 1]
                pos += 2
                if r is None:
                    fargs.append(None)
                else:
                    both = _join(r.contiguous(), f.contiguous())
                    fargs.append(both)
                    joined[id(both)] = len(reals)
                    reals.append(r)
                    fakes.append(f)
            else:
                fargs.append(args[pos])
                pos += 1
        if hasattr(F, 'pair_full'):
            fargs = F.pair_full(fargs)
        rec = _PairCtx()
        need, sub_need, pos = ctx.needs_input_grad, [], 1            # (what F's forward may consult to prepare its backward)
        for kind in layout:
            sub_need.append((need[pos] or need[pos + 1]) if kind == 'b' else need[pos])
            pos += 2 if kind == 'b' else 1
        rec.needs_input_grad = tuple(sub_need)
        out = F.forward(rec, *fargs)
        outs = out if isinstance(out, tuple) else (out,)
        kinds = out_spec if out_spec != '*' else 'b' * len(outs)
        # A saved tensor that IS a joined batch input is remembered as such: the real-half pass of ``backward`` must hand
        # ``F.backward`` the ORIGINAL real input (a saved input is re-attached to the graph when unpacked; the joined alias
        # made in here has no history), or the R1 penalty's second derivative through that input would be lost.
        ctx.from_input = [None if t is None else joined.get(id(t)) for t in rec.saved_tensors]
        ctx.n_halves = len(reals)
        flat, out_at = [], {}
        for kind, o in zip(kinds, outs):
            if kind == 'b':
                n = o.shape[0] // 2
                out_at[id(o)] = len(flat)
                flat += [o[:n], o[n:]]
            else:
                flat.append(o)
        # Likewise a saved tensor that IS a batch output (tanh keeps y): its halves are saved as outputs of THIS node, so the
        # single-half backward differentiates through them.
        ctx.from_output = [None if t is None else out_at.get(id(t)) for t in rec.saved_tensors]
        ctx.out_keep = sorted({at for at in ctx.from_output if at is not None})
        ctx.save_for_backward(*rec.saved_tensors, *reals, *fakes, *[flat[at + k] for at in ctx.out_keep for k in (0, 1)])
        ctx.F, ctx.kinds, ctx.materialize = F, kinds, rec.materialize
        ctx.attrs = {k: v for k, v in vars(rec).items() if k not in ('saved_tensors', 'materialize', 'needs_input_grad')}
        if 'data_input' in ctx.attrs:
            first = layout.index('b')
            pos0 = sum(2 if k == 'b' else 1 for k in layout[:first])
            ctx.attrs['data_input'] = is_data(args[pos0]) and is_data(args[pos0 + 1])
        ctx.set_materialize_grads(False)
        ctx.out_meta = [(tuple(o.shape), o.dtype) for o in flat]
        return tuple(flat)

    @staticmethod
    def backward(ctx, *gouts):
        F, kinds = ctx.F, ctx.kinds
        layout, saved_spec, _ = PAIR_SPECS[F]
        n_in = sum(2 if k == 'b' else 1 for k in layout)
        if all(g is None for g in gouts):
            return (None,) * (1 + n_in)
        groups, pos = [], 0                      # per output of F: (kind, real grad, fake grad) / (kind, grad, None)
        for kind in kinds:
            if kind == 'b':
                groups.append((kind, gouts[pos], gouts[pos + 1], pos))
                pos += 2
            else:
                groups.append((kind, gouts[pos], None, pos))
                pos += 1
        has_b = any(kind == 'b' for kind, *_ in groups)
        half = has_b and all(gf is None for kind, _, gf, _ in groups if kind == 'b')          # the first half alone
        half_f = has_b and not half and all(g is None for kind, g, _, _ in groups if kind == 'b')   # the second half alone
        need = ctx.needs_input_grad
        sub = _PairCtx()
        sub.__dict__.update(ctx.attrs)
        sub_need, pos = [], 1
        for kind in layout:
            if kind == 'b':
                sub_need.append(need[pos] if half else need[pos + 1] if half_f else (need[pos] or need[pos + 1]))
                pos += 2
            else:
                sub_need.append(need[pos])
                pos += 1
        sub.needs_input_grad = tuple(sub_need)
        n_saved = len(ctx.from_input)
        saved = ctx.saved_tensors[:n_saved]
        reals = ctx.saved_tensors[n_saved:n_saved + ctx.n_halves]
        fakes = ctx.saved_tensors[n_saved + ctx.n_halves:n_saved + 2 * ctx.n_halves]
        kept = ctx.saved_tensors[n_saved + 2 * ctx.n_halves:]
        device = next(g for g in gouts if g is not None).device

        def zeros(at):
            shape, dtype = ctx.out_meta[at]
            return torch.zeros(shape, dtype=dtype, device=device)

        gin = []
        if half or half_f:
            own = reals if half else fakes

            def view(t):                    # this half's rows of a saved tensor (statistics: this group's entries)
                n = t.shape[0] // 2
                return t[:n] if half else t[n:]
            own_out = {at: kept[2 * i + (0 if half else 1)] for i, at in enumerate(ctx.out_keep)}
            sub.saved_tensors = tuple(own[j] if j is not None else own_out[o] if o is not None
                                      else (view(t) if (k == 'h' and t is not None) else t)
                                      for k, t, j, o in zip(saved_spec, saved, ctx.from_input, ctx.from_output))
            if getattr(sub, '_out_meta', None):          # (what the base wrapper zero-fills missing gradients from: B rows here)
                sub._out_meta = [m if (m is None or kind != 'b') else ((m[0][0] // 2,) + tuple(m[0][1:]), m[1], m[2])
                                 for kind, m in zip(kinds, sub._out_meta)]
            if hasattr(F, 'pair_half'):
                F.pair_half(sub)
            for kind, g, gf, at in groups:
                g = g if (half or kind != 'b') else gf
                gin.append(zeros(at + (1 if (half_f and kind == 'b') else 0)) if (g is None and ctx.materialize) else g)
        else:
            sub.saved_tensors = saved
            for kind, g, gf, at in groups:
                if kind != 'b':
                    gin.append(zeros(at) if (g is None and ctx.materialize) else g)
                elif g is None and gf is None:
                    gin.append(_join(zeros(at), zeros(at + 1)) if ctx.materialize else None)
                else:
                    g = zeros(at) if g is None else g.contiguous()
                    gf = zeros(at + 1) if gf is None else gf.contiguous()
                    gin.append(_join(g, gf))
        res = F.backward(sub, *gin)
        res = res if isinstance(res, tuple) else (res,)
        res = tuple(res) + (None,) * (len(layout) - len(res))
        out = [None]
        for kind, g in zip(layout, res):
            if kind != 'b':
                out.append(g)
            elif g is None:
                out += [None, None]
            elif half:
                out += [g, None]
            elif half_f:
                out += [None, g]
            else:
                n = g.shape[0] // 2
                out += [g[:n], g[n:]]
        return tuple(out)


def pair_apply(F, *fargs):
    """``F.apply`` for arguments that contain ``Pair``s (every batch argument a Pair or None) -> Pair(s) for batch outputs."""
    layout, _, out_spec = PAIR_SPECS[F]
    if len(fargs) != len(layout):
        raise TypeError(f'{F.__name__}: {len(layout)} arguments expected for the paired form, got {len(fargs)}')
    flat = []
    for kind, a in zip(layout, fargs):
        if kind != 'b':
            flat.append(a)
        elif a is None:
            flat += [None, None]
        elif isinstance(a, Pair):
            flat += [a.r, a.f]
        else:
            raise TypeError(f'{F.__name__}: a batch argument of the paired form must be a Pair (or None)')
    outs = _Paired.apply(F, *flat)
    kinds = out_spec if out_spec != '*' else 'b' * (len(outs) // 2)
    res, pos = [], 0
    for kind in kinds:
        if kind == 'b':
            res.append(Pair(outs[pos], outs[pos + 1]))
            pos += 2
        else:
            res.append(outs[pos])
            pos += 1
    return res[0] if len(res) == 1 else tuple(res)


def _is_pair(x):
    return isinstance(x, Pair)


def per_half(fn, x, *args, **kw):
    """Fallback for an op without a paired form: run it on each half (real first -- RNG / running-statistics order)."""
    r = fn(x.r, *args, **kw)
    f = fn(x.f, *args, **kw)
    return Pair(r, f)


# =========================================================================== conv
class _ConvFwd(Function):
    """y = conv(x, w) + bias [+ residual]; the residual add of the reference's blocks (x + h) rides in the epilogue.
    ``residual_up``: the residual has half the resolution and is added nearest-neighbour upsampled (tg_conv2d_fwd_up2res)."""

    @staticmethod
    def forward(ctx, x, w, bias, residual=None, residual_up=False):
        x, w = x.contiguous(), w.contiguous()
        B, Cin, H, W = x.shape
        Cout, ks = w.shape[0], w.shape[2]
        y = x.new_empty(B, Cout, H, W)
        if residual is not None:
            residual = residual.contiguous()
        if residual_up:
            if residual is None or ks != 3 or tuple(residual.shape) != (B, Cout, H // 2, W // 2) or H % 2 or W % 2:
                raise RuntimeError('residual_up needs a (B, Cout, H/2, W/2) residual on a 3x3 convolution of an even plane')
            K().conv2d_fwd_up2res(x, w, bias, residual, y, B, Cin, Cout, H, W)
        else:
            K().conv2d_fwd(x, w, bias, residual, y, B, Cin, Cout, H, W, ks)
        ctx.save_for_backward(x, w, bias)
        ctx.has_residual = residual is not None
        ctx.residual_up = residual_up
        ctx.data_input = is_data(x)          # (under _Paired: overwritten from the original halves)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, bias = ctx.saved_tensors
        gy = gy.contiguous()
        gx = gw = gb = None
        need_w = ctx.needs_input_grad[1] and _param_grads_wanted()
        need_b = bias is not None and ctx.needs_input_grad[2] and _param_grads_wanted()
        if ctx.needs_input_grad[0] and not _skip_input_grad(ctx):
            gx = _ConvDgrad.apply(gy, w)
        sink_w, sink_b = _grad_sink(w), _grad_sink(bias)
        if need_w and sink_w is not None and (not need_b or sink_b is not None):
            _conv_wgrad_into(x, gy, sink_w, sink_b if need_b else None, w.shape[2], accumulate=1)
        else:
            if need_w:
                gw = _ConvWgrad.apply(x, gy, w.shape[2])
            if need_b:
                gb = _ChannelSum.apply(gy)
        gres = None
        if ctx.has_residual and ctx.needs_input_grad[3]:
            gres = _Pool2.apply(gy, 1.0) if ctx.residual_up else gy      # transpose of the nearest-neighbour upsampling: 2x2 sums
        return gx, gw, gb, gres, None


class _ConvDgrad(Function):
    @staticmethod
    def forward(ctx, gy, w):
        gy, w = gy.contiguous(), w.contiguous()
        B, Cout, H, W = gy.shape
        Cin, ks = w.shape[1], w.shape[2]
        gx = gy.new_empty(B, Cin, H, W)
        K().conv2d_dgrad(gy, w, gx, B, Cin, Cout, H, W, ks)
        ctx.save_for_backward(gy, w)
        return gx

    @staticmethod
    def backward(ctx, v):
        gy, w = ctx.saved_tensors
        v = v.contiguous()
        a_gy = a_w = None
        if ctx.needs_input_grad[0]:
            a_gy = _ConvFwd.apply(v, w, None, None)
        if ctx.needs_input_grad[1]:
            # (the R1 penalty's contribution to the filter gradient) straight into .grad when it is a plain backward
            sink = _grad_sink(w) if _param_grads_wanted() else None
            if sink is not None:
                _conv_wgrad_into(v, gy, sink, None, w.shape[2], accumulate=1)
            else:
                a_w = _ConvWgrad.apply(v, gy, w.shape[2])
        return a_gy, a_w


class _DeferredWgrad:
    """State of ``deferred_wgrad()``: rows of the batch-reduce table + the partial-sum buffers they point at."""
    active = False
    items = []
    s2_items = []       # the stride-2 forms' rows (tg_s2_wgrad_reduce_batch)
    keep = []


@contextlib.contextmanager
def deferred_wgrad():
    """Batch the second stage of every conv weight gradient that accumulates straight into ``.grad``.

    A backward pass runs ~30-50 weight-gradient kernels, each followed by a reduction of its per-workgroup partials
    that is ~1 us of work behind ~5 us of launch latency.  Inside this context the reductions are recorded instead
    and run as ONE launch on exit (``tg_conv2d_wgrad_reduce_batch``; same partials, same summation order, so
    bit-identical gradients).  Only the trainers use it, around their own ``backward`` calls: until the context
    exits the ``.grad`` of conv weights is incomplete, which a foreign training loop could not know.
    """
    outer = _DeferredWgrad.active
    _DeferredWgrad.active = True
    try:
        with grads_into_buckets():      # the batched reduce only exists for gradients that go straight into .grad
            yield
    finally:
        _DeferredWgrad.active = outer
        if not outer:
            flush_wgrad()


class _WgradStream:
    """Weight gradients beside the backward's critical path.  In a backward pass the chain is
    BatchNorm backward (HBM-bound) -> input-gradient convolution (MFMA) -> BatchNorm backward -> ...; a layer's weight
    gradient hangs off the side of it (it needs the same gy, nothing downstream needs its result before the optimiser).
    Inside ``deferred_wgrad()`` those launches go to a second stream that forks from the current one where gy is ready and
    joins it again when the context exits: the MFMA-bound weight gradients then run under the bandwidth-bound BatchNorm
    passes of the layers below.  Only gradients that accumulate straight into ``.grad`` take this path (nothing on the main
    stream reads them before the join); their operands are kept alive until the join.
    MEASURED (round 3, 128:3 batch 64, graph replay, same box back to back): 10.49 / 10.52 ms per step with it, 9.90 / 9.86
    without -- the co-running kernels take LDS and issue slots from the convolution kernels on the critical path, which lose
    more than the overlap gains (round 1 found the same with its register-staged kernels).  OFF unless TG_WGRAD_STREAM=1."""
    import os as _os
    enabled = _os.environ.get('TG_WGRAD_STREAM', '0') == '1'
    stream = None
    used = False


@contextlib.contextmanager
def _beside_backward(*operands):
    st = _WgradStream
    if not (st.enabled and _DeferredWgrad.active and operands[0].is_cuda):
        yield
        return
    if st.stream is None:
        st.stream = torch.cuda.Stream()
    st.stream.wait_stream(torch.cuda.current_stream())
    _DeferredWgrad.keep.append(operands)
    st.used = True
    with torch.cuda.stream(st.stream):
        yield


def flush_wgrad():
    st = _DeferredWgrad
    if _WgradStream.used:
        torch.cuda.current_stream().wait_stream(_WgradStream.stream)
        _WgradStream.used = False
        if not st.items and not st.s2_items:
            st.keep = []
    if st.items or st.s2_items:
        items, st.items = st.items, []
        s2_items, st.s2_items = st.s2_items, []
        try:
            # A weight that collected several contributions in this pass (the discriminator sees the real and the
            # fake batch) must not be updated by two workgroups of one launch: contribution k of every weight goes
            # into launch k, in recording order (= the order the per-layer calls would have accumulated in).
            def in_rounds(table):
                rounds, seen = [], {}
                for row in table:
                    k = seen.get(row[1], 0)
                    seen[row[1]] = k + 1
                    if k == len(rounds):
                        rounds.append([])
                    rounds[k].append(row)
                return rounds
            for rows in in_rounds(items):
                K().conv2d_wgrad_reduce_batch(torch.tensor(rows, dtype=torch.int64), len(rows))
            for rows in in_rounds(s2_items):          # (a layer is one form or the other: the two tables share no gradient)
                K().s2_wgrad_reduce_batch(torch.tensor(rows, dtype=torch.int64), len(rows))
        finally:
            st.keep = []


def _conv_wgrad_into(x, gy, gw, gbias, ks, accumulate):
    B, Cin, H, W = x.shape
    Cout = gy.shape[1]
    nbytes = K().conv2d_wgrad_workspace(B, Cin, Cout, H, W, ks)
    if accumulate and _DeferredWgrad.active:
        ws = x.new_empty(nbytes // 4 + 4)               # its own buffer: must survive until the batch reduce
        with _beside_backward(x, gy, ws):
            K().conv2d_wgrad_partials(x, gy, ws, ws.numel() * 4, B, Cin, Cout, H, W, ks, int(gbias is not None))
        _DeferredWgrad.items.append([ws.data_ptr(), gw.data_ptr(), gbias.data_ptr() if gbias is not None else 0,
                                     B, Cin, Cout, H, W, ks, 1])
        _DeferredWgrad.keep.append((ws, gw, gbias))
        return
    ws = _ws(x, nbytes)
    K().conv2d_wgrad(x, gy, gw, gbias, ws, ws.numel() * 4, B, Cin, Cout, H, W, ks, accumulate)


def _s2_wgrad_into(mode, lo_or_x, gy, gw, gbias, B, Cin, Cout, H, W):
    """Accumulate the weight (and bias) gradient of a stride-2 layer into ``gw`` (``gbias``): mode 0 the pooled conv (x high
    resolution, gy low), mode 1 the up-conv (a low resolution, gy high); (H, W) the LOW-resolution plane.  Inside
    ``deferred_wgrad()`` only stage 1 runs now; the sums, the fold onto the 3x3 taps and the bias gradients of all such layers of
    the pass are finished by one launch when the context exits."""
    name = 'poolconv3x3' if mode == 0 else 'upconv3x3'
    nbytes = getattr(K(), name + '_wgrad_workspace')(B, Cin, Cout, H, W)
    if _DeferredWgrad.active:
        ws = gy.new_empty(nbytes // 4 + 4)               # its own buffer: must survive until the batch reduce
        with _beside_backward(lo_or_x, gy, ws):
            getattr(K(), name + '_wgrad_partials')(lo_or_x, gy, ws, ws.numel() * 4, B, Cin, Cout, H, W, int(gbias is not None))
        _DeferredWgrad.s2_items.append([ws.data_ptr(), gw.data_ptr(), gbias.data_ptr() if gbias is not None else 0,
                                        B, Cin, Cout, H, W, mode, 1])
        _DeferredWgrad.keep.append((ws, gw, gbias))
        return
    ws = _ws(gy, nbytes)
    with _beside_backward(lo_or_x, gy, ws):
        getattr(K(), name + '_wgrad')(lo_or_x, gy, gw, ws, ws.numel() * 4, B, Cin, Cout, H, W, 1, gbias)


class _ConvWgrad(Function):
    @staticmethod
    def forward(ctx, x, gy, ks):
        x, gy = x.contiguous(), gy.contiguous()
        gw = x.new_empty(gy.shape[1], x.shape[1], ks, ks)
        _conv_wgrad_into(x, gy, gw, None, ks, accumulate=0)
        ctx.save_for_backward(x, gy)
        return gw

    @staticmethod
    def backward(ctx, vw):
        x, gy = ctx.saved_tensors
        vw = vw.contiguous()
        a_x = a_gy = None
        if ctx.needs_input_grad[0]:
            a_x = _ConvDgrad.apply(gy, vw)
        if ctx.needs_input_grad[1]:
            a_gy = _ConvFwd.apply(x, vw, None, None)
        return a_x, a_gy, None


class _ChannelSum(Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        B, C = x.shape[0], x.shape[1]
        hw = x[0, 0].numel()
        out = x.new_empty(C)
        K().channel_sum(x, out, _ws(x, K().bn_workspace(B, C, hw)), B, C, hw, 0)
        ctx.shape = x.shape
        return out

    @staticmethod
    def backward(ctx, v):
        return _ChannelBcast.apply(v, ctx.shape)


class _ChannelBcast(Function):
    @staticmethod
    def forward(ctx, v, shape):
        v = v.contiguous()
        out = v.new_empty(shape)
        hw = 1
        for s in shape[2:]:
            hw *= s
        K().channel_bcast(v, out, shape[0], shape[1], hw)
        return out

    @staticmethod
    def backward(ctx, g):
        return _ChannelSum.apply(g), None


def conv2d(x, weight, bias=None, residual=None, residual_up=False):
    """3x3 (pad 1) or 1x1 (pad 0) stride-1 convolution, NCHW fp32; optional fused ``+ residual`` (``residual_up``: a half-resolution
    residual, added nearest-neighbour upsampled by the 3x3 kernel's epilogue)."""
    if _is_pair(x):
        return pair_apply(_ConvFwd, x, weight, bias, residual, residual_up)
    return _ConvFwd.apply(x, weight, bias, residual, residual_up)


PAIR_SPECS[_ConvFwd] = ('b--b-', 'h--', 'b')


# =========================================================================== from-RGB 1x1 composed into the first 3x3
class _ComposeRGB(Function):
    """wc[co][c][tap] = sum_m [w1 | b1][m][c] * w3[co][m][tap]: the filter of conv3x3(conv1x1(img) + b1) as ONE convolution over
    [img, 1] (discriminator.py:11-22 + :60-61).  Parameters in, a derived filter out: its backward is only ever a plain
    first-order one (the R1 penalty differentiates w.r.t. the images, never w.r.t. a filter with create_graph)."""

    @staticmethod
    def forward(ctx, w1, b1, w3):
        w1, b1, w3 = w1.contiguous(), b1.contiguous(), w3.contiguous()
        Cout, C = w3.shape[:2]
        Cimg = w1.shape[1]
        wc = w3.new_empty(Cout, Cimg + 1, 3, 3)
        K().rgb_compose_fwd(w1, b1, w3, wc, Cout, C, Cimg)
        ctx.save_for_backward(w1, b1, w3)
        return wc

    @staticmethod
    @once_differentiable
    def backward(ctx, gwc):
        w1, b1, w3 = ctx.saved_tensors
        if not _param_grads_wanted() or not any(ctx.needs_input_grad):
            return None, None, None
        Cout, C = w3.shape[:2]
        Cimg = w1.shape[1]
        sinks = [_grad_sink(p) for p in (w1, b1, w3)]
        if all(ctx.needs_input_grad) and all(sk is not None for sk in sinks):
            K().rgb_compose_bwd(gwc.contiguous(), w1, b1, w3, sinks[0], sinks[1], sinks[2], Cout, C, Cimg, 1)
            return None, None, None
        g1, gb, g3 = torch.empty_like(w1), torch.empty_like(b1), torch.empty_like(w3)
        K().rgb_compose_bwd(gwc.contiguous(), w1, b1, w3, g1, gb, g3, Cout, C, Cimg, 0)
        need = ctx.needs_input_grad
        return (g1 if need[0] else None), (gb if need[1] else None), (g3 if need[2] else None)


def compose_rgb_filter(w1, b1, w3):
    """-> (Cout, Cimg + 1, 3, 3): conv3x3(conv1x1(img, w1) + b1, w3) == conv3x3([img, 1], result) with zero padding."""
    return _ComposeRGB.apply(w1, b1, w3)


# =========================================================================== theta | phi | g of SelfAttention2d
def _adjacent(ts):
    """One (rows, cols) view over tensors that lie back to back in one storage (the parameters of a network, and their
    gradients, do: tartangan_amd.optim keeps each network in one flat bucket, in registration order), else None."""
    first = ts[0]
    if first is None:
        return None
    cols = first[0].numel()
    off = first.storage_offset()
    for t in ts:
        if (t is None or not t.is_contiguous() or t[0].numel() != cols or t.dtype != first.dtype
                or t.untyped_storage().data_ptr() != first.untyped_storage().data_ptr() or t.storage_offset() != off):
            return None
        off += t.numel()
    return first.as_strided((sum(t.shape[0] for t in ts), cols), (cols, 1), first.storage_offset())


def _stacked(ws):
    view = _adjacent(ws)
    return view if view is not None else torch.cat([w.reshape(w.shape[0], -1) for w in ws])


def _qkv_wgrad(x, gys, ws, need):
    """Filter gradients of the three projections from (x, their output gradients): ONE pass over x when the three ``.grad``
    buffers can be written in place as one matrix, else the generic (differentiable) per-layer path."""
    if not any(need) or not _param_grads_wanted():
        return [None, None, None]
    sinks = [_grad_sink(w) for w in ws]
    cat = _adjacent(sinks) if all(need) and all(sk is not None for sk in sinks) else None
    if cat is None:
        return [_ConvWgrad.apply(x, g, 1) if n else None for g, n in zip(gys, need)]
    B, Cin, H, W = x.shape
    cs = [w.shape[0] for w in ws]
    nbytes = K().conv1x1_multi_wgrad_workspace(*cs, B, Cin, H, W)
    wsb = _ws(x, nbytes)
    K().conv1x1_multi_wgrad(x, gys[0], gys[1], gys[2], cat, wsb, wsb.numel() * 4, *cs, B, Cin, H, W, 1)
    return [None, None, None]


class _QKV(Function):
    """(conv1x1(x, w_theta), conv1x1(x, w_phi), conv1x1(x, w_g)) of SelfAttention2d (attention.py:22-26) in one pass over x
    (tg_conv1x1_multi_*); backward: one pass for the input gradient (no three-way add), one for the three filter gradients."""

    @staticmethod
    def forward(ctx, x, wt, wp, wg):
        x = x.contiguous()
        B, Cin, H, W = x.shape
        cs = (wt.shape[0], wp.shape[0], wg.shape[0])
        ys = [x.new_empty(B, c, H, W) for c in cs]
        K().conv1x1_multi_fwd(x, _stacked((wt, wp, wg)), ys[0], ys[1], ys[2], *cs, B, Cin, H, W)
        ctx.save_for_backward(x, wt, wp, wg)
        ctx.data_input = is_data(x)
        return tuple(ys)

    @staticmethod
    def backward(ctx, gt, gp, gg):
        x, wt, wp, wg = ctx.saved_tensors
        gys = [g.contiguous() for g in (gt, gp, gg)]
        gx = None
        if ctx.needs_input_grad[0] and not _skip_input_grad(ctx):
            gx = _QKVDgrad.apply(gys[0], gys[1], gys[2], wt, wp, wg)
        gws = _qkv_wgrad(x, gys, (wt, wp, wg), ctx.needs_input_grad[1:4])
        return (gx, *gws)


class _QKVDgrad(Function):
    """sum_k conv1x1_dgrad(gy_k, w_k): the input gradient of ``_QKV``; its own backward closes the loop (R1 penalty)."""

    @staticmethod
    def forward(ctx, gt, gp, gg, wt, wp, wg):
        gys = [g.contiguous() for g in (gt, gp, gg)]
        B, _, H, W = gys[0].shape
        Cin = wt[0].numel()
        cs = (wt.shape[0], wp.shape[0], wg.shape[0])
        gx = gys[0].new_empty(B, Cin, H, W)
        K().conv1x1_multi_dgrad(gys[0], gys[1], gys[2], _stacked((wt, wp, wg)), gx, *cs, B, Cin, H, W)
        ctx.save_for_backward(gys[0], gys[1], gys[2], wt, wp, wg)
        return gx

    @staticmethod
    def backward(ctx, v):
        gt, gp, gg, wt, wp, wg = ctx.saved_tensors
        v = v.contiguous()
        a_g = [None, None, None]
        if any(ctx.needs_input_grad[:3]):
            a_g = list(_QKV.apply(v, wt, wp, wg))
        # (the R1 penalty's contribution to the three filter gradients: the same product with v in the place of x)
        a_w = _qkv_wgrad(v, (gt, gp, gg), (wt, wp, wg), ctx.needs_input_grad[3:6])
        return (*a_g, *a_w)


def qkv_supported(x, w_theta, w_phi, w_g):
    B, Cin, H, W = x.shape
    cs = (w_theta.shape[0], w_phi.shape[0], w_g.shape[0])
    return all(bool(K().conv1x1_multi_supported(*cs, b, Cin, H, W)) for b in ((B, B // 2) if _is_pair(x) else (B,)))


def qkv_projections(x, w_theta, w_phi, w_g):
    """-> (theta(x), phi(x), g(x)): three bias-free 1x1 convolutions of one input, one pass."""
    if _is_pair(x):
        return pair_apply(_QKV, x, w_theta, w_phi, w_g)
    return _QKV.apply(x, w_theta, w_phi, w_g)


PAIR_SPECS[_QKV] = ('b---', 'h---', 'bbb')


# =========================================================================== GEMM
class _UpConv3x3(Function):
    """conv3x3(nearest_up2x(a)) + bias [+ residual] as four 2x2-tap convolutions at the low resolution
    (tg_upconv3x3_*): 2.25x fewer FLOPs than the reference's up -> conv, and the upsampled tensor is never written in
    the forward pass."""

    @staticmethod
    def forward(ctx, a, w, bias, residual=None):
        a, w = a.contiguous(), w.contiguous()
        B, Cin, H, W = a.shape
        Cout = w.shape[0]
        if w.shape[2:] != (3, 3):
            raise RuntimeError('upconv3x3 needs a 3x3 filter')
        wp = a.new_empty(4, Cout, Cin, 2, 2)
        ctx.w4t = None
        need = getattr(ctx, 'needs_input_grad', ())
        if need and need[0] and K().upconv3x3_dgrad_supported(B, Cin, Cout, H, W):
            # the backward's filter layout in the same launch (the filter cannot change between this pass and its backward
            # without autograd's version check failing on the saved ``w``)
            ctx.w4t = a.new_empty(Cin, Cout, 4, 4)
            K().upconv3x3_weights_pair(w, wp, ctx.w4t, Cout, Cin)
        else:
            K().upconv3x3_weights(w, wp, Cout, Cin)
        if residual is not None:
            residual = residual.contiguous()
        y = a.new_empty(B, Cout, 2 * H, 2 * W)
        K().upconv3x3_fwd(a, wp, bias, residual, y, B, Cin, Cout, H, W)
        ctx.save_for_backward(a, w, bias)
        ctx.has_residual = residual is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        a, w, bias = ctx.saved_tensors
        need = ctx.needs_input_grad
        gres = gy if ctx.has_residual and need[3] else None
        if torch.is_grad_enabled():
            # double backward requested: rebuild from the twice-differentiable primitives
            with torch.enable_grad():
                y = conv2d(upsample_nearest2x(a), w, bias)
            inputs = [t for t, n in zip((a, w, bias), need[:3]) if n and t is not None]
            got = iter(torch.autograd.grad(y, inputs, gy, create_graph=True))
            ga, gw, gb = (next(got) if (n and t is not None) else None for t, n in zip((a, w, bias), need[:3]))
            return ga, gw, gb, gres
        gy = gy.contiguous()
        ga = gw = gb = None
        if need[0]:
            B, Cin, H, W = a.shape
            Cout = w.shape[0]
            if K().upconv3x3_dgrad_supported(B, Cin, Cout, H, W):
                # one 4x4-tap stride-2 pass over gy instead of dgrad3x3 at the high resolution + a 2x2 sum
                w4t = getattr(ctx, 'w4t', None)
                if w4t is None:
                    w4t = a.new_empty(Cin, Cout, 4, 4)
                    K().upconv3x3_weights_t(w, w4t, Cout, Cin)
                ga = torch.empty_like(a)
                K().upconv3x3_dgrad(gy, w4t, ga, B, Cin, Cout, H, W)
            else:
                ga = _Pool2.apply(_ConvDgrad.apply(gy, w), 1.0)
        need_w = need[1] and _param_grads_wanted()
        need_b = bias is not None and need[2] and _param_grads_wanted()
        sink_w, sink_b = _grad_sink(w), _grad_sink(bias)
        if need_w:
            B, Cin, H, W = a.shape
            Cout = w.shape[0]
            if K().upconv3x3_dgrad_supported(B, Cin, Cout, H, W):
                # 16-tap stride-2 weight gradient on (a, gy): no up2x(a), 2.25x fewer FLOPs; the bias gradient rides along
                # (the kernel sums the gy values it stages) when both go the same way (both sinks, or both returned)
                ws = _ws(a, K().upconv3x3_wgrad_workspace(B, Cin, Cout, H, W))
                with_b = need_b and ((sink_w is None) == (sink_b is None))
                if sink_w is None:
                    gw = torch.empty_like(w)
                    gb = torch.empty_like(bias) if with_b else None
                    K().upconv3x3_wgrad(a, gy, gw, ws, ws.numel() * 4, B, Cin, Cout, H, W, 0, gb)
                else:
                    _s2_wgrad_into(1, a, gy, sink_w, sink_b if with_b else None, B, Cin, Cout, H, W)
                need_b = need_b and not with_b
            elif sink_w is not None:
                _conv_wgrad_into(upsample_nearest2x(a), gy, sink_w, None, 3, accumulate=1)
            else:
                gw = _ConvWgrad.apply(upsample_nearest2x(a), gy, 3)
        if need_b:
            if sink_b is not None:
                Bn, Cn = gy.shape[:2]
                hw = gy[0, 0].numel()
                K().channel_sum(gy, sink_b, _ws(gy, K().bn_workspace(Bn, Cn, hw)), Bn, Cn, hw, 1)
            else:
                gb = _ChannelSum.apply(gy)
        return ga, gw, gb, gres


def upconv3x3_pays(a, weight):
    """True where tg_upconv3x3_fwd runs as ONE kernel (enough workgroups of 256 low-resolution pixels x 16 output
    channels, source planes of 8x8 or more); smaller layers are faster as up2x + the 3x3 kernel."""
    B, _, H, W = a.shape
    if (H, W) == (8, 8):
        tiles, need = (B + 3) // 4, 128
    elif (H, W) == (16, 16):
        tiles, need = B, 256
    elif H >= 8 and W >= 8 and (H, W) != (4, 4):
        tiles, need = B * ((H + 7) // 8) * ((W + 31) // 32), 256
    else:
        return False
    return tiles * ((weight.shape[0] + 15) // 16) >= need


def upconv3x3(a, weight, bias=None, residual=None):
    """conv2d(F.interpolate(a, scale_factor=2), weight, bias, padding=1) [+ residual]"""
    if _is_pair(a):
        return pair_apply(_UpConv3x3, a, weight, bias, residual)
    return _UpConv3x3.apply(a, weight, bias, residual)


PAIR_SPECS[_UpConv3x3] = ('b--b', 'h--', 'b')


class _FilterForms:
    """Derived filter layouts of the stride-2 forms, kept for the length of one ``filter_forms()`` scope."""
    cache = None
    owner = None            # the module whose pass the innermost scope brackets (or None)


@contextlib.contextmanager
def filter_forms(owner=None):
    """Scope in which the parameters do not change (one half of a training step: the discriminator is evaluated up to six
    times between two of its optimiser steps): the 4x4 / four-phase layouts derived from a 3x3 filter are computed once
    per scope instead of once per use.  Nothing outlives the scope, so nothing can go stale.

    ``owner`` (a module): the filters this scope had to derive are remembered FOR the module, and the next scope opened for it
    derives all of them up front in ONE launch (tg_poolconv3x3_weights_batch) instead of one launch per layer as they come."""
    outer, outer_owner = _FilterForms.cache, _FilterForms.owner
    _FilterForms.cache, _FilterForms.owner = {}, owner
    try:
        if owner is not None:
            _derive_known_forms(owner)
        yield
    finally:
        _FilterForms.cache, _FilterForms.owner = outer, outer_owner


_KNOWN_FORMS = weakref.WeakKeyDictionary()       # module -> {id(weight): weakref(weight)}; beside the module, not in it (pickling)


def _known_forms(owner):
    known = _KNOWN_FORMS.get(owner)
    if known is None:
        # start from the module's structure: every ``Sequential`` that ends in conv3x3 -> AvgPool2d(2) is a candidate for the
        # stride-2 form (models.layers.run_layers decides per call, by shape); the per-scope learning below adds whatever
        # else derives a filter.  So the very first pass -- the one a HIP graph is captured from -- already batches.
        known = _KNOWN_FORMS[owner] = {}
        for m in owner.modules():
            if isinstance(m, torch.nn.Sequential) and len(m) >= 2:
                conv, pool = m[len(m) - 2], m[len(m) - 1]
                w = getattr(conv, 'weight', None)
                if (type(pool).__name__ == 'AvgPool2d' and isinstance(w, torch.nn.Parameter) and w.dim() == 4
                        and tuple(w.shape[2:]) == (3, 3)):
                    known[id(w)] = weakref.ref(w)
    return known


def _derive_known_forms(owner):
    known = _known_forms(owner)
    ws = []
    for key, ref in list(known.items()):
        w = ref()
        if w is None or not w.is_contiguous():
            del known[key]
        else:
            ws.append(w)
    if len(ws) < 2:
        return                                   # (a single layer: its own call is the same one launch)
    rows = []
    for w in ws:
        Cout, Cin = w.shape[:2]
        w4 = w.new_empty(Cout, Cin, 4, 4)
        wp = w.new_empty(4, Cin, Cout, 2, 2)
        rows.append([w.data_ptr(), w4.data_ptr(), wp.data_ptr(), Cout, Cin])
        _FilterForms.cache[(w.data_ptr(), tuple(w.shape))] = (w, w._version, w4.detach(), wp.detach())
    K().poolconv3x3_weights_batch(torch.tensor(rows, dtype=torch.int64), len(rows))


def _poolconv_weights(x_like, w):
    # an entry keeps its source tensor alive (its address cannot be handed to another tensor inside the scope) and is
    # only valid for the version it was derived from
    cache = _FilterForms.cache
    key = (w.data_ptr(), tuple(w.shape))
    if cache is not None and key in cache and cache[key][1] == w._version:
        return cache[key][2], cache[key][3]
    Cout, Cin = w.shape[:2]
    w4 = x_like.new_empty(Cout, Cin, 4, 4)
    wp = x_like.new_empty(4, Cin, Cout, 2, 2)
    K().poolconv3x3_weights(w.contiguous(), w4, wp, Cout, Cin)
    if cache is not None and w.is_contiguous():
        cache[key] = (w, w._version, w4, wp)
        if _FilterForms.owner is not None and isinstance(w, torch.nn.Parameter):
            _known_forms(_FilterForms.owner)[id(w)] = weakref.ref(w)
    return w4, wp


class _PoolConv(Function):
    """AvgPool2d(2)(conv3x3(x) + bias) [+ residual] as one 4x4-tap stride-2 convolution (tg_poolconv3x3_*).
    Twice differentiable: its input gradient is ``_PoolConvT``, its weight gradient the ordinary 3x3 weight gradient of
    (x, 0.25 * up2x(gy)) -- all Functions whose own backwards close the loop."""

    @staticmethod
    def forward(ctx, x, w, bias, residual=None):
        x = x.contiguous()
        B, Cin, H2, W2 = x.shape
        Cout = w.shape[0]
        w4, _ = _poolconv_weights(x, w)
        if residual is not None:
            residual = residual.contiguous()
        y = x.new_empty(B, Cout, H2 // 2, W2 // 2)
        K().poolconv3x3_fwd(x, w4, bias, residual, y, B, Cin, Cout, H2 // 2, W2 // 2)
        ctx.save_for_backward(x, w, bias)
        ctx.has_residual = residual is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, bias = ctx.saved_tensors
        need = ctx.needs_input_grad
        gy = gy.contiguous()
        gx = gw = gb = None
        need_w = need[1] and _param_grads_wanted()
        need_b = bias is not None and need[2] and _param_grads_wanted()
        gres = gy if ctx.has_residual and need[3] else None
        if need[0] and gres is not None and torch.is_grad_enabled() and gy.requires_grad:
            # R1 first-order pass (create_graph): gy feeds the transposed conv AND the shortcut -- one node for both uses, so
            # that the second-order sweep adds their two adjoints in one kernel instead of an autograd add
            gy, gres = _ForkN.apply(gy, 2)
        if need[0]:
            gx = _PoolConvT.apply(gy, w)
        if need_w:
            if torch.is_grad_enabled():
                gw = _ConvWgrad.apply(x, _Up2x.apply(gy, 0.25), 3)  # differentiable form (what AvgPool2d's backward hands the conv)
            else:
                B, Cin, H2, W2 = x.shape
                Cout = w.shape[0]
                sink_w = _grad_sink(w)
                ws = _ws(x, K().poolconv3x3_wgrad_workspace(B, Cin, Cout, H2 // 2, W2 // 2))
                # the bias gradient (at the LOW resolution) rides along: the kernel sums the gy values it stages
                sink_b = _grad_sink(bias) if need_b else None
                with_b = need_b and ((sink_w is None) == (sink_b is None))
                if sink_w is None:
                    gw = torch.empty_like(w)
                    gb = torch.empty_like(bias) if with_b else None
                    K().poolconv3x3_wgrad(x, gy, gw, ws, ws.numel() * 4, B, Cin, Cout, H2 // 2, W2 // 2, 0, gb)
                else:
                    _s2_wgrad_into(0, x, gy, sink_w, sink_b if with_b else None, B, Cin, Cout, H2 // 2, W2 // 2)
                need_b = need_b and not with_b
        if need_b:                                                   # bias gradient at the LOW resolution
            sink_b = _grad_sink(bias)
            if sink_b is not None:
                Bn, Cn = gy.shape[:2]
                hw = gy[0, 0].numel()
                K().channel_sum(gy, sink_b, _ws(gy, K().bn_workspace(Bn, Cn, hw)), Bn, Cn, hw, 1)
            else:
                gb = _ChannelSum.apply(gy)
        return gx, gw, gb, gres


class _PoolConvT(Function):
    """transpose of ``_PoolConv`` in x: gy (B,Cout,H,W) -> gx (B,Cin,2H,2W) (the four-phase up-conv kernel)"""

    @staticmethod
    def forward(ctx, gy, w):
        gy = gy.contiguous()
        B, Cout, H, W = gy.shape
        Cin = w.shape[1]
        _, wp = _poolconv_weights(gy, w)
        gx = gy.new_empty(B, Cin, 2 * H, 2 * W)
        K().poolconv3x3_dgrad(gy, wp, gx, B, Cin, Cout, H, W)
        ctx.save_for_backward(gy, w)
        return gx

    @staticmethod
    def backward(ctx, v):
        gy, w = ctx.saved_tensors
        v = v.contiguous()
        a_gy = a_w = None
        if ctx.needs_input_grad[0]:
            a_gy = _PoolConv.apply(v, w, None, None)
        if ctx.needs_input_grad[1]:
            if torch.is_grad_enabled():
                a_w = _ConvWgrad.apply(v, _Up2x.apply(gy, 0.25), 3)
            else:                                   # <v, PoolConvT(gy, w)> = <PoolConv(v, w), gy>: the pooled conv's weight gradient
                B, Cout, H, W = gy.shape
                Cin = w.shape[1]
                sink = _grad_sink(w)
                if sink is None:
                    ws = _ws(v, K().poolconv3x3_wgrad_workspace(B, Cin, Cout, H, W))
                    a_w = torch.empty_like(w)
                    K().poolconv3x3_wgrad(v, gy, a_w, ws, ws.numel() * 4, B, Cin, Cout, H, W, 0, None)
                else:
                    _s2_wgrad_into(0, v, gy, sink, None, B, Cin, Cout, H, W)
        return a_gy, a_w


def pool_conv3x3_supported(x, weight):
    B, Cin, H2, W2 = x.shape
    if not (H2 % 2 == 0 and W2 % 2 == 0 and tuple(weight.shape[2:]) == (3, 3)):
        return False
    # a Pair runs forward / final backward on 2B images but the R1 passes on the real half alone: both batch sizes must
    # take this form (the kernel's own precondition includes "enough workgroups", which depends on the batch)
    return all(bool(K().poolconv3x3_supported(b, Cin, weight.shape[0], H2 // 2, W2 // 2)) for b in ((B, B // 2) if _is_pair(x) else (B,)))


def pool_conv3x3(x, weight, bias=None, residual=None):
    """F.avg_pool2d(F.conv2d(x, weight, bias, padding=1), 2) [+ residual]"""
    if _is_pair(x):
        return pair_apply(_PoolConv, x, weight, bias, residual)
    return _PoolConv.apply(x, weight, bias, residual)


PAIR_SPECS[_PoolConv] = ('b--b', 'h--', 'b')


class _Gemm(Function):
    """C = op(A) op(B); A,B 2-D or batched 3-D row-major."""

    @staticmethod
    def forward(ctx, A, Bm, ta, tb):
        A, Bm = A.contiguous(), Bm.contiguous()
        batch = A.shape[0] if A.dim() == 3 else 1
        ar, ac = A.shape[-2], A.shape[-1]
        br, bc = Bm.shape[-2], Bm.shape[-1]
        M, Kd = (ac, ar) if ta else (ar, ac)
        Kb, N = (bc, br) if tb else (br, bc)
        assert Kd == Kb, (A.shape, Bm.shape, ta, tb)
        shape = (batch, M, N) if A.dim() == 3 else (M, N)
        C = A.new_empty(shape)
        if A.dim() == 2 and ta and not tb:
            _gemm_tn_into(A, Bm, C, accumulate=False)
        else:
            K().gemm(A, Bm, C, None, M, N, Kd, ac, bc, N, int(ta), int(tb), batch,
                     ar * ac, br * bc, M * N, 0.0)
        ctx.save_for_backward(A, Bm)
        ctx.ta, ctx.tb = ta, tb
        return C

    @staticmethod
    def backward(ctx, g):
        A, Bm = ctx.saved_tensors
        ta, tb = ctx.ta, ctx.tb
        gA = gB = None
        if ctx.needs_input_grad[0]:
            gA = _Gemm.apply(Bm, g, tb, True) if ta else _Gemm.apply(g, Bm, False, not tb)
        if ctx.needs_input_grad[1]:
            gB = _Gemm.apply(g, A, True, ta) if tb else _Gemm.apply(A, g, not ta, False)
        return gA, gB, None, None


def matmul(A, Bm, transA=False, transB=False):
    return _Gemm.apply(A, Bm, transA, transB)


def _gemm_tn_into(A, Bm, out, accumulate):
    """out (+)= A^T B for 2-D A (rows x M), B (rows x N): the weight gradient of a Linear layer.  With many rows and a
    small M x N (the IQN head: 8 quantiles x 64 images into 20 -> 128 and 128 -> 1 layers) a single launch is one or two
    workgroups walking the whole reduction, one dependent global load per 16 rows (86-98 us for 512 rows); there the rows
    are cut into slices that run as ONE batched GEMM, and the slices are summed."""
    rows, M = A.shape
    N = Bm.shape[1]
    S = 1
    if rows >= 256 and M * N <= 256 * 256:      # (one k-step of the GEMM kernel is 16 rows)
        S = 64
        while S > 1 and (rows % S or rows // S < 16):
            S //= 2
    if S == 1:
        K().gemm(A, Bm, out, None, M, N, rows, M, N, N, 1, 0, 1, 0, 0, 0, 1.0 if accumulate else 0.0)
        return
    r = rows // S
    part = A.new_empty(S, M, N)
    K().gemm(A, Bm, part, None, M, N, r, M, N, N, 1, 0, S, r * M, r * N, M * N, 0.0)
    if accumulate:
        total = A.new_empty(M, N)
        K().sum_reps(part, total, 1.0, M, N, S)
        K().add(out, total, out, out.numel())
    else:
        K().sum_reps(part, out, 1.0, M, N, S)


class _Linear(Function):
    @staticmethod
    def forward(ctx, x, w, bias):
        x, w = x.contiguous(), w.contiguous()
        M, Kd = x.shape
        N = w.shape[0]
        y = x.new_empty(M, N)
        K().gemm(x, w, y, bias, M, N, Kd, Kd, Kd, N, 0, 1, 1, 0, 0, 0, 0.0)
        ctx.save_for_backward(x, w, bias)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, bias = ctx.saved_tensors
        gy = gy.contiguous()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = _Gemm.apply(gy, w, False, False)
        if ctx.needs_input_grad[1] and _param_grads_wanted():
            sink = _grad_sink(w)
            if sink is not None:          # gw += gy^T x straight into the flat bucket
                _gemm_tn_into(gy, x, sink, accumulate=True)
            else:
                gw = _Gemm.apply(gy, x, True, False)
        if bias is not None and ctx.needs_input_grad[2] and _param_grads_wanted():
            sink = _grad_sink(bias)
            if sink is not None:
                M, N = gy.shape
                K().channel_sum(gy, sink, _ws(gy, K().bn_workspace(M, N, 1)), M, N, 1, 1)
            else:
                gb = _ChannelSum.apply(gy)
        return gx, gw, gb


def linear(x, weight, bias=None):
    if _is_pair(x):
        return pair_apply(_Linear, x, weight, bias)
    return _Linear.apply(x, weight, bias)


PAIR_SPECS[_Linear] = ('b--', 'h--', 'b')


# =========================================================================== BatchNorm (+LeakyReLU)
class SyncGroup:
    """Handle a BatchNorm2d carries in a data-parallel run with synchronised statistics (parallel.DataParallel(sync_bn=True)):
    every pass all-reduces its per-channel sums over ``group`` (tg_bn_sync_*: local sums -> all-reduce -> finish)."""

    def __init__(self, group, world, rehearse=False):
        self.group, self.world = group, int(world)
        # rehearse: issue the collectives even with ONE rank (parallel.DataParallel(rehearse=True): the RCCL leg of the
        # step -- in-graph collectives, a communicator of their own -- exercised on a one-GPU box; same numbers)
        self.active = self.world > 1 or bool(rehearse)

    def all_reduce(self, sums):
        import torch.distributed as dist
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=self.group)
        return sums


def _sums(like, C, k):
    return torch.empty(C * k, dtype=torch.float64, device=like.device)


class _BNCell:
    """Shared between one BatchNorm forward node and the first-order-backward node the R1 pass derives from it.  In the
    final backward of the discriminator step the input x of a real-branch BatchNorm receives TWO gradients: the
    second-order one (from ``_BNActBwd.backward``, reached first: the second-order sweep climbs the layers before the
    ordinary backward descends them) and the first-order one (``_BNAct.backward``).  Inside ``grads_into_buckets()`` the
    former is parked here and the latter adds it inside its own apply kernel (``gx_add``), instead of leaving a
    full-tensor add to the autograd engine.  If the forward node ran first, nothing is parked (``consumed``)."""
    __slots__ = ('pending', 'consumed')

    def __init__(self):
        self.pending, self.consumed = None, False


class _BNAct(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, training, momentum, eps, slope, num_batches_tracked=None,
                replicate=1, sync=None, groups=1):
        x = x.contiguous()
        B, C = x.shape[0] // groups, x.shape[1]
        hw = x[0, 0].numel()
        mean, invstd = x.new_empty(groups * C), x.new_empty(groups * C)
        z = torch.empty_like(x)
        if groups > 1:
            # the real and the fake batch in one tensor (``Pair``): statistics per half, running statistics updated
            # real-then-fake, num_batches_tracked += 2 -- what the reference's two forwards leave behind
            if not training or (sync is not None and sync.active) or x.shape[0] != groups * B:
                raise RuntimeError('grouped BatchNorm: training mode, local statistics, equal groups only')
            sync = None
            ws = _ws(x, K().bn_workspace(B, groups * C, hw))
            K().bn_train_fwd_groups(x, mean, invstd, running_mean, running_var, num_batches_tracked, gamma, beta,
                                    float(slope), float(momentum), float(eps), z, ws, groups, B, C, hw, int(replicate))
        elif training and sync is not None and sync.active:
            ws = _ws(x, K().bn_workspace(B, C, hw))
            sums = _sums(x, C, 3)
            K().bn_sync_stats_local(x, sums, ws, B, C, hw)
            sync.all_reduce(sums)
            K().bn_sync_stats_finish(sums, sync.world, mean, invstd, running_mean, running_var, num_batches_tracked,
                                     float(momentum), float(eps), sync.world * B * hw, int(replicate), C)
            K().bn_act_fwd(x, mean, invstd, gamma, beta, float(slope), z, B, C, hw)
        elif training:
            sync = None
            ws = _ws(x, K().bn_workspace(B, C, hw))
            K().bn_train_fwd(x, mean, invstd, running_mean, running_var, num_batches_tracked, gamma, beta,
                             float(slope), float(momentum), float(eps), z, ws, B, C, hw, int(replicate))
        else:
            sync = None
            K().bn_eval_stats(running_mean, running_var, mean, invstd, float(eps), C)
            K().bn_act_fwd(x, mean, invstd, gamma, beta, float(slope), z, B, C, hw)
        ctx.save_for_backward(x, gamma, beta, mean, invstd)
        ctx.training, ctx.slope, ctx.sync, ctx.groups = bool(training), float(slope), sync, groups
        ctx.cell = _BNCell()
        return z

    @staticmethod
    def pair_full(fargs):
        fargs[12] = 2
        return fargs

    @staticmethod
    def pair_half(sub):
        sub.groups = 1                  # the real-half view: one group, its own statistics (the first C entries)

    @staticmethod
    def backward(ctx, gz):
        x, gamma, beta, mean, invstd = ctx.saved_tensors
        sink_g, sink_b = _grad_sink(gamma), _grad_sink(beta)
        nones = (None,) * 10
        if sink_g is not None and sink_b is not None:
            # plain backward: ggamma / gbeta accumulate straight into the flat bucket
            gz = gz.contiguous()
            gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
            cell = ctx.cell
            extra, cell.pending, cell.consumed = cell.pending, None, True
            if gx is None:
                extra = None
            _bn_bwd_into(gz, x, mean, invstd, gamma, beta, ctx.slope, ctx.training, gx, sink_g, sink_b, 1, ctx.sync, extra, ctx.groups)
            return (gx, None, None) + nones
        # (a second-order gradient is only ever parked inside grads_into_buckets(), i.e. for the branch above)
        assert ctx.cell.pending is None, 'BatchNorm: a parked second-order input gradient would be dropped'
        gx, gg, gb = _BNActBwd.apply(gz, x, gamma, beta, mean, invstd, ctx.slope, ctx.training,
                                     ctx.needs_input_grad[0], ctx.sync, ctx.cell, ctx.groups)
        return (gx, gg, gb) + nones


def _bn_bwd_into(gz, x, mean, invstd, gamma, beta, slope, training, gx, gg, gb, accumulate, sync, gx_add=None, groups=1):
    B, C = x.shape[0] // groups, x.shape[1]
    hw = x[0, 0].numel()
    ws = _ws(x, K().bn_workspace(B, groups * C, hw))
    if groups > 1:
        # gx_add: the R1 second-order gradient, which exists for the leading (real) group(s) only
        add_groups = 1 if gx_add is None else gx_add.shape[0] // B
        K().bn_act_bwd_groups(gz, x, mean, invstd, gamma, beta, slope, int(training), gx, gg, gb, ws, groups, B, C, hw, accumulate,
                              gx_add, add_groups)
        return
    if sync is None:
        K().bn_act_bwd(gz, x, mean, invstd, gamma, beta, slope, int(training), gx, gg, gb, ws, B, C, hw, accumulate, gx_add)
        return
    local = _sums(x, C, 2)
    K().bn_sync_bwd_local(gz, x, mean, invstd, gamma, beta, slope, local, ws, B, C, hw)
    glob = sync.all_reduce(local.clone())
    K().bn_sync_bwd_finish(gz, x, mean, invstd, gamma, beta, slope, local, glob, sync.world * B * hw, gx, gg, gb, ws,
                           B, C, hw, accumulate, gx_add)


class _BNActBwd(Function):
    handles_none_grads = True

    @staticmethod
    def forward(ctx, gz, x, gamma, beta, mean, invstd, slope, training, need_gx=True, sync=None, cell=None, groups=1):
        gz = gz.contiguous()
        C = x.shape[1]
        gx = torch.empty_like(x) if need_gx else None
        gg, gb = x.new_empty(C), x.new_empty(C)
        _bn_bwd_into(gz, x, mean, invstd, gamma, beta, slope, training, gx, gg, gb, 0, sync, None, groups)
        ctx.save_for_backward(gz, x, gamma, beta, mean, invstd)
        ctx.slope, ctx.training, ctx.sync, ctx.cell, ctx.groups = slope, training, sync, cell, groups
        ctx.set_materialize_grads(False)        # the R1 pass never differentiates ggamma / gbeta: no zero fills for them
        return gx, gg, gb

    @staticmethod
    @once_differentiable
    def backward(ctx, v, vg, vb):
        if not ctx.training:
            raise NotImplementedError('double backward through eval-mode BatchNorm is not implemented')
        if ctx.groups != 1:
            raise NotImplementedError('double backward through grouped BatchNorm (the R1 pass runs on the real half alone)')
        gz, x, gamma, beta, mean, invstd = ctx.saved_tensors
        B, C = x.shape[0], x.shape[1]
        hw = x[0, 0].numel()
        a_gz, a_x = torch.empty_like(x), torch.empty_like(x)
        sink = _grad_sink(gamma)              # the R1 penalty's contribution to gamma's gradient straight into the bucket
        a_gamma = x.new_empty(C) if sink is None else sink
        acc = 0 if sink is None else 1
        ws = _ws(x, K().bn_workspace(B, C, hw))
        if v is None:
            v = torch.zeros_like(x)
        v = v.contiguous()
        nones = (None,) * 9
        cell = ctx.cell

        def park(a_x):
            # inside the trainers' backward: hand the second-order gradient of x to the forward node's backward (see _BNCell)
            if _GradSinks.active and cell is not None and not cell.consumed and cell.pending is None and x.requires_grad:
                cell.pending = a_x
                return None
            return a_x
        if ctx.sync is not None:
            if vg is not None or vb is not None:
                raise NotImplementedError('synchronised BatchNorm: adjoints of ggamma / gbeta are not propagated '
                                          '(the R1 penalty differentiates the input gradient only)')
            sums = _sums(x, C, 5)
            K().bn_sync_dbwd_local(v, gz, x, mean, invstd, gamma, beta, ctx.slope, sums, ws, B, C, hw)
            ctx.sync.all_reduce(sums)
            K().bn_sync_dbwd_finish(v, gz, x, mean, invstd, gamma, beta, ctx.slope, sums, ctx.sync.world * B * hw,
                                    ctx.sync.world, a_gz, a_x, a_gamma, ws, B, C, hw, acc)
            return (a_gz, park(a_x), a_gamma if sink is None else None) + nones
        K().bn_act_dbwd(v, None if vg is None else vg.contiguous(), None if vb is None else vb.contiguous(),
                        gz, x, mean, invstd, gamma, beta, ctx.slope, a_gz, a_x, a_gamma, ws, B, C, hw, acc)
        return (a_gz, park(a_x), a_gamma if sink is None else None) + nones


def batch_norm_act(x, gamma, beta, running_mean, running_var, training, momentum=0.1, eps=1e-5, slope=1.0,
                   num_batches_tracked=None, replicate=1, sync=None):
    """BatchNorm2d followed by LeakyReLU(slope) in one pass (slope=1: plain BN).  In training mode the
    stats kernel also bumps ``num_batches_tracked`` (int64 device scalar) when given.  ``replicate``: ``x`` stands
    for a tensor holding every element that many times (see tg_bn_train_stats); only running_var depends on it.
    ``sync``: a ``SyncGroup`` -- batch statistics and backward sums over all ranks' shards (SyncBN)."""
    if _is_pair(x):
        if training and (sync is None or not sync.active):
            return pair_apply(_BNAct, x, gamma, beta, running_mean, running_var, training, momentum, eps, slope,
                              num_batches_tracked, replicate, None, 1)
        return per_half(batch_norm_act, x, gamma, beta, running_mean, running_var, training, momentum, eps, slope,
                        num_batches_tracked, replicate, sync)
    return _BNAct.apply(x, gamma, beta, running_mean, running_var, training, momentum, eps, slope,
                        num_batches_tracked, replicate, sync, 1)


PAIR_SPECS[_BNAct] = ('b------------', 'h--hh', 'b')


# =========================================================================== resampling
class _Up2x(Function):
    @staticmethod
    def forward(ctx, x, alpha):
        x = x.contiguous()
        B, C, H, W = x.shape
        y = x.new_empty(B, C, 2 * H, 2 * W)
        K().up2x(x, y, alpha, B * C, H, W)
        ctx.alpha = alpha
        return y

    @staticmethod
    def backward(ctx, g):
        return _Pool2.apply(g, ctx.alpha), None


class _Pool2(Function):
    """alpha * (2x2 sum pool) [+ residual]"""

    @staticmethod
    def forward(ctx, x, alpha, residual=None):
        x = x.contiguous()
        B, C, H, W = x.shape
        if H % 2 or W % 2:
            raise RuntimeError('pool2 needs even spatial dims')
        y = x.new_empty(B, C, H // 2, W // 2)
        if residual is not None:
            residual = residual.contiguous()
        K().pool2(x, residual, y, alpha, B * C, H, W)
        ctx.alpha = alpha
        ctx.has_residual = residual is not None
        return y

    @staticmethod
    def backward(ctx, g):
        return _Up2x.apply(g, ctx.alpha), None, (g if ctx.has_residual and ctx.needs_input_grad[2] else None)


def upsample_nearest2x(x):
    if _is_pair(x):
        return pair_apply(_Up2x, x, 1.0)
    return _Up2x.apply(x, 1.0)


PAIR_SPECS[_Up2x] = ('b-', '', 'b')


def avg_pool2(x, residual=None):
    """nn.AvgPool2d(2) [+ residual]"""
    if _is_pair(x):
        return pair_apply(_Pool2, x, 0.25, residual)
    return _Pool2.apply(x, 0.25, residual)


PAIR_SPECS[_Pool2] = ('b-b', '', 'b')


class _BilinearHalf(Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        B, C, H, W = x.shape
        y = x.new_empty(B, C, H // 2, W // 2)
        K().bilinear_half_fwd(x, y, B * C, H, W)
        ctx.hw = (H, W)
        return y

    @staticmethod
    def backward(ctx, g):
        return _BilinearHalfT.apply(g, ctx.hw)


class _BilinearHalfT(Function):
    """transpose of the bilinear x0.5 map [+ residual]"""

    @staticmethod
    def forward(ctx, gy, hw, residual=None):
        gy = gy.contiguous()
        B, C = gy.shape[:2]
        gx = gy.new_empty(B, C, hw[0], hw[1])
        if residual is not None:
            residual = residual.contiguous()
        K().bilinear_half_bwd(gy, residual, gx, B * C, hw[0], hw[1])
        ctx.has_residual = residual is not None
        return gx

    @staticmethod
    def backward(ctx, v):
        return _BilinearHalf.apply(v), None, (v if ctx.has_residual and ctx.needs_input_grad[2] else None)


class _ForkBilinearHalf(Function):
    """x -> (bilinear_half(x), x): the two uses of a discriminator block's input (shortcut and main path) as ONE graph
    node, so that its backward sees both incoming gradients and can add them inside the transpose kernel instead of
    leaving a separate full-resolution add to the autograd engine."""
    handles_none_grads = True

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        B, C, H, W = x.shape
        y = x.new_empty(B, C, H // 2, W // 2)
        K().bilinear_half_fwd(x, y, B * C, H, W)
        ctx.hw = (H, W)
        ctx.set_materialize_grads(False)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, g_half, g_pass):
        if g_half is None:
            return g_pass
        return _BilinearHalfT.apply(g_half, ctx.hw, g_pass)


class _ForkUp2x(Function):
    """x -> (up2x(x), up2x(x)) (one tensor, two graph edges): the generator block's upsampled input feeds the shortcut
    and the main path; backward pools both incoming gradients without first adding them at the high resolution."""
    handles_none_grads = True

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        B, C, H, W = x.shape
        y = x.new_empty(B, C, 2 * H, 2 * W)
        K().up2x(x, y, 1.0, B * C, H, W)
        ctx.set_materialize_grads(False)
        return y, y.view_as(y)

    @staticmethod
    def backward(ctx, ga, gb):
        if ga is None or gb is None:
            g = ga if gb is None else gb
            return None if g is None else _Pool2.apply(g, 1.0)
        return _Pool2.apply(ga, 1.0, _Pool2.apply(gb, 1.0))


def fork_bilinear_half(x):
    """-> (F.interpolate(x, scale_factor=0.5, mode='bilinear', align_corners=True), x)"""
    if _is_pair(x):
        return pair_apply(_ForkBilinearHalf, x)
    return _ForkBilinearHalf.apply(x)


PAIR_SPECS[_ForkBilinearHalf] = ('b', '', 'bb')


def fork_upsample_nearest2x(x):
    """-> (up, up) with up = F.interpolate(x, scale_factor=2), for two consumers"""
    if _is_pair(x):
        return pair_apply(_ForkUp2x, x)
    return _ForkUp2x.apply(x)


PAIR_SPECS[_ForkUp2x] = ('b', '', 'bb')


def bilinear_half(x):
    """F.interpolate(x, scale_factor=0.5, mode='bilinear', align_corners=True)"""
    if _is_pair(x):
        return mark_data(pair_apply(_BilinearHalf, x), x)
    return mark_data(_BilinearHalf.apply(x), x)


PAIR_SPECS[_BilinearHalf] = ('b', '', 'b')


class _MaxPool2(Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        B, C, H, W = x.shape
        y = x.new_empty(B, C, H // 2, W // 2)
        idx = torch.empty(B, C, H // 2, W // 2, dtype=torch.uint8, device=x.device)
        K().maxpool2_fwd(x, y, idx, B * C, H, W)
        ctx.save_for_backward(idx)
        ctx.hw = (H, W)
        return y

    @staticmethod
    def backward(ctx, g):
        idx, = ctx.saved_tensors
        return _MaxPoolScatter.apply(g, idx, ctx.hw)


class _MaxPoolScatter(Function):
    @staticmethod
    def forward(ctx, gy, idx, hw):
        gy = gy.contiguous()
        B, C = gy.shape[:2]
        gx = gy.new_empty(B, C, hw[0], hw[1])
        K().maxpool2_bwd(gy, idx, gx, B * C, hw[0], hw[1])
        ctx.save_for_backward(idx)
        ctx.hw = hw
        return gx

    @staticmethod
    def backward(ctx, v):
        idx, = ctx.saved_tensors
        return _MaxPoolGather.apply(v, idx), None, None


class _MaxPoolGather(Function):
    @staticmethod
    def forward(ctx, x, idx):
        x = x.contiguous()
        B, C, H, W = x.shape
        y = x.new_empty(B, C, H // 2, W // 2)
        K().maxpool2_gather(x, idx, y, B * C, H, W)
        ctx.save_for_backward(idx)
        ctx.hw = (H, W)
        return y

    @staticmethod
    def backward(ctx, g):
        idx, = ctx.saved_tensors
        return _MaxPoolScatter.apply(g, idx, ctx.hw), None


def max_pool2(x):
    if _is_pair(x):
        return pair_apply(_MaxPool2, x)
    return _MaxPool2.apply(x)


PAIR_SPECS[_MaxPool2] = ('b', 'h', 'b')


class _CopyChannels(Function):
    """x (B, C, H, W) -> (B, C2, H, W): channels [0, min(C, C2)) copied, channels beyond C filled with ``fill``.
    C2 > C appends constant channels (the all-ones channel that carries the from-RGB bias into the composed first
    convolution of the discriminator), C2 < C drops trailing ones; each is the other's transpose (with fill 0)."""

    @staticmethod
    def forward(ctx, x, channels, fill):
        x = x.contiguous()
        B, C, H, W = x.shape
        out = x.new_empty(B, channels, H, W)
        K().copy_channels(x, out, B, C, channels, H * W, float(fill))
        ctx.C = C
        return out

    @staticmethod
    def backward(ctx, g):
        return _CopyChannels.apply(g, ctx.C, 0.0), None, None


def copy_channels(x, channels, fill=0.0):
    if _is_pair(x):
        return mark_data(pair_apply(_CopyChannels, x, channels, fill), x)
    return mark_data(_CopyChannels.apply(x, channels, fill), x)


PAIR_SPECS[_CopyChannels] = ('b--', '', 'b')


# =========================================================================== row ops
class _RowSum(Function):
    """(..., H, W) -> (...) * alpha over the trailing ``nd`` dims."""

    @staticmethod
    def forward(ctx, x, nd, alpha):
        x = x.contiguous()
        lead, tail = x.shape[:-nd], x.shape[-nd:]
        cols = 1
        for s in tail:
            cols *= s
        rows = x.numel() // cols
        out = x.new_empty(lead)
        K().row_sum(x, out, alpha, rows, cols)
        ctx.tail, ctx.alpha = tuple(tail), alpha
        return out

    @staticmethod
    def backward(ctx, g):
        return _RowBcast.apply(g, ctx.tail, ctx.alpha), None, None


class _RowBcast(Function):
    @staticmethod
    def forward(ctx, v, tail, alpha):
        v = v.contiguous()
        out = v.new_empty(tuple(v.shape) + tuple(tail))
        cols = 1
        for s in tail:
            cols *= s
        K().row_bcast(v, out, alpha, v.numel(), cols)
        ctx.nd, ctx.alpha = len(tail), alpha
        return out

    @staticmethod
    def backward(ctx, g):
        return _RowSum.apply(g, ctx.nd, ctx.alpha), None, None


def sum_hw(x):
    """torch.sum(x, [2, 3])"""
    if _is_pair(x):
        return pair_apply(_RowSum, x, 2, 1.0)
    return _RowSum.apply(x, 2, 1.0)


PAIR_SPECS[_RowSum] = ('b--', '', 'b')          # (only for reductions that keep the image dimension)


class _RepeatRows(Function):
    """x.repeat(reps, 1) -- for ``groups`` row blocks back to back, each repeated on its own (tg_repeat_rows_groups)."""

    @staticmethod
    def forward(ctx, x, reps, alpha, groups=1):
        x = x.contiguous()
        rows, cols = x.shape[0] // groups, x.shape[1]
        out = x.new_empty(groups * rows * reps, cols)
        K().repeat_rows_groups(x, out, alpha, rows, cols, reps, groups)
        ctx.reps, ctx.alpha, ctx.groups = reps, alpha, groups
        return out

    @staticmethod
    def backward(ctx, g):
        return _SumReps.apply(g, ctx.reps, ctx.alpha, ctx.groups), None, None, None

    @staticmethod
    def pair_full(fargs):
        return fargs[:3] + [2]

    @staticmethod
    def pair_half(sub):
        sub.groups = 1


class _SumReps(Function):
    @staticmethod
    def forward(ctx, x, reps, alpha, groups=1):
        x = x.contiguous()
        rows, cols = x.shape[0] // (reps * groups), x.shape[1]
        out = x.new_empty(groups * rows, cols)
        K().sum_reps_groups(x, out, alpha, rows, cols, reps, groups)
        ctx.reps, ctx.alpha, ctx.groups = reps, alpha, groups
        return out

    @staticmethod
    def backward(ctx, g):
        return _RepeatRows.apply(g, ctx.reps, ctx.alpha, ctx.groups), None, None, None

    pair_full = _RepeatRows.pair_full
    pair_half = _RepeatRows.pair_half


PAIR_SPECS[_RepeatRows] = ('b---', '', 'b')
PAIR_SPECS[_SumReps] = ('b---', '', 'b')


def repeat_rows(x, reps):
    """x.repeat(reps, 1); a Pair: each half repeated on its own (rows stay [half][rep][row])"""
    if _is_pair(x):
        return pair_apply(_RepeatRows, x, reps, 1.0, 1)
    return _RepeatRows.apply(x, reps, 1.0)


def mean_reps(x, reps):
    """x.reshape(reps, -1, cols).mean(0)"""
    if _is_pair(x):
        return pair_apply(_SumReps, x, reps, 1.0 / reps, 1)
    return _SumReps.apply(x, reps, 1.0 / reps)


# =========================================================================== elementwise
class _Add(Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        out = torch.empty_like(a)
        K().add(a, b, out, a.numel())
        return out

    @staticmethod
    def backward(ctx, g):
        return g, g


def add(a, b):
    if _is_pair(a):
        return pair_apply(_Add, a, b)
    return _Add.apply(a, b)


PAIR_SPECS[_Add] = ('bb', '', 'b')


class _SumN(Function):
    """((a + b) + c) + d for 2-4 same-shaped tensors in one kernel."""

    @staticmethod
    def forward(ctx, *ts):
        ts = [t.contiguous() for t in ts]
        out = torch.empty_like(ts[0])
        c = ts[2] if len(ts) > 2 else None
        d = ts[3] if len(ts) > 3 else None
        K().add4(ts[0], ts[1], c, d, out, out.numel())
        ctx.n = len(ts)
        return out

    @staticmethod
    def backward(ctx, g):
        return (g,) * ctx.n


class _ForkN(Function):
    """x -> n aliases of x, one graph node: its backward sees all n incoming gradients at once and adds them in ONE kernel
    (tg_add4) instead of the autograd engine's n - 1 separate adds.  For a tensor with several consumers inside a block
    (SelfAttention2d: theta, phi, g and the residual all read x)."""
    handles_none_grads = True

    @staticmethod
    def forward(ctx, x, n):
        ctx.set_materialize_grads(False)
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *gs):
        gs = [g for g in gs if g is not None]
        if not gs:
            return None, None
        if len(gs) == 1:
            return gs[0], None
        return _SumN.apply(*gs), None


def fork(x, n):
    """-> n tensors equal to x whose gradients are summed by one kernel (2 <= n <= 4)."""
    if _is_pair(x):
        return mark_data(pair_apply(_ForkN, x, n), x)
    return mark_data(_ForkN.apply(x, n), x)


PAIR_SPECS[_ForkN] = ('b-', '', '*')


class _Mul(Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        out = torch.empty_like(a)
        K().mul(a, b, out, a.numel())
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        ga = _Mul.apply(g, b) if ctx.needs_input_grad[0] else None
        gb = _Mul.apply(g, a) if ctx.needs_input_grad[1] else None
        return ga, gb


def mul(a, b):
    if _is_pair(a) or _is_pair(b):
        return pair_apply(_Mul, a, b)
    return _Mul.apply(a, b)


PAIR_SPECS[_Mul] = ('bb', 'hh', 'b')


class _Scale(Function):
    @staticmethod
    def forward(ctx, x, alpha):
        x = x.contiguous()
        out = torch.empty_like(x)
        K().scale(x, alpha, out, x.numel())
        ctx.alpha = alpha
        return out

    @staticmethod
    def backward(ctx, g):
        return _Scale.apply(g, ctx.alpha), None


def scale(x, alpha):
    return _Scale.apply(x, float(alpha))


class _ScaleDev(Function):
    """(alpha * s) * x with s a 0-d device tensor."""

    @staticmethod
    def forward(ctx, s, x, alpha):
        x = x.contiguous()
        out = torch.empty_like(x)
        K().scale_dev(s, alpha, x, out, x.numel())
        ctx.save_for_backward(s, x)
        ctx.alpha = alpha
        return out

    @staticmethod
    def backward(ctx, g):
        s, x = ctx.saved_tensors
        gs = _Dot.apply(g, x, ctx.alpha).reshape(s.shape) if ctx.needs_input_grad[0] else None
        gx = _ScaleDev.apply(s, g, ctx.alpha) if ctx.needs_input_grad[1] else None
        return gs, gx, None


class _Dot(Function):
    """alpha * sum(a*b) as a 0-d tensor."""

    @staticmethod
    def forward(ctx, a, b, alpha):
        a, b = a.contiguous(), b.contiguous()
        out = a.new_empty(())
        ws = _ws(a, K().reduce_workspace(a.numel()))
        K().dot(a, b, alpha, out, ws, a.numel(), 0)
        ctx.save_for_backward(a, b)
        ctx.alpha = alpha
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = g.contiguous()
        ga = _ScaleDev.apply(g, b, ctx.alpha) if ctx.needs_input_grad[0] else None
        gb = _ScaleDev.apply(g, a, ctx.alpha) if ctx.needs_input_grad[1] else None
        return ga, gb, None


class _ScaleAddDev(Function):
    """s * a + b  (attention.py:35: gamma * o + x)"""

    @staticmethod
    def forward(ctx, s, a, b):
        a, b = a.contiguous(), b.contiguous()
        out = torch.empty_like(a)
        K().scale_add_dev(s, a, b, out, a.numel())
        ctx.save_for_backward(s, a)
        return out

    @staticmethod
    def backward(ctx, g):
        s, a = ctx.saved_tensors
        gs = None
        if ctx.needs_input_grad[0] and (_param_grads_wanted() or not s.is_leaf):
            sink = _grad_sink(s)
            if sink is not None:
                gc = g.contiguous()
                K().dot(gc, a, 1.0, sink, _ws(a, K().reduce_workspace(a.numel())), a.numel(), 1)
            else:
                gs = _Dot.apply(g, a, 1.0).reshape(s.shape)
        ga = _ScaleDev.apply(s, g, 1.0) if ctx.needs_input_grad[1] else None
        return gs, ga, g


def scale_add(s, a, b):
    if _is_pair(a):
        return pair_apply(_ScaleAddDev, s, a, b)
    return _ScaleAddDev.apply(s, a, b)


PAIR_SPECS[_ScaleAddDev] = ('-bb', '-h', 'b')


class _Recip(Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        out = torch.empty_like(x)
        K().recip(x, out, x.numel())
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, g):
        out, = ctx.saved_tensors
        return _Scale.apply(_Mul.apply(_Mul.apply(out, out), g), -1.0)        # d(1/x) = -(1/x)^2


def spectral_normalize(weight, u, v, training, n_power_iterations=1, eps=1e-12):
    """torch.nn.utils.spectral_norm's compute_weight: one power iteration on the (u, v) buffers in place while
    training, then weight / (u^T W v) with u, v treated as constants -- differentiable twice w.r.t. weight."""
    rows = weight.shape[0]
    wmat = weight.reshape(rows, -1)
    cols = wmat.shape[1]
    if training and n_power_iterations > 0:
        with torch.no_grad():
            K().sn_power_iter(wmat.detach().contiguous(), u, v, None, rows, cols, n_power_iterations, float(eps))
        u, v = u.clone(), v.clone()      # a later forward updates the buffers in place; this graph keeps its own copy
    wv = matmul(wmat, v.view(cols, 1)).view(rows)
    sigma = _Dot.apply(u, wv, 1.0)
    return _ScaleDev.apply(_Recip.apply(sigma), weight, 1.0)


class _LReluBwd(Function):
    """g * lrelu'(x); LeakyReLU forward is the case g is x."""

    @staticmethod
    def forward(ctx, g, x, slope):
        g, x = g.contiguous(), x.contiguous()
        out = torch.empty_like(x)
        K().lrelu_bwd(g, x, slope, out, x.numel())
        ctx.save_for_backward(x)
        ctx.slope = slope
        return out

    @staticmethod
    def backward(ctx, v):
        x, = ctx.saved_tensors
        return _LReluBwd.apply(v, x, ctx.slope), None, None


class _LRelu(Function):
    @staticmethod
    def forward(ctx, x, slope):
        x = x.contiguous()
        out = torch.empty_like(x)
        K().lrelu_bwd(x, x, slope, out, x.numel())
        ctx.save_for_backward(x)
        ctx.slope = slope
        return out

    @staticmethod
    def backward(ctx, g):
        x, = ctx.saved_tensors
        return _LReluBwd.apply(g, x, ctx.slope), None


def leaky_relu(x, slope=0.2):
    if _is_pair(x):
        return pair_apply(_LRelu, x, float(slope))
    return _LRelu.apply(x, float(slope))


PAIR_SPECS[_LRelu] = ('b-', 'h', 'b')


# nn.ELU / nn.SELU (reference trainers/cnn.py:42-44).  SELU = ELU with ATen's constants (torch.selu calls
# at::elu(x, alpha, scale)).
SELU_ALPHA = 1.6732632423543772848170429916717
SELU_SCALE = 1.0507009873554804934193349852946


class _Elu(Function):
    @staticmethod
    def forward(ctx, x, alpha, scale):
        x = x.contiguous()
        y = torch.empty_like(x)
        K().elu_fwd(x, alpha, scale, y, x.numel())
        ctx.save_for_backward(x)
        ctx.alpha, ctx.scale = alpha, scale
        return y

    @staticmethod
    def backward(ctx, g):
        x, = ctx.saved_tensors
        return _EluBwd.apply(g, x, ctx.alpha, ctx.scale, 1), None, None


class _EluBwd(Function):
    """order 1: g f'(x) (differentiable once more: the R1 real branch); order 2: g f''(x)."""

    @staticmethod
    def forward(ctx, g, x, alpha, scale, order):
        g, x = g.contiguous(), x.contiguous()
        out = torch.empty_like(x)
        K().elu_bwd(g, x, alpha, scale, order, out, x.numel())
        ctx.save_for_backward(g, x)
        ctx.alpha, ctx.scale, ctx.order = alpha, scale, order
        return out

    @staticmethod
    def backward(ctx, v):
        g, x = ctx.saved_tensors
        a_g = _EluBwd.apply(v, x, ctx.alpha, ctx.scale, ctx.order) if ctx.needs_input_grad[0] else None
        a_x = None
        if ctx.needs_input_grad[1]:
            if ctx.order != 1:
                raise NotImplementedError('third-order ELU derivative')
            a_x = _Mul.apply(v, _EluBwd.apply(g, x, ctx.alpha, ctx.scale, 2))
        return a_g, a_x, None, None, None


def elu(x, alpha=1.0):
    if _is_pair(x):
        return pair_apply(_Elu, x, float(alpha), 1.0)
    return _Elu.apply(x, float(alpha), 1.0)


def selu(x):
    if _is_pair(x):
        return pair_apply(_Elu, x, SELU_ALPHA, SELU_SCALE)
    return _Elu.apply(x, SELU_ALPHA, SELU_SCALE)


PAIR_SPECS[_Elu] = ('b--', 'h', 'b')


class _Tanh(Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        y = torch.empty_like(x)
        K().tanh_fwd(x, y, x.numel())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        y, = ctx.saved_tensors
        return _TanhBwd.apply(g, y)


class _TanhBwd(Function):
    @staticmethod
    def forward(ctx, g, y):
        g, y = g.contiguous(), y.contiguous()
        out = torch.empty_like(y)
        K().tanh_bwd(g, y, out, y.numel())
        ctx.save_for_backward(g, y)
        return out

    @staticmethod
    def backward(ctx, v):
        g, y = ctx.saved_tensors
        a_g = _TanhBwd.apply(v, y) if ctx.needs_input_grad[0] else None
        a_y = None
        if ctx.needs_input_grad[1]:
            a_y = _Scale.apply(_Mul.apply(_Mul.apply(v, g), y), -2.0)
        return a_g, a_y


def tanh(x):
    if _is_pair(x):
        return pair_apply(_Tanh, x)
    return _Tanh.apply(x)


PAIR_SPECS[_Tanh] = ('b', 'h', 'b')


# =========================================================================== softmax
class _Softmax(Function):
    @staticmethod
    def forward(ctx, s):
        s = s.contiguous()
        cols = s.shape[-1]
        y = torch.empty_like(s)
        K().softmax_fwd(s, y, s.numel() // cols, cols)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, gy):
        y, = ctx.saved_tensors
        return _SoftmaxBwd.apply(gy, y)


class _SoftmaxBwd(Function):
    @staticmethod
    def forward(ctx, gy, y):
        gy, y = gy.contiguous(), y.contiguous()
        cols = y.shape[-1]
        gs = torch.empty_like(y)
        K().softmax_bwd(gy, y, gs, y.numel() // cols, cols)
        ctx.save_for_backward(gy, y)
        return gs

    @staticmethod
    def backward(ctx, v):
        gy, y = ctx.saved_tensors
        a_gy = _SoftmaxBwd.apply(v, y) if ctx.needs_input_grad[0] else None
        a_y = None
        if ctx.needs_input_grad[1]:
            a_y = _SoftmaxDbwdY.apply(v, gy, y)
        return a_gy, a_y


class _SoftmaxDbwdY(Function):
    @staticmethod
    def forward(ctx, v, gy, y):
        v = v.contiguous()
        cols = y.shape[-1]
        out = torch.empty_like(y)
        K().softmax_dbwd(v, gy, y, out, y.numel() // cols, cols)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        raise NotImplementedError('third-order softmax derivative')


def softmax_lastdim(s):
    return _Softmax.apply(s)


# =========================================================================== fused attention core
def attention_composed(theta, phi, g):
    """g softmax(theta^T phi)^T from differentiable-twice primitives; materialises the (N x M) map."""
    beta = softmax_lastdim(matmul(theta, phi, transA=True))          # (B, N, M)
    return matmul(g, beta, transB=True)                               # (B, DV, N)


def _bgemm(A, Bm, ta=False, tb=False, out=None):
    """out (+)= op(A) op(B), batched 3-D, no autograd (out given: accumulate into it)."""
    batch, (ar, ac), (br, bc) = A.shape[0], A.shape[1:], Bm.shape[1:]
    M, Kd = (ac, ar) if ta else (ar, ac)
    N = br if tb else bc
    beta = 1.0
    if out is None:
        out, beta = A.new_empty(batch, M, N), 0.0
    K().gemm(A, Bm, out, None, M, N, Kd, ac, bc, N, int(ta), int(tb), batch, ar * ac, br * bc, M * N, beta)
    return out


class _AttnBwd(Function):
    """(dtheta, dphi, dg) of the fused core, itself differentiable once more (what the R1 penalty asks of the
    discriminator's real branch).  Forward: the fused first-order kernels.  Backward: one fused kernel (tg_attn_dbwd) when
    a wave can hold a row of the map; else the (N x M) maps by GEMM, one row-wise kernel for the softmax algebra
    (tg_attn_dbwd_rows) and GEMMs back down to the operand shapes."""
    handles_none_grads = True

    @staticmethod
    def forward(ctx, go, theta, phi, g, o, lse):
        go = go.contiguous()
        B, D, N = theta.shape
        DV, M = g.shape[1], g.shape[2]
        dtheta, dphi, dg = torch.empty_like(theta), torch.empty_like(phi), torch.empty_like(g)
        K().attn_bwd(go, theta, phi, g, o, lse, dtheta, dphi, dg, _ws(go, K().attn_bwd_workspace(B, D, DV, N, M)),
                     B, D, DV, N, M)
        ctx.save_for_backward(go, theta, phi, g, lse)
        ctx.set_materialize_grads(False)
        return dtheta, dphi, dg

    @staticmethod
    @once_differentiable
    def backward(ctx, a, b, c):
        go, theta, phi, g, lse = ctx.saved_tensors
        if a is None and b is None and c is None:
            return None, None, None, None, None, None
        B, D, N = theta.shape
        DV, M = g.shape[1], g.shape[2]
        a, b, c = (None if t is None else t.contiguous() for t in (a, b, c))
        if K().attn_dbwd_supported(D, DV, M):                             # one kernel, the maps stay in registers
            a, b, c = (torch.zeros_like(like) if t is None else t for t, like in ((a, theta), (b, phi), (c, g)))
            d_go, d_theta, d_phi, d_g = (torch.empty_like(t) for t in (go, theta, phi, g))
            K().attn_dbwd(go, theta, phi, g, lse, a, b, c, d_go, d_theta, d_phi, d_g,
                          _ws(go, K().attn_dbwd_workspace(B, D, DV, N, M)), B, D, DV, N, M)
            return d_go, d_theta, d_phi, d_g, None, None
        s = _bgemm(theta, phi, ta=True)                                   # (B, N, M) maps
        gp = _bgemm(go, g, ta=True)
        u = _bgemm(a, phi, ta=True) if a is not None else None
        if b is not None:
            u = _bgemm(theta, b, ta=True, out=u)
        if u is None:
            u = torch.zeros_like(s)
        v = _bgemm(go, c, ta=True) if c is not None else torch.zeros_like(s)
        K().attn_dbwd_rows(s, lse, gp, u, v, B * N, M)
        p, gs, dgp, ds = s, gp, u, v
        d_theta = _bgemm(phi, ds, tb=True)
        if b is not None:
            _bgemm(b, gs, tb=True, out=d_theta)
        d_phi = _bgemm(theta, ds)
        if a is not None:
            _bgemm(a, gs, out=d_phi)
        d_go = _bgemm(g, dgp, tb=True)
        if c is not None:
            _bgemm(c, p, tb=True, out=d_go)
        d_g = _bgemm(go, dgp)
        return d_go, d_theta, d_phi, d_g, None, None


class _AttnCore(Function):
    @staticmethod
    def forward(ctx, theta, phi, g):
        theta, phi, g = theta.contiguous(), phi.contiguous(), g.contiguous()
        B, D, N = theta.shape
        DV, M = g.shape[1], g.shape[2]
        o = theta.new_empty(B, DV, N)
        lse = theta.new_empty(B, N)
        K().attn_fwd(theta, phi, g, o, lse, B, D, DV, N, M)
        ctx.save_for_backward(theta, phi, g, o, lse)
        return o

    @staticmethod
    def backward(ctx, go):
        theta, phi, g, o, lse = ctx.saved_tensors
        # under create_graph=True (R1 penalty on the real branch) this node lands in the graph and is differentiated again
        return _AttnBwd.apply(go, theta, phi, g, o, lse)


def attention_core(theta, phi, g):
    """theta (B,D,N), phi (B,D,M), g (B,DV,M) -> (B,DV,N).  Fused kernel when the head dims are compiled in."""
    fused = K().attn_supported(theta.shape[1], g.shape[1])
    if _is_pair(theta):
        if fused:
            return pair_apply(_AttnCore, theta, phi, g)
        return Pair(attention_composed(theta.r, phi.r, g.r), attention_composed(theta.f, phi.f, g.f))
    if fused:
        return _AttnCore.apply(theta, phi, g)
    return attention_composed(theta, phi, g)


PAIR_SPECS[_AttnCore] = ('bbb', 'hhhhh', 'b')


# =========================================================================== IQN / losses
def iqn_cos_embed(taus, embedding_range):
    """cos((tau*pi)*range) -- no gradient (taus are sampled, range is a buffer)."""
    if _is_pair(taus):
        both = iqn_cos_embed(_join(taus.r.contiguous(), taus.f.contiguous()), embedding_range)
        n = both.shape[0] // 2
        return Pair(both[:n], both[n:])
    taus = taus.contiguous()
    n, dims = taus.shape[0], embedding_range.shape[0]
    out = taus.new_empty(n, dims)
    K().iqn_cos_embed(taus, embedding_range, out, n, dims)
    return out


class _ScaledGrad(Function):
    """Common backward of the scalar losses: d loss / d preds was produced by
    the forward kernel; the backward only scales it by the upstream scalar."""

    @staticmethod
    def _finish(ctx, preds, dpreds):
        ctx.save_for_backward(dpreds)
        ctx.shape = preds.shape

    @staticmethod
    def _bwd(ctx, g):
        dpreds, = ctx.saved_tensors
        out = torch.empty_like(dpreds)
        K().scale_dev(g.contiguous(), 1.0, dpreds, out, dpreds.numel())
        return out.view(ctx.shape)


class _IQNLoss(Function):
    """models/iqn.py:111-130; ``groups`` evaluations back to back: the SUM of their losses (tg_iqn_loss_groups)."""

    @staticmethod
    def forward(ctx, preds, target, taus, num_quantiles, k, groups=1):
        pc, target, taus = preds.contiguous(), target.contiguous(), taus.contiguous()
        B = target.shape[0] // groups
        loss = pc.new_empty(())
        dpreds = torch.empty_like(pc)
        ws = _ws(pc, K().reduce_workspace(pc.numel()))
        K().iqn_loss_groups(pc, target, taus, float(k), loss, dpreds, ws, num_quantiles, B, groups)
        _ScaledGrad._finish(ctx, preds, dpreds)
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        return _ScaledGrad._bwd(ctx, g), None, None, None, None, None

    @staticmethod
    def pair_full(fargs):
        return fargs[:5] + [2]


PAIR_SPECS[_IQNLoss] = ('b-b---', 'h', '-')


def iqn_quantile_huber_loss(preds, target, taus, num_quantiles, k=1.0):
    """models/iqn.py:111-130 for out_dims == 1 (preds/taus (Q*B,1), row = q*B + b).  ``preds`` / ``taus`` Pairs (``target``: the
    (2B, 1) targets of both halves): loss_real + loss_fake (trainers/iqn.py:118-120) from one launch."""
    if _is_pair(preds):
        return pair_apply(_IQNLoss, preds, target, taus, num_quantiles, k, 1)
    return _IQNLoss.apply(preds, target, taus, num_quantiles, k)


class _BCELogits(Function):
    @staticmethod
    def forward(ctx, logits, targets):
        lc, targets = logits.contiguous(), targets.contiguous()
        loss = lc.new_empty(())
        dl = torch.empty_like(lc)
        ws = _ws(lc, K().reduce_workspace(lc.numel()))
        K().bce_logits(lc, targets, loss, dl, ws, lc.numel())
        _ScaledGrad._finish(ctx, logits, dl)
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        return _ScaledGrad._bwd(ctx, g), None


def bce_with_logits(logits, targets):
    """nn.BCEWithLogitsLoss() (mean reduction).  ``logits`` may be a Pair: the loss over the concatenated halves
    (trainers/cnn.py:124-131 concatenates p_real and p_fake), without the concatenation."""
    if _is_pair(logits):
        return pair_apply(_BCELogits, logits, targets)
    return _BCELogits.apply(logits, targets)


PAIR_SPECS[_BCELogits] = ('b-', 'h', '-')


class _SumSq(Function):
    @staticmethod
    def forward(ctx, x, alpha):
        x = x.contiguous()
        out = x.new_empty(())
        ws = _ws(x, K().reduce_workspace(x.numel()))
        K().sumsq(x, alpha, out, ws, x.numel())
        ctx.save_for_backward(x)
        ctx.alpha = alpha
        return out

    @staticmethod
    def backward(ctx, g):
        x, = ctx.saved_tensors
        return _ScaleDev.apply(g.contiguous(), x, 2.0 * ctx.alpha), None


def sumsq(x, alpha=1.0):
    return _SumSq.apply(x, float(alpha))
