"""Flat-buffer Adam and EMA for the G+D step.

The reference steps ``torch.optim.Adam(betas=(0., 0.999))`` over 32-59 small
tensors per network and updates the EMA ("target") generator tensor by tensor
(trainers/cnn.py:84-85,158-165): some 300 tiny launches per step.  Here every
network's parameters live in ONE contiguous fp32 buffer (the ``nn.Parameter``s
are views into it, their ``.grad``s views into a twin buffer), so Adam is one
elementwise kernel, the EMA is one kernel, ``zero_grad`` is one fill and a
data-parallel gradient all-reduce is one RCCL call on one bucket.
"""
import math

import torch

from . import backend as _be


ALIGN = 4      # floats: every parameter starts on a 16-byte boundary of the bucket


def param_offsets(params):
    """-> ([offset of each parameter in floats], bucket length).  Offsets are multiples of 4 floats: the conv kernels fetch
    filters in 16-byte chunks (LDS-DMA, float4), and a 1- or 3-element parameter in front (the attention gamma, the to-RGB
    bias) would otherwise knock every later tensor of the network off that alignment.  The padding floats stay zero for
    ever: zero gradient, so Adam and the EMA leave them alone, and they add nothing to an all-reduce."""
    offs, off = [], 0
    for p in params:
        offs.append(off)
        off += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
    return offs, off


def pack(tensors, like):
    """Per-parameter tensors -> one bucket-shaped tensor on ``like``'s device (e.g. torch.optim.Adam moments ->
    ``FusedAdam.load_state_dict``)."""
    offs, n = param_offsets(tensors)
    flat = torch.zeros(n, dtype=torch.float32, device=like.device)
    for t, o in zip(tensors, offs):
        flat[o:o + t.numel()].copy_(t.reshape(-1))
    return flat


def _param_list(module):
    """``list(module.parameters())``, cached on the module: the recursive walk costs ~100 us for these networks, and a
    replayed step checks its bindings several times."""
    cached = module.__dict__.get('_tg_params')
    key = (len(module._modules), len(module._parameters))
    if cached is not None and cached[0] == key and cached[2][0] < 64:
        cached[2][0] += 1
        return cached[1]
    # (a Parameter OBJECT replaced deep inside the tree -- ``setattr(m, 'weight', new)``; ``module.to()`` and
    # ``load_state_dict`` keep the objects -- is noticed at the latest after 64 uses, when the walk is taken again)
    params = list(module.parameters())
    module.__dict__['_tg_params'] = (key, params, [0])
    return params


def invalidate_param_cache(module):
    module.__dict__.pop('_tg_params', None)


def flatten_parameters(module):
    """Re-home ``module``'s parameters into one flat buffer; returns (flat_params, flat_grads).

    Parameter objects are preserved (same identity, order, names, state_dict keys);
    ``load_state_dict`` / in-place updates keep working because they copy into the views.
    """
    params = _param_list(module)
    cached = getattr(module, '_tg_flat', None)
    if cached is not None and _is_bound(params, *cached):
        return cached
    invalidate_param_cache(module)                 # about to rebuild: take a fresh walk of the module tree
    params = _param_list(module)
    # first call, or the binding was broken from outside: module.to(), zero_grad(set_to_none=True), p.grad = None,
    # torch.save(module) / torch.load (the reference's ModelCheckpoint pickles whole models,
    # components/model_checkpoint.py:35-45) all leave parameters that are no longer views of the buckets.
    # Rebuild from the CURRENT parameter (and gradient) values.
    offs, n = param_offsets(params)
    dev = params[0].device
    flat = torch.zeros(n, dtype=torch.float32, device=dev)
    grads = torch.zeros(n, dtype=torch.float32, device=dev)
    with torch.no_grad():
        for p, off in zip(params, offs):
            k = p.numel()
            flat[off:off + k].copy_(p.detach().reshape(-1))
            if p.grad is not None:
                grads[off:off + k].copy_(p.grad.detach().reshape(-1))
            p.data = flat[off:off + k].view(p.shape)
            p.grad = grads[off:off + k].view(p.shape)
    module._tg_flat = (flat, grads)
    return flat, grads


def _is_bound(params, flat, grads):
    """Host-side check (pointer compares only) that every parameter / gradient is still the view it was made."""
    offs, n = param_offsets(params)
    fp, gp = flat.data_ptr(), grads.data_ptr()
    for p, off in zip(params, offs):
        if p.device != flat.device or p.data_ptr() != fp + 4 * off:
            return False
        g = p.grad
        if g is None or g.data_ptr() != gp + 4 * off or not g.is_contiguous():
            return False
    return n == flat.numel()


class FusedAdam:
    """torch.optim.Adam semantics (no weight decay / amsgrad) on one flat buffer.

    The step-dependent scalars (lr / bias_correction1, sqrt(bias_correction2)) are
    computed on the host in double precision like torch's single-tensor path and
    handed to the kernel through a 6-float device buffer, so a captured HIP graph
    of the step stays valid for every step index.
    """

    def __init__(self, module, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.module = module
        self.flat, self.grads = flatten_parameters(module)
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.step_count = 0
        self.generation = 0          # bumped whenever the buckets had to be rebuilt (captured graphs are then stale)
        self._make_hyper()

    def _make_hyper(self):
        self._hyper = torch.zeros(6, dtype=torch.float32, device=self.flat.device)
        self._hyper_host = torch.zeros(6, dtype=torch.float32)
        if self.flat.is_cuda:
            self._hyper_host = self._hyper_host.pin_memory()
        self._hyper_np = self._hyper_host.numpy()            # (one vectorised host write per step instead of six tensor setitems)

    def ensure_bound(self):
        """Re-home the parameters if something outside detached them from the buckets (see flatten_parameters); without
        this, autograd would allocate fresh ``.grad`` tensors and Adam would step on a stale zero bucket."""
        flat, grads = flatten_parameters(self.module)
        if flat is not self.flat or grads is not self.grads:
            if flat.numel() != self.exp_avg.numel():
                raise RuntimeError('FusedAdam: the module gained or lost parameters after the optimiser was built')
            self.flat, self.grads = flat, grads
            if self.exp_avg.device != flat.device:
                self.exp_avg, self.exp_avg_sq = self.exp_avg.to(flat.device), self.exp_avg_sq.to(flat.device)
                self._make_hyper()
            self.generation += 1
        return self.generation

    def __getstate__(self):
        state = dict(self.__dict__)
        state.pop('_hyper_host', None)       # pinned host memory does not pickle as pinned; rebuilt on load
        state.pop('_hyper_np', None)
        state.pop('_hyper', None)
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)
        self.flat, self.grads = flatten_parameters(self.module)     # unpickled parameters are plain tensors again
        self._make_hyper()

    def zero_grad(self, set_to_none=False):
        self.ensure_bound()
        _be.get().fill(self.grads, 0.0, self.grads.numel())

    def advance(self, checked=False):
        """Host half of a step: bump the step index and upload its scalars (not capturable).  ``checked``: the caller has
        just run ``ensure_bound()`` itself."""
        if not checked:
            self.ensure_bound()
        self.step_count += 1
        b1, b2 = self.betas
        bc1 = 1 - b1 ** self.step_count
        bc2 = 1 - b2 ** self.step_count
        self._hyper_np[:] = (self.lr / bc1, math.sqrt(bc2), b1, b2, 1 - b1, 1 - b2)
        self._hyper.copy_(self._hyper_host, non_blocking=True)

    def apply(self):
        """Device half of a step: one kernel over the flat buffer (capturable)."""
        _be.get().adam_step(self.flat, self.grads, self.exp_avg, self.exp_avg_sq, self._hyper,
                            self.eps, self.flat.numel())

    def step(self):
        self.advance()
        self.apply()

    def state_dict(self):
        return dict(step=self.step_count, lr=self.lr, betas=self.betas, eps=self.eps,
                    exp_avg=self.exp_avg.clone(), exp_avg_sq=self.exp_avg_sq.clone())

    def load_state_dict(self, sd):
        """Accepts this class's own ``state_dict()`` and the stock ``torch.optim.Adam`` layout ``{'state': {i: {'step',
        'exp_avg', 'exp_avg_sq'}}, 'param_groups': [{'lr', 'betas', 'eps', 'params'}]}`` -- what ``cp_model.state_dict()``
        returns for the ``opt_d.pt`` / ``opt_g.pt`` a reference run pickled (components/model_checkpoint.py:39-45): the
        per-parameter moments are packed into the flat buckets in parameter order.  Validated before anything is
        written, so a layout this optimiser cannot take leaves it untouched."""
        if 'param_groups' in sd and 'state' in sd:
            sd = self._from_torch_adam(sd)
        step, exp_avg, exp_avg_sq = int(sd['step']), sd['exp_avg'], sd['exp_avg_sq']
        if exp_avg.numel() != self.exp_avg.numel() or exp_avg_sq.numel() != self.exp_avg_sq.numel():
            raise ValueError(f'FusedAdam.load_state_dict: moments of {exp_avg.numel()} floats for a bucket of {self.exp_avg.numel()}')
        self.step_count = step
        self.lr = float(sd.get('lr', self.lr))
        self.betas = tuple(float(b) for b in sd.get('betas', self.betas))
        self.eps = float(sd.get('eps', self.eps))
        self.exp_avg.copy_(exp_avg)
        self.exp_avg_sq.copy_(exp_avg_sq)

    def _from_torch_adam(self, sd):
        groups = sd['param_groups']
        if len(groups) != 1:
            raise ValueError('FusedAdam.load_state_dict: one parameter group expected (the trainers build Adam over model.parameters())')
        group, state = groups[0], sd['state']
        if group.get('amsgrad') or group.get('weight_decay'):
            raise ValueError('FusedAdam.load_state_dict: amsgrad / weight_decay are outside the hot path (trainers/cnn.py:84-85)')
        params = list(self.module.parameters())
        ids = list(group['params'])
        if len(ids) != len(params):
            raise ValueError(f'FusedAdam.load_state_dict: {len(ids)} parameters in the checkpoint, {len(params)} in the module')
        zeros = [torch.zeros_like(p, device='cpu') for p in params]
        avg, sq, steps = list(zeros), list(zeros), set()
        for k, (pid, p) in enumerate(zip(ids, params)):
            st = state.get(pid)
            if st is None:                       # a parameter that never received a gradient has no state yet
                continue
            if tuple(st['exp_avg'].shape) != tuple(p.shape):
                raise ValueError(f'FusedAdam.load_state_dict: parameter {k} has shape {tuple(p.shape)}, its moments {tuple(st["exp_avg"].shape)}')
            avg[k], sq[k] = st['exp_avg'].detach().float().cpu(), st['exp_avg_sq'].detach().float().cpu()
            steps.add(int(st['step']))
        if len(steps) > 1:
            raise ValueError(f'FusedAdam.load_state_dict: per-parameter step counts differ ({sorted(steps)}); one flat bucket has one step')
        lr = group['lr']
        return dict(step=steps.pop() if steps else 0, lr=float(lr), betas=tuple(group['betas']), eps=group['eps'],
                    exp_avg=pack(avg, self.flat), exp_avg_sq=pack(sq, self.flat))


def ema_update(target_module, source_module, lr):
    """target += (source - target) * lr over all parameters, one kernel."""
    t, _ = flatten_parameters(target_module)
    s, _ = flatten_parameters(source_module)
    _be.get().ema(t, s, float(lr), t.numel())
