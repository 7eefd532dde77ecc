"""Hot-path half of the reference's ``Trainer`` (trainers/trainer.py).

In scope: the latent / batch assembly helpers
``sample_z / sample_g / make_adversarial_batch / make_generator_batch``
(trainer.py:153-176) and the hyper-parameter flags of the step
(trainer.py:268-313).  The epoch loop, datasets, callbacks and filesystem
handling of the reference stay with the reference: a tartangan ``Trainer`` calls
``train_batch(images)`` once per DataLoader batch (trainer.py:96) and that call
is the drop-in boundary.
"""
import argparse

import torch

from .utils import set_device_from_args


class RngFeed:
    """Host-side random inputs of one step (latents z, IQN quantile fractions tau).

    The reference draws them from the CPU default generator inside the step and moves them
    to the device (trainer.py:153-156, models/iqn.py:105-108).  To replay the step from
    captured HIP graphs the draws have to happen before the launch, into static device
    buffers, in exactly the reference's order.  The first (eager) step records that order;
    afterwards ``refill`` reproduces it.  Under data parallelism every rank draws the GLOBAL
    tensor from the same seed and keeps its own rows (tau rows are quantile-major:
    row = q * B + b), so the union over ranks is the single-process stream.
    """

    def __init__(self, device, rank=0, world=1):
        self.device, self.rank, self.world = device, rank, world
        self.plan = []          # [(kind, local_rows, cols)]
        self.static = []
        self.host = []
        self.mode = 'record'
        self.cursor = 0
        self.key = None         # batch shape the plan was recorded for
        self._spec = None       # (generator state before, after) a speculative draw of the next step's inputs
        self._uploaded = None   # event behind the last host -> device copies of the staging buffers
        self._arena = self._host_arena = None   # one allocation behind all of an adopted plan's buffers

    def _draw_cpu(self, kind, rows, cols):
        if kind == 'z':
            full = torch.randn(rows * self.world, cols)
            return full[self.rank * rows:(self.rank + 1) * rows]
        # tau: (Q*B_global, 1) with row = q*B_global + b  ->  this rank's (Q*B, 1)
        q = cols
        full = torch.rand(rows * self.world, 1)
        b_local = rows // q
        return full.view(q, b_local * self.world)[:, self.rank * b_local:(self.rank + 1) * b_local].reshape(rows, 1)

    def draw(self, kind, rows, cols):
        """kind 'z': (rows, cols) normal;  kind 'tau': (rows, 1) uniform with cols = num_quantiles."""
        if self.mode == 'off':
            return self._draw_cpu(kind, rows, cols).contiguous().to(self.device)
        if self.mode == 'record':
            val = self._draw_cpu(kind, rows, cols).contiguous()
            buf = val.to(self.device)
            self.plan.append((kind, rows, cols))
            self.static.append(buf)
            host = torch.empty_like(val)
            self.host.append(host.pin_memory() if buf.is_cuda else host)
            return buf
        buf = self.static[self.cursor]          # 'serve'
        assert self.plan[self.cursor] == (kind, rows, cols), 'step structure changed since it was recorded'
        self.cursor += 1
        return buf

    def adopt(self, plan):
        """Start from a plan that is known in advance (the stock trainers' own draw order) instead of recording it during a
        first inline-drawing step: every step, the first included, then has the same structure.

        All of the plan's buffers are slices of ONE device arena (and one pinned staging arena): a step uploads its random
        inputs with a single copy, and the latents lie back to back -- whatever else is drawn between them -- so that the two
        generator passes of a step run over them as one (2B, latent) tensor without joining them first (functional._join)."""
        self.plan = [tuple(p) for p in plan]
        sizes = [rows * (cols if kind == 'z' else 1) for kind, rows, cols in self.plan]
        order = sorted(range(len(self.plan)), key=lambda i: (self.plan[i][0] != 'z', i))
        offsets, total = {}, 0
        for i in order:
            offsets[i] = total
            total += (sizes[i] + 3) // 4 * 4                      # 16-byte aligned slices
        self._arena = torch.zeros(max(total, 1), dtype=torch.float32, device=self.device)
        host = torch.zeros(max(total, 1), dtype=torch.float32)
        self._host_arena = host.pin_memory() if self._arena.is_cuda else host
        self.static, self.host = [], []
        for i, (kind, rows, cols) in enumerate(self.plan):
            width = cols if kind == 'z' else 1
            self.static.append(self._arena[offsets[i]:offsets[i] + sizes[i]].view(rows, width))
            self.host.append(self._host_arena[offsets[i]:offsets[i] + sizes[i]].view(rows, width))
        self.cursor = 0

    def _draw_into_host(self):
        for (kind, rows, cols), host in zip(self.plan, self.host):
            if self.world == 1:
                # straight into the pinned staging buffer: ``torch.randn(r, c)`` is ``empty(r, c).normal_()`` and
                # ``torch.rand(r, 1)`` is ``empty(r, 1).uniform_()`` -- the same values from the same generator state
                host.normal_() if kind == 'z' else host.uniform_()
            else:
                host.copy_(self._draw_cpu(kind, rows, cols))

    def prefetch(self):
        """Draw the NEXT step's inputs while the GPU is busy with this one -- invisibly.  The CPU generator is serial: 63 us per
        (64, 256) latent, and a rank of an N-GPU run draws the GLOBAL tensors (N times that: ~1 ms per step at 8 GPUs), all of
        it GPU idle time if done between a step's loss read-back and the next step's first launch.  So, after a step's work
        has been launched: remember the generator state, draw the plan into the staging buffers, remember the state after,
        and PUT THE GENERATOR BACK.  ``refill`` adopts the values only if the generator is still exactly where it was left
        (then drawing now would produce the same values and end in the remembered state); if anything drew or re-seeded in
        between -- a sampler's ``sample_z(4)``, ``torch.manual_seed`` -- those draws saw the untouched stream, the
        speculation is dropped and the step draws as usual."""
        self._spec = None
        if not self.plan:
            return
        if self._uploaded is not None:
            self._uploaded.synchronize()          # the staging buffers are free (the copies ran at the head of this step)
        before = torch.get_rng_state()
        self._draw_into_host()
        after = torch.get_rng_state()
        torch.set_rng_state(before)
        self._spec = (before, after)

    def refill(self):
        spec, self._spec = self._spec, None
        if spec is not None and torch.equal(torch.get_rng_state(), spec[0]):
            torch.set_rng_state(spec[1])
        else:
            if self._uploaded is not None:
                self._uploaded.synchronize()
            self._draw_into_host()
        if self._arena is not None:
            self._arena.copy_(self._host_arena, non_blocking=True)       # the whole plan in one upload
        else:
            for buf, host in zip(self.static, self.host):
                buf.copy_(host, non_blocking=True)
        if self.static and self.static[0].is_cuda:
            self._uploaded = torch.cuda.Event()
            self._uploaded.record()
        self.cursor = 0


class Trainer:
    def __init__(self, args, components=()):
        self.args = args
        if not hasattr(args, 'device'):
            set_device_from_args(args)
        self.components = list(components)
        self.steps = 0
        self.epoch = 1
        self.data_parallel = None
        self.rng_feed = RngFeed(self.device)
        self.rng_feed.mode = 'off'
        self._in_step = False        # True only inside train_batch (and graph capture): draws then go through the feed

    # ------------------------------------------------------------------ hot-path helpers
    @property
    def device(self):
        return self.args.device

    def sample_z(self, n=None):
        """Latents come from the CPU default generator, then move (trainer.py:153-156):
        a seed gives the same z on any device."""
        if n is None:
            n = self.args.batch_size
        return self._draw('z', n, self.gan_config.latent_dims)

    def _draw(self, kind, rows, cols):
        """Random inputs of the step go through ``RngFeed`` (recorded plan, static buffers, per-rank slices).  A draw made
        OUTSIDE a step -- a component's ``sample_z(32)`` at train begin, a user's ``trainer.d(x)`` -- is the reference's
        plain ``torch.randn(n, latent).to(device)`` (trainer.py:153-156) / ``torch.rand(rows, 1)`` (iqn.py:105-108): a fresh
        tensor, never part of the recorded plan, the same values on every rank."""
        if self._in_step:
            return self.rng_feed.draw(kind, rows, cols)
        val = torch.randn(rows, cols) if kind == 'z' else torch.rand(rows, 1)
        return val.to(self.device)

    def sample_g(self, n=None, target_g=False, **g_kwargs):
        z = self.sample_z(n)
        return (self.target_g if target_g else self.g)(z, **g_kwargs)

    def make_adversarial_batch(self, real_data, **g_kwargs):
        generated = self.sample_g(len(real_data), **g_kwargs)
        batch = torch.cat([real_data, generated], dim=0)
        labels = torch.zeros(len(batch), 1, device=self.device)
        labels[:len(labels) // 2] = 1
        return batch, labels

    def make_generator_batch(self, real_data, **g_kwargs):
        generated = self.sample_g(len(real_data), **g_kwargs)
        return generated, torch.ones(len(generated), 1, device=self.device)

    def build_models(self):
        raise NotImplementedError

    # ------------------------------------------------------------------ what the components read (trainer.py:192-216)
    def get_state(self):
        return dict(epoch=self.epoch, steps=self.steps)

    def set_state(self, state):
        for key, value in state.items():
            setattr(self, key, value)

    @property
    def run_id(self):
        return getattr(self.args, 'run_id', None) or 'run'

    @property
    def output_root(self):
        return f"{getattr(self.args, 'output', './output')}/{self.run_id}"

    def attach(self, *components):
        """Give components their back-reference, like ``ComponentContainer.add_components`` (trainer.py:43-50)."""
        for c in components:
            c.trainer = self
            self.components.append(c)
        return self

    def train_batch(self, imgs):
        raise NotImplementedError

    # ------------------------------------------------------------------ flags of the step
    @classmethod
    def add_args_to_parser(cls, p):
        p.add_argument('--batch-size', type=int, default=128)
        p.add_argument('--lr-g', type=float, default=1e-4)
        p.add_argument('--lr-d', type=float, default=4e-4)
        p.add_argument('--lr-target-g', type=float, default=1e-3)
        p.add_argument('--no-cuda', action='store_true')
        p.add_argument('--grad-penalty', type=float, default=5.)
        p.add_argument('--config', default='64')
        p.add_argument('--model-scale', type=float, default=1.)
        p.add_argument('--g-base', default='mlp')
        p.add_argument('--norm', default='bn')
        p.add_argument('--activation', default='relu')

    @classmethod
    def default_args(cls, **overrides):
        p = argparse.ArgumentParser()
        cls.add_args_to_parser(p)
        args = p.parse_args([])
        for k, v in overrides.items():
            setattr(args, k, v)
        return args
