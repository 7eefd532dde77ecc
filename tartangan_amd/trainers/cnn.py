"""SA-GAN trainer: drop-in for ``tartangan.trainers.cnn.CNNTrainer``
(reference trainers/cnn.py:28-165) on the HIP engine.

``build_models`` assembles G / target-G / D through the same factory API and
``train_batch(imgs) -> {g_loss, d_loss, gp}`` performs the reference step
 -- D phase with BCE-with-logits and the R1 penalty (second
order), Adam(0, .999), G phase, EMA of G -- with identical op order, RNG
consumption order, BatchNorm mode and label layout.
"""
import functools

import numpy as np
import torch
from torch import nn

from .. import backend as _be
from .. import functional as TF
from ..models.blocks import (
    DiscriminatorOutput, GeneratorInputMLP, GeneratorOutput, ResidualDiscriminatorBlock,
    ResidualGeneratorBlock, TiledZGeneratorInput,
)
from ..models.layers import ELU, SELU, BatchNorm2d, LeakyReLU
from ..models.losses import gradient_penalty
from ..models.pluggan import GAN_CONFIGS, Discriminator, Generator
from ..optim import FusedAdam, ema_update
from .trainer import Trainer
from .utils import toggle_grad


_PREFETCH_RNG = __import__('os').environ.get('TG_PREFETCH_RNG', '1') != '0'      # (development knob; args.prefetch_rng overrides)


class CNNTrainer(Trainer):
    discriminator_class = Discriminator
    d_output_class = DiscriminatorOutput
    # --activation (cnn.py:41-45); an unknown name is a KeyError, as in the reference
    activations = {'relu': functools.partial(LeakyReLU, 0.2), 'selu': SELU, 'elu': ELU}

    # ------------------------------------------------------------------ model assembly
    def _factories(self):
        norm = {'id': nn.Identity, 'bn': BatchNorm2d}[self.args.norm]
        g_input = {'mlp': GeneratorInputMLP, 'tiledz': TiledZGeneratorInput}[self.args.g_base]
        act = self.activations[self.args.activation]
        bind = functools.partial
        return dict(
            g_input=bind(g_input, activation_factory=act),
            g_block=bind(ResidualGeneratorBlock, norm_factory=norm, activation_factory=act),
            d_block=bind(ResidualDiscriminatorBlock, norm_factory=norm, activation_factory=act),
            g_output=bind(GeneratorOutput, norm_factory=norm, activation_factory=act),
            d_output=bind(self.d_output_class, norm_factory=norm, activation_factory=act),
        )

    def build_models(self):
        config = self.args.config
        self.gan_config = GAN_CONFIGS[config] if isinstance(config, str) else config
        self.gan_config = self.gan_config.scale_model(self.args.model_scale)
        f = self._factories()

        def make_g():
            return Generator(self.gan_config, input_factory=f['g_input'], block_factory=f['g_block'],
                             output_factory=f['g_output']).to(self.device)
        # construction order g, target_g, d consumes the init RNG like the reference (cnn.py:66-83)
        self.g = make_g()
        self.target_g = make_g()
        self.d = self.discriminator_class(self.gan_config, block_factory=f['d_block'],
                                          output_factory=f['d_output']).to(self.device)
        self.optimizer_g = FusedAdam(self.g, lr=self.args.lr_g, betas=(0., 0.999))
        self.optimizer_d = FusedAdam(self.d, lr=self.args.lr_d, betas=(0., 0.999))
        if self.args.activation == 'selu':
            self.init_params_selu(self.g.parameters())
            self.init_params_selu(self.d.parameters())
        self.update_target_generator(1.)     # the reference's "copy weights" (see below)

    @torch.no_grad()
    def init_params_selu(self, params):
        """cnn.py:97-105 / iqn.py:94-102: vectors zeroed, matrices and filters N(0, 1/fan_in).  Drawn from the CPU
        default generator in parameter order (what the reference does on its CPU device), then moved, so that a seed
        gives the same initial weights on any device.  A 0-d parameter (the attention gamma) raises ValueError from
        ``_calculate_fan_in_and_fan_out``, exactly as the reference does for --activation selu with attention."""
        for p in params:
            if p.dim() == 1:
                p.zero_()
            else:
                fan_in, _ = nn.init._calculate_fan_in_and_fan_out(p)
                p.copy_(torch.empty(p.shape).normal_(std=float(np.sqrt(1. / fan_in))))

    # ------------------------------------------------------------------ the step
    def _d_losses(self, real, fake, labels):
        """-> (p_real, d_loss without penalty); BCE over the concatenated logits (cnn.py:122-131).  The two evaluations of
        D share every kernel launch when the model allows it (``functional.Pair``)."""
        if self._d_pairable():
            p = self.d(TF.Pair(real, fake))
            return p.r, TF.bce_with_logits(p, labels)
        p_real = self.d(real)
        p_fake = self.d(fake)
        return p_real, TF.bce_with_logits(torch.cat([p_real, p_fake], dim=0), labels)

    def _d_pairable(self):
        """May D(real) and D(fake) of the D phase run as ONE pass over 2B images?  Yes when every module of the discriminator
        is one of this package's (their forwards understand ``Pair``; a foreign block plugged in through the factory API gets
        the reference's two separate calls), BatchNorm statistics are local (SyncBN keeps the separate passes), and
        ``args.pair_d`` (default on) does not say otherwise."""
        if not getattr(self.args, 'pair_d', True):
            return False
        cached = getattr(self, '_pair_ok', None)
        key = (id(self.d), self.data_parallel is not None and self.data_parallel.sync_bn)
        if cached is not None and cached[0] == key:
            return cached[1]
        ok = True
        for m in self.d.modules():
            mod = type(m).__module__
            if not (mod.startswith('tartangan_amd.') or type(m) in (nn.Sequential, nn.Identity, nn.ModuleList)):
                ok = False
            if isinstance(m, BatchNorm2d) and m.sync_group is not None:
                ok = False
            if isinstance(m, ResidualDiscriminatorBlock) and not m.default_resampling():
                ok = False
        self._pair_ok = (key, ok)
        return ok

    def _g_loss(self, fake, ones):
        return TF.bce_with_logits(self.d(fake), ones)

    def _d_phase(self, imgs):
        """Discriminator half up to and including d_loss.backward() (cnn.py:112-136).
        Returns (d_loss, d_grad_penalty or None) as device tensors."""
        bs = len(imgs)
        both_fakes = self._generator_forward_for_both_phases(bs)
        toggle_grad(self.g, False)
        toggle_grad(self.d, True)
        self.optimizer_d.zero_grad()
        if both_fakes is None:
            fake = self.sample_g(bs)
        else:
            fake, self._fake_for_g_phase = both_fakes
        labels = self._labels(bs)
        fake = fake.detach()
        if self._d_pairable() and fake.shape == imgs.shape:
            # real | fake back to back in ONE buffer: every paired op then sees its input as an alias, not a copy
            both = imgs.new_empty((2 * bs,) + tuple(imgs.shape[1:]))
            both[:bs].copy_(imgs)
            both[bs:].copy_(fake)
            imgs, fake = both[:bs], both[bs:]
        real = imgs.detach()
        if self.args.grad_penalty:
            real = real.requires_grad_()
        with TF.filter_forms(self.d):   # the discriminator's parameters are fixed until its optimiser step
            p_real, d_loss = self._d_losses(real, fake, labels)
            d_grad_penalty = None
            if self.args.grad_penalty:
                d_grad_penalty = TF.scale(gradient_penalty(p_real, real), self.args.grad_penalty)
                d_loss = TF.add(d_loss, d_grad_penalty)
            # same parameter gradients as d_loss.backward(); naming the leaves just spares autograd the gradient
            # w.r.t. the real images, which the reference computes (real.requires_grad_) and never reads
            with TF.deferred_wgrad(), TF.params_only():     # one launch finishes all conv weight-gradient reductions of this pass
                torch.autograd.backward(d_loss, inputs=[p for p in self.d.parameters() if p.requires_grad])
        return d_loss.detach(), (d_grad_penalty.detach() if d_grad_penalty is not None else None)

    def _labels(self, bs):
        """(2 bs, 1): ones for the real half, zeros for the fake half (cnn.py:118-121); its first half doubles as the
        all-ones target of the generator phase (:146).  Constant per batch size: built once, not refilled every step."""
        cache = self.__dict__.setdefault('_label_cache', {})
        key = (bs, str(self.device))
        if key not in cache:
            labels = torch.zeros(2 * bs, 1, device=self.device)
            labels[:bs] = 1
            cache[key] = labels
        return cache[key]

    def _g_forward(self, bs):
        """First part of the generator half (cnn.py:139-143): the generator's forward pass.  It reads no discriminator
        weight, so in a data-parallel run it is issued BEFORE the discriminator's optimiser step and hides the
        all-reduce of the discriminator's gradient bucket (same numbers: nothing here depends on that step)."""
        toggle_grad(self.g, True)
        toggle_grad(self.d, False)
        self.optimizer_g.zero_grad()
        fake = self.__dict__.pop('_fake_for_g_phase', None)
        if fake is not None:
            self.sample_z(bs)          # this phase's latent in the step's RNG plan: the paired forward of the D phase used it
            return fake
        return self.sample_g(bs)

    def _generator_forward_for_both_phases(self, bs):
        """The generator runs twice per step with the SAME weights -- ``sample_g`` of the D phase (cnn.py:117, no gradient)
        and of the G phase (:143) -- on different latents.  When the step's random inputs are pre-drawn (``RngFeed`` serving
        a recorded plan: every step but the first) both latents exist up front, and the two forwards run as ONE pass over
        2B latents (``functional.Pair``: BatchNorm statistics per half, running statistics updated D-phase half first).
        -> (fake for the D phase (detached), fake for the G phase (with its graph)) or None."""
        feed = self.rng_feed
        if not (self._in_step and feed.mode == 'serve' and self._g_pairable()):
            return None
        i = feed.cursor
        later = [k for k in range(i + 1, len(feed.plan)) if feed.plan[k][0] == 'z']
        if i >= len(feed.plan) or feed.plan[i] != ('z', bs, self.gan_config.latent_dims) or not later or feed.plan[later[0]] != feed.plan[i]:
            return None
        z_d = self.sample_z(bs)
        z_g = feed.static[later[0]]              # (its plan entry is consumed by the G phase, see _g_forward)
        toggle_grad(self.g, True)                # the G-phase half needs its graph; the D-phase half is detached below
        out = self.g(TF.Pair(z_d, z_g))
        return out.r.detach(), out.f

    def _rng_plan(self, bs):
        """Random inputs of one step in draw order: the D phase's latents, the G phase's latents (trainer.py:153-156)."""
        z = ('z', bs, self.gan_config.latent_dims)
        return [z, z]

    def _known_rng_plan(self, bs):
        """``_rng_plan`` when this is a stock trainer on stock models (the draw order is then this package's own);
        otherwise None: the first step records what is drawn."""
        from .iqn import IQNTrainer
        if type(self) not in (CNNTrainer, IQNTrainer) or type(self.g) is not Generator or type(self.d) is not self.discriminator_class:
            return None
        return self._rng_plan(bs)

    def _g_pairable(self):
        if not getattr(self.args, 'pair_g', True):
            return False
        cached = self.__dict__.get('_pair_g_ok')
        key = (id(self.g), self.data_parallel is not None and self.data_parallel.sync_bn)
        if cached is not None and cached[0] == key:
            return cached[1]
        ok = isinstance(self.g.blocks[0], GeneratorInputMLP)
        for m in self.g.modules():
            if not (type(m).__module__.startswith('tartangan_amd.') or type(m) in (nn.Sequential, nn.Identity, nn.ModuleList)):
                ok = False
            if isinstance(m, BatchNorm2d) and m.sync_group is not None:
                ok = False
        self._pair_g_ok = (key, ok)
        return ok

    def _g_backward(self, fake):
        """Second part (cnn.py:144-148): D(fake) with the stepped discriminator, the loss, g_loss.backward()."""
        with TF.filter_forms(self.d):
            g_loss = self._g_loss(fake, self._labels(len(fake))[:len(fake)])
            with TF.deferred_wgrad():
                g_loss.backward()
        return g_loss.detach()

    def _g_phase(self, bs):
        """Generator half up to and including g_loss.backward() (cnn.py:139-148)."""
        return self._g_backward(self._g_forward(bs))

    def train_batch(self, imgs):
        imgs = imgs.to(self.device)
        self._training_mode()
        dp = self.data_parallel
        graphs_ok = dp is None or dp.capturable      # SyncBN over gloo: host-side collectives inside the passes
        # The step's random inputs (latents, IQN quantile fractions) go through RngFeed: the FIRST step draws them inline,
        # in the reference's order, and records that order; every later step of the same batch shape draws them all up
        # front in the recorded order (same stream: nothing else draws inside a step) -- which is what lets a step replay
        # from HIP graphs and lets the generator's two forwards share one pass.  Another batch shape (a ragged last
        # batch): plain inline draws for that call.
        feed = self.rng_feed
        self._route_rng_through_feed()
        shape, restore = tuple(imgs.shape), None
        if not feed.plan:
            known = self._known_rng_plan(len(imgs))
            if known is not None:
                feed.adopt(known)         # the stock step's draw order is known in advance: no recording step
        if not feed.plan:
            feed.mode, feed.key = 'record', shape
        elif feed.key is None or feed.key == shape:
            feed.key = shape
            feed.mode = 'serve'
            feed.refill()
        else:
            restore, feed.mode = feed.mode, 'off'
        self._in_step = True
        try:
            if graphs_ok and (getattr(self, '_graphs', None) is not None or getattr(self, '_graph_requested', False)):
                vals = self._train_batch_graphed(imgs)
            else:
                vals = self._train_batch_eager(imgs)
        finally:
            self._in_step = False
            self.__dict__.pop('_fake_for_g_phase', None)
            if restore is not None:
                feed.mode = restore
        self.steps += 1
        if not torch.is_tensor(vals):
            vals = torch.stack([v for v in vals if v is not None])
        if feed.mode == 'serve' and restore is None and getattr(self.args, 'prefetch_rng', _PREFETCH_RNG):
            feed.prefetch()          # the next step's latents, drawn under this step's GPU time (see RngFeed.prefetch)
        vals = self._read_back(vals)                                         # one device->host read
        return dict(g_loss=vals[0], d_loss=vals[1], gp=vals[2] if len(vals) > 2 else 0.)

    def _read_back(self, vals):
        """The step's three losses as Python floats (the reference's ``float(g_loss)`` ...: its one host sync per step)."""
        if not vals.is_cuda:
            return vals.tolist()
        host = self.__dict__.get('_loss_host')
        if host is None or host.numel() != vals.numel():
            host = self._loss_host = torch.empty(vals.numel(), dtype=torch.float32).pin_memory()
        host.copy_(vals, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        return host.tolist()

    def _training_mode(self):
        """``self.g.train(); self.d.train()`` (cnn.py:110-111) -- as a check over a cached module list when nothing has to
        change: the recursive ``train()`` walks ~250 modules and costs the replayed step ~150 us of host time in which the
        GPU idles; reading 250 flags costs ~10."""
        cache = self.__dict__.get('_mode_cache')
        if cache is None or cache[0] is not self.g or cache[1] is not self.d:
            cache = (self.g, self.d, list(self.g.modules()) + list(self.d.modules()))
            self._mode_cache = cache
        if not all(m.training for m in cache[2]):
            self.g.train()
            self.d.train()
            self._mode_cache = None          # (a module added since the list was taken is picked up by the next walk)

    def _train_batch_eager(self, imgs):
        d_loss, d_grad_penalty = self._d_phase(imgs)
        self._begin_reduce('d', self.optimizer_d)          # D-bucket all-reduce under the generator forward below
        fake = self._g_forward(len(imgs))
        self._finish_reduce('d')
        self.optimizer_d.step()
        g_loss = self._g_backward(fake)
        self._begin_reduce('g', self.optimizer_g)
        self._finish_reduce('g')
        self.optimizer_g.step()
        self.update_target_generator()
        return g_loss, d_loss, d_grad_penalty

    # ------------------------------------------------------------------ HIP-graph replay of the step
    def enable_graphs(self):
        """Replay the step from captured HIP graphs (D phase | G forward | D Adam + D(fake) + G backward |
        G Adam + EMA; the middle two are one graph on a single GPU) instead of ~1500 eager launches.  The
        cuts are where a data-parallel run starts / joins the all-reduces of its gradient buckets.  Host-side RNG is pre-drawn by ``RngFeed`` in the reference's order, so
        results are identical to eager mode.  The first call runs eagerly (recording the RNG plan),
        the second captures, later calls only replay."""
        if self.device == 'cpu':
            raise RuntimeError('HIP graphs need a ROCm device')
        self._graph_requested = True
        self._graphs = None

    def _route_rng_through_feed(self):
        iqn = getattr(getattr(self.d, 'to_output', None), 'iqn', None)
        if iqn is not None and iqn.tau_source is None:
            iqn.tau_source = lambda rows, q: self._draw('tau', rows, q)

    def _capture(self, imgs):
        """Capture the step's graphs.  The cyclic garbage collector is held off for the duration: a collection that happens to
        run INSIDE a capture and frees something whose destructor talks to the driver (an earlier trainer's CUDAGraph or its memory
        pool -- trainers are cyclic: components, RNG hooks) is an illegal call under stream capture and aborts the process.
        ``torch.cuda.graph`` collects before it starts capturing for the same reason; that does not cover garbage that becomes
        collectable, or a generation threshold that trips, while the capture is running."""
        import gc
        was_enabled = gc.isenabled()
        gc.collect()
        gc.disable()
        try:
            self._capture_graphs(imgs)
        finally:
            if was_enabled:
                gc.enable()

    def _capture_graphs(self, imgs):
        feed = self.rng_feed
        feed.cursor = 0
        self._static_imgs = imgs.clone()
        torch.cuda.synchronize()
        pool = torch.cuda.graph_pool_handle()
        g1, g2a, g2b, g3 = (torch.cuda.CUDAGraph() for _ in range(4))
        # with a process group alive, its watchdog thread polls events; only this thread's calls matter here
        kw = dict(pool=pool)
        dp = self.data_parallel
        multi = dp is not None and dp.multi
        # serial schedule on RCCL: the two bucket all-reduces are captured INTO the graphs (three graphs, as on one GPU);
        # side-stream schedule, or a host-staged backend (gloo): the graphs are cut where the collectives start / join
        inside = multi and dp.buckets_in_graph
        split = multi and not inside
        if multi:
            kw['capture_error_mode'] = 'thread_local'
        with torch.cuda.graph(g1, **kw):
            self._out_d = self._d_phase(self._static_imgs)
            if inside:
                dp.all_reduce_mean(self.optimizer_d.grads)
        if split:           # the generator forward in a graph of its own: it runs while the D bucket is all-reduced
            with torch.cuda.graph(g2a, **kw):
                fake = self._g_forward(len(imgs))
            with torch.cuda.graph(g2b, **kw):
                self.optimizer_d.apply()
                self._out_g = self._g_backward(fake)
        else:
            g2a = None
            with torch.cuda.graph(g2b, **kw):
                self.optimizer_d.apply()
                self._out_g = self._g_phase(len(imgs))
                if inside:
                    dp.all_reduce_mean(self.optimizer_g.grads)
        with torch.cuda.graph(g3, **kw):
            self.optimizer_g.apply()
            self.update_target_generator()
            # the losses side by side in one static buffer: the replayed step reads back ONE tensor, no launch of its own
            self._out_losses = torch.stack([v for v in (self._out_g, self._out_d[0], self._out_d[1]) if v is not None])
        self._buckets_in_graph = inside
        self._graphs = (g1, g2a, g2b, g3)
        feed.cursor = 0

    def _train_batch_graphed(self, imgs):
        feed = self.rng_feed
        if feed.mode != 'serve':
            # call 1 (the RNG plan is being recorded), or a batch of another shape than the plan / the captured graphs
            # (a ragged last batch): this call runs eagerly
            return self._train_batch_eager(imgs)
        gen = (self.optimizer_d.ensure_bound(), self.optimizer_g.ensure_bound())
        if self._graphs is not None and gen != self._graph_gen:
            self._graphs = None                                # parameter buckets were rebuilt: recapture
        dp = self.data_parallel
        if self._graphs is not None and dp is not None and dp.multi and dp.buckets_in_graph != self._buckets_in_graph:
            self._graphs = None                                # the collective schedule changed (autotune_overlap): recapture
        if self._graphs is None:                               # call 2: capture (nothing executes yet)
            try:
                self._capture(imgs)
                self._graph_gen = gen
            except _be.KernelError:
                raise                                          # a tg_* call failed: a real error, never downgraded
            except RuntimeError as exc:                        # e.g. a collective library that refuses capture
                if 'capture' not in str(exc).lower() and 'graph' not in str(exc).lower():
                    raise
                import warnings
                warnings.warn(f'HIP-graph capture failed ({exc!r}); continuing in eager mode')
                self._graph_requested = False
                self._graphs = None
                feed.cursor = 0
                self.__dict__.pop('_fake_for_g_phase', None)
                return self._train_batch_eager(imgs)           # consumes the values refill() drew for this step
        g1, g2a, g2b, g3 = self._graphs
        self._static_imgs.copy_(imgs, non_blocking=True)
        self.optimizer_d.advance(checked=True)
        self.optimizer_g.advance(checked=True)
        if self._buckets_in_graph:                             # (RCCL, serial schedule: the collectives replay with the graphs)
            g1.replay()
            g2b.replay()
            g3.replay()
            return self._out_losses
        g1.replay()
        self._begin_reduce('d', self.optimizer_d)
        if g2a is not None:
            g2a.replay()                                       # generator forward beside the D-bucket all-reduce
        self._finish_reduce('d')
        g2b.replay()
        self._begin_reduce('g', self.optimizer_g)
        self._finish_reduce('g')
        g3.replay()
        return self._out_losses

    def _begin_reduce(self, key, optimizer):
        """Data-parallel hooks: start / join the averaging of a flat gradient bucket over the ranks (no-ops on 1 GPU)."""
        if self.data_parallel is not None:
            self.data_parallel.begin_all_reduce(key, optimizer.grads)

    def _finish_reduce(self, key):
        if self.data_parallel is not None:
            self.data_parallel.finish_all_reduce(key)

    @torch.no_grad()
    def update_target_generator(self, lr=None):
        """target += (g - target) * lr_target_g.  Like the reference (cnn.py:158-165) the
        ``lr`` argument is accepted and ignored: build_models' update_target_generator(1.)
        therefore moves target_g only 1e-3 of the way towards g, it does not copy."""
        ema_update(self.target_g, self.g, self.args.lr_target_g)


def main():
    raise SystemExit('The CLI/epoch loop stays with tartangan; see INTEGRATION.md for the drop-in.')
