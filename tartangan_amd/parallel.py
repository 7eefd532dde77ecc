"""Data parallelism for the G+D step: one process per GPU, RCCL over xGMI.

The path shards along the batch: every loss is a batch mean, so the global
gradient is the mean of the shard gradients.  Each network's gradients live in ONE flat fp32
bucket (tartangan_amd.optim), so a step needs exactly two gradient collectives -- all-reduce of
the D bucket (2.4 MB at 128:3) after the D backward, of the G bucket (5.1 MB) after the G backward.
At these sizes the collectives are latency-bound; there is nothing to bucket further.

Schedule.  Default (``overlap=False``): the SERIAL schedule -- both all-reduces on the compute stream right after the
backward that fills the bucket; with RCCL they are device work and are captured INTO the step's graphs (three graphs, as on
one GPU: ``buckets_in_graph``), with a host-staged backend (gloo) they stay eager calls between four graph replays.
``overlap=True``: both all-reduces on a side stream that waits for the producing backward, joined only where consumed (the D
bucket after the G phase's generator forward, trainers.cnn._g_forward; the G bucket immediately).  Measured with RCCL on one
rank (``rehearse=True``, bench.py --rehearse-rccl, DESIGN.md "Multi-GPU"): 9.47 ms/step serial in-graph, 9.92 side stream, 9.43
without collectives -- since the generator's two forwards share one pass (``functional.Pair``) there is nothing left to hide the D
bucket under, and un-pairing D(real) | D(fake) to hide ~50-100 us of the G bucket would cost ~0.4 ms.  All schedules give
bit-identical parameters (tests/test_dp_gloo.py, tests/test_dp_gpu.py).

BatchNorm: with ``sync_bn=True`` every BatchNorm2d of G and D normalises with the statistics of
the GLOBAL batch (functional.SyncGroup: per-layer all-reduces of the forward / backward /
second-backward sums), so an N-rank run on B/N images per rank IS the reference at batch B -- what
BASELINE.json's configs 4 and 5 (batch 256 / 512 over 8 GPUs) need to match the CPU reference.
With ``sync_bn=False`` statistics are per shard: the run equals the reference at the local batch
size with averaged gradients (what torch DDP does without SyncBatchNorm), at ~120 fewer small
collectives per step.  Latents and IQN quantile fractions are drawn as GLOBAL tensors from the
same CPU seed on every rank and sliced per rank (trainers.trainer.RngFeed); images are sharded by
the caller.
"""
import torch
import torch.distributed as dist

from . import backend as _be
from . import functional as TF
from .models.layers import BatchNorm2d
from .optim import flatten_parameters


class DataParallel:
    def __init__(self, trainer, process_group=None, sync_bn=False, overlap=None, rehearse=False, graph_buckets=True):
        if not dist.is_initialized():
            raise RuntimeError('init torch.distributed first (backend "nccl" = RCCL on ROCm)')
        self.group = process_group
        self.rank = dist.get_rank(process_group)
        self.world = dist.get_world_size(process_group)
        # ``rehearse``: a ONE-rank group behaves like an N-rank one -- every collective of the step is issued (bucket all-reduces
        # on the side stream, SyncBN's sums inside the captured passes, the split graphs) and changes nothing.  It is how the
        # RCCL leg is exercised on a one-GPU box (RCCL refuses two ranks on one device): tests/test_dp_gpu.py.
        self.multi = self.world > 1 or bool(rehearse)
        self.backend = dist.get_backend(process_group)
        self.trainer = trainer
        self.sync_bn = bool(sync_bn)
        self.graph_buckets = bool(graph_buckets)     # False: bucket all-reduces stay eager calls between graph replays
        on_device = str(trainer.device) != 'cpu'
        # default: the SERIAL schedule.  Measured with RCCL on one rank (bench.py --rehearse-rccl, DESIGN.md "Multi-GPU"): the
        # side-stream schedule costs the step 0.4 ms (four graphs, two stream joins) and has nothing left to hide the D bucket
        # under since the generator's two forwards share one pass; gloo's device path (host staging with blocking
        # synchronisation) even stalls for seconds when a side-stream collective meets a replaying graph.
        self.overlap = False if overlap is None else bool(overlap)
        self._side = torch.cuda.Stream() if on_device else None
        self._pending = {}
        trainer.data_parallel = self
        feed = trainer.rng_feed
        feed.rank, feed.world = self.rank, self.world
        trainer._route_rng_through_feed()       # IQN taus must come through the feed to be sliced per rank
        self.bn_group = None
        if self.sync_bn and self.multi:
            # SyncBN's collectives sit INSIDE the passes (with RCCL: captured into the step's graphs) while a gradient bucket
            # may be in flight on the side stream.  Two collectives of ONE communicator issued from independent streams can
            # reach the device in a different order on different ranks, which RCCL/NCCL does not allow (it may hang) -- so
            # the BatchNorm sums get a communicator of their own; the bucket all-reduces keep ``process_group``.
            ranks = dist.get_process_group_ranks(process_group) if process_group is not None else None
            self.bn_group = dist.new_group(ranks=ranks) if self.overlap else process_group
            handle = TF.SyncGroup(self.bn_group, self.world, rehearse=self.multi)
            for net in (trainer.g, trainer.d):
                for m in net.modules():
                    if isinstance(m, BatchNorm2d):
                        m.sync_group = handle
        self.sync_state()

    @property
    def collective_name(self):
        return {'nccl': 'RCCL', 'gloo': 'gloo'}.get(self.backend, self.backend)

    @property
    def capturable(self):
        """May a step with in-graph collectives (SyncBN) be captured into HIP graphs?  Only RCCL enqueues device work."""
        return not (self.sync_bn and self.multi) or self.backend == 'nccl'

    @property
    def buckets_in_graph(self):
        """Serial schedule on RCCL: the gradient-bucket all-reduces are device work on the compute stream and are captured
        into the step's graphs like every other launch (no eager call between graph replays)."""
        return self.multi and not self.overlap and self.backend == 'nccl' and self.graph_buckets

    def sync_state(self):
        """Make every rank start from rank 0's parameters and buffers."""
        for module in (self.trainer.g, self.trainer.target_g, self.trainer.d):
            flat, _ = flatten_parameters(module)
            dist.broadcast(flat, src=0, group=self.group)
            for buf in module.buffers():
                dist.broadcast(buf, src=0, group=self.group)

    # ------------------------------------------------------------------ gradient averaging
    def _reduce_now(self, flat_grads):
        dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM, group=self.group)
        _be.get().scale(flat_grads, 1.0 / self.world, flat_grads, flat_grads.numel())

    def all_reduce_mean(self, flat_grads):
        """Blocking form (serial schedule)."""
        if self.multi:
            self._reduce_now(flat_grads)

    def begin_all_reduce(self, key, flat_grads):
        """Start averaging ``flat_grads`` over the ranks; ``finish_all_reduce(key)`` must run before anything reads
        them.  With a side stream the collective runs beside whatever the compute stream does in between."""
        if not self.multi:
            return
        if self._side is None or not self.overlap:
            self._reduce_now(flat_grads)
            return
        self._side.wait_stream(torch.cuda.current_stream())      # after the backward that produced the bucket
        with torch.cuda.stream(self._side):
            self._reduce_now(flat_grads)
        self._pending[key] = flat_grads

    def finish_all_reduce(self, key):
        if self._pending.pop(key, None) is not None:
            torch.cuda.current_stream().wait_stream(self._side)

    def autotune_overlap(self, step, steps=2):
        """Time ``steps`` calls of ``step()`` with the side-stream schedule and with the serial one and keep the faster
        (same decision on every rank: the timings are max-reduced).  Both schedules give identical numbers, so this is
        purely a speed choice -- a guard against a collective library whose side-stream path misbehaves next to graph replay."""
        import time
        if not self.multi or self._side is None:
            return self.overlap
        took = []
        for mode in (True, False):
            self.overlap = mode
            step()                                   # settle
            torch.cuda.synchronize()
            dist.barrier(group=self.group)
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            torch.cuda.synchronize()
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=self.trainer.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            took.append(float(t))
        self.overlap = took[0] <= took[1] * 1.02
        return self.overlap

    def shard(self, global_batch):
        """Rows of a globally-seeded batch that belong to this rank."""
        n = global_batch.shape[0] // self.world
        return global_batch[self.rank * n:(self.rank + 1) * n]
