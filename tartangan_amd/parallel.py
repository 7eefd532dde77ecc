"""Data parallelism for the G+D step: one process per GPU, RCCL over xGMI.

The path shards along the batch: every loss is a batch mean, so the global
gradient is the mean of the shard gradients.  Each network's gradients live in ONE flat fp32
bucket (tartangan_amd.optim), so a step needs exactly two collectives -- all-reduce of the D
bucket (2.4 MB at 128:3) after the D backward, of the G bucket (5.1 MB) after the G backward --
issued between the captured HIP graphs of the step.  At these sizes the collectives are
latency-bound; there is nothing to bucket further.

BatchNorm uses per-shard batch statistics (not synchronised): an N-GPU run equals the
reference at the local batch size with averaged gradients, NOT the reference at the global
batch.  Latents and IQN quantile fractions are drawn as GLOBAL tensors from the same CPU seed
on every rank and sliced per rank (trainers.trainer.RngFeed), images are sharded by the caller.
"""
import torch
import torch.distributed as dist

from . import backend as _be
from .optim import flatten_parameters


class DataParallel:
    def __init__(self, trainer, process_group=None):
        if not dist.is_initialized():
            raise RuntimeError('init torch.distributed first (backend "nccl" = RCCL on ROCm)')
        self.group = process_group
        self.rank = dist.get_rank(process_group)
        self.world = dist.get_world_size(process_group)
        self.trainer = trainer
        trainer.data_parallel = self
        feed = trainer.rng_feed
        feed.rank, feed.world = self.rank, self.world
        trainer._route_rng_through_feed()       # IQN taus must come through the feed to be sliced per rank
        if not getattr(trainer, '_graph_requested', False):
            feed.mode = 'off'                   # eager: draw (and slice) the global tensors on every call
        self.sync_state()

    def sync_state(self):
        """Make every rank start from rank 0's parameters and buffers."""
        for module in (self.trainer.g, self.trainer.target_g, self.trainer.d):
            flat, _ = flatten_parameters(module)
            dist.broadcast(flat, src=0, group=self.group)
            for buf in module.buffers():
                dist.broadcast(buf, src=0, group=self.group)

    def all_reduce_mean(self, flat_grads):
        if self.world == 1:
            return
        dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM, group=self.group)
        _be.get().scale(flat_grads, 1.0 / self.world, flat_grads, flat_grads.numel())

    def shard(self, global_batch):
        """Rows of a globally-seeded batch that belong to this rank."""
        n = global_batch.shape[0] // self.world
        return global_batch[self.rank * n:(self.rank + 1) * n]
