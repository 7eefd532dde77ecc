"""Data-parallel path on CPU: world_size 2 over gloo, kernels emulated (tests/emulator.py).

A 2-rank run on half batches with averaged gradients must equal the single-process run on the
full batch: same losses (mean over ranks), same parameters after the step, same RNG stream (global
z / tau drawn identically on every rank and sliced, tau rows quantile-major).  That holds with
norm='id' (no BatchNorm, the reference's --norm id) and, with BatchNorm, when the statistics are
synchronised (DataParallel(sync_bn=True): forward, backward and R1 second-backward sums all-reduced
per layer) -- the property BASELINE.json's configs 4 / 5 (batch 256 / 512 over 8 GPUs) rest on."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from emulator import Emulator
from oracle.procedural import procedural_state, synthetic_images


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _build(kind, batch, seed=0, norm='id'):
    from tartangan_amd import backend
    from tartangan_amd.models.pluggan import GAN_CONFIGS
    from tartangan_amd.trainers.cnn import CNNTrainer
    from tartangan_amd.trainers.iqn import IQNTrainer
    backend._set_backend_for_testing(Emulator())
    cls = {'cnn': CNNTrainer, 'iqn': IQNTrainer}[kind]
    cfg = GAN_CONFIGS['32']._replace(attention=(2,))
    tr = cls(cls.default_args(config=cfg, batch_size=batch, device='cpu', norm=norm))
    torch.manual_seed(seed)
    tr.build_models()
    tr.g.load_state_dict(procedural_state(tr.g.state_dict(), 7))
    tr.target_g.load_state_dict(procedural_state(tr.target_g.state_dict(), 8))
    tr.d.load_state_dict(procedural_state(tr.d.state_dict(), 9))
    return tr


def _worker(rank, world, port, kind, global_batch, out, norm='id'):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from tartangan_amd.parallel import DataParallel
    tr = _build(kind, global_batch // world, seed=rank, norm=norm)   # different init per rank: sync_state must fix it
    if rank != 0:
        with torch.no_grad():
            for p in tr.d.parameters():
                p.add_(0.5)
    dp = DataParallel(tr, sync_bn=(norm == 'bn'))
    imgs = dp.shard(synthetic_images(global_batch, 32, 4321))
    torch.manual_seed(1234)
    logs = [tr.train_batch(imgs), tr.train_batch(imgs)]
    vals = torch.tensor([[l['g_loss'], l['d_loss'], l['gp']] for l in logs], dtype=torch.float64)
    dist.all_reduce(vals)
    vals /= world
    flat_d = tr.optimizer_d.flat.clone()
    gathered = [torch.zeros_like(flat_d) for _ in range(world)]
    dist.all_gather(gathered, flat_d)
    if rank == 0:
        bufs = {k: v.tolist() for k, v in tr.d.state_dict().items() if 'running' in k}
        out.put(dict(losses=vals.tolist(), d=flat_d.tolist(), g=tr.optimizer_g.flat.tolist(), d_buffers=bufs,
                     replicas_equal=all(torch.equal(gathered[0], t) for t in gathered),
                     rng_after=float(torch.rand(1))))
    dist.destroy_process_group()


@pytest.mark.parametrize('kind,norm', [('cnn', 'id'), ('iqn', 'id'), ('cnn', 'bn'), ('iqn', 'bn')])
def test_two_rank_dp_equals_single_process_full_batch(kind, norm):
    world, global_batch = 2, 8
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, kind, global_batch, out, norm)) for r in range(world)]
    for p in procs:
        p.start()
    res = out.get(timeout=300)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res['replicas_equal']

    torch.set_num_threads(1)
    single = _build(kind, global_batch, norm=norm)
    imgs = synthetic_images(global_batch, 32, 4321)
    torch.manual_seed(1234)
    logs = [single.train_batch(imgs), single.train_batch(imgs)]
    rng_after = float(torch.rand(1))
    from tartangan_amd import backend
    backend._set_backend_for_testing(None)
    for step in range(2):
        for got, name in zip(res['losses'][step], ('g_loss', 'd_loss', 'gp')):
            want = logs[step][name]
            tol = 2e-5 if step == 0 else 2e-3
            assert abs(got - want) <= tol * max(abs(want), 1e-6), (step, name, got, want)
    d_single = single.optimizer_d.flat
    # Adam(beta1 = 0) turns rounding-level gradients into +-lr moves: parameters whose true gradient is zero (every conv
    # bias in front of a BatchNorm) may end up to ~5 lr apart after two steps; everything else agrees to rounding
    diff = (torch.tensor(res['d']) - d_single).abs()
    assert float(diff.max()) <= 5 * 4e-4, float(diff.max())
    frac_close = diff.lt(1e-6).float().mean()
    assert frac_close > (0.98 if norm == 'id' else 0.95), float(frac_close)
    assert res['rng_after'] == rng_after                        # identical CPU RNG consumption
    for k, v in res['d_buffers'].items():                       # SyncBN: running statistics of the global batch on every rank
        # (two steps = three discriminator passes after the first optimiser step: weights that moved by Adam's +-lr
        # rounding noise, see above, shift the later layers' statistics by ~1e-3; per-shard statistics of 4 instead of
        # 8 images would be off by ~1e-1)
        assert torch.allclose(torch.tensor(v), single.d.state_dict()[k], rtol=1e-3, atol=3e-3), k
