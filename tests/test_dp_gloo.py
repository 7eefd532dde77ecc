"""Data-parallel path on CPU: world_size 2 over gloo, kernels emulated (tests/emulator.py).

With norm='id' (no BatchNorm, the reference's --norm id) a 2-rank run on half batches with
averaged gradients must equal the single-process run on the full batch: same losses (mean over
ranks), same parameters after the step, same RNG stream (global z / tau drawn identically on
every rank and sliced, tau rows quantile-major)."""
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


def _build(kind, batch, seed=0):
    from tartangan_amd import backend
    from tartangan_amd.models.pluggan import GAN_CONFIGS
    from tartangan_amd.trainers.cnn import CNNTrainer
    from tartangan_amd.trainers.iqn import IQNTrainer
    backend._set_backend_for_testing(Emulator())
    cls = {'cnn': CNNTrainer, 'iqn': IQNTrainer}[kind]
    cfg = GAN_CONFIGS['32']._replace(attention=(2,))
    tr = cls(cls.default_args(config=cfg, batch_size=batch, device='cpu', norm='id'))
    torch.manual_seed(seed)
    tr.build_models()
    tr.g.load_state_dict(procedural_state(tr.g.state_dict(), 7))
    tr.target_g.load_state_dict(procedural_state(tr.target_g.state_dict(), 8))
    tr.d.load_state_dict(procedural_state(tr.d.state_dict(), 9))
    return tr


def _worker(rank, world, port, kind, global_batch, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from tartangan_amd.parallel import DataParallel
    tr = _build(kind, global_batch // world, seed=rank)          # different init per rank: sync_state must fix it
    if rank != 0:
        with torch.no_grad():
            for p in tr.d.parameters():
                p.add_(0.5)
    dp = DataParallel(tr)
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
        out.put(dict(losses=vals.tolist(), d=flat_d.tolist(), g=tr.optimizer_g.flat.tolist(),
                     replicas_equal=all(torch.equal(gathered[0], t) for t in gathered),
                     rng_after=float(torch.rand(1))))
    dist.destroy_process_group()


@pytest.mark.parametrize('kind', ['cnn', 'iqn'])
def test_two_rank_dp_equals_single_process_full_batch(kind):
    world, global_batch = 2, 8
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, kind, global_batch, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = out.get(timeout=300)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res['replicas_equal']

    torch.set_num_threads(1)
    single = _build(kind, global_batch)
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
    assert torch.allclose(torch.tensor(res['d']), d_single, rtol=0, atol=2.5 * 4e-4)     # Adam sign noise bound
    frac_close = (torch.tensor(res['d']) - d_single).abs().lt(1e-6).float().mean()
    assert frac_close > 0.98
    assert res['rng_after'] == rng_after                        # identical CPU RNG consumption
