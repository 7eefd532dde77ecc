"""Input side of the FID / Inception-score pipeline against ``fid_frontend.json`` -- written by
tests/golden/make_fid_golden.py from the REFERENCE's own ``accumulate_inception_activations`` and ``WrapInception``
(inception_utils.py:34-95, 249-268) around a stand-in network with torchvision's attribute names (the pretrained
Inception-v3 is unreachable offline): sampling loop, both normalisations, 299 x 299 align_corners resize, softmax, and
the metrics tail.  CPU: host logic over the emulator; ``-m gpu``: ``tg_inception_preprocess`` and the rest on the HIP kernels."""
import json
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN_DIR
from emulator import Emulator
from oracle.fid_features import blocky_images, procedural_features, tiny_inception
from oracle.procedural import summarize

with open(os.path.join(GOLDEN_DIR, 'fid_frontend.json')) as f:
    FX = json.load(f)


def _check_summary(t, ref, rel, what):
    got = summarize(t, len(ref['idx']))
    assert got['numel'] == ref['numel'], what
    assert abs(got['l2'] - ref['l2']) <= rel * ref['l2'], (what, got['l2'], ref['l2'])
    for g, r in zip(got['samples'], ref['samples']):
        assert abs(g - r) <= 4 * rel * ref['max_abs'] + 1e-9, (what, g, r)


def _moments_file(tmp_path):
    d = FX['data_features']
    data = procedural_features(d['n'], FX['D'], d['seed'], shift=d['shift']).double().numpy()
    path = os.path.join(tmp_path, 'moments.npz')
    np.savez(path, mu=data.mean(0), sigma=np.cov(data, rowvar=False))
    return path


def _run(device, tmp_path, rel):
    from tartangan_amd import inception_utils as IU
    inner = tiny_inception(FX['D'], FX['classes'], 5).to(device)
    net = IU.WrapInception(inner).to(device)
    seen = []
    inner.Conv2d_1a_3x3.register_forward_pre_hook(lambda m, inp: seen.append(inp[0].detach().clone()))
    calls = [0]

    def sample():
        calls[0] += 1
        return blocky_images(FX['batch'], FX['size'], 700 + calls[0]).to(device)
    pool, probs = IU.accumulate_inception_activations(sample, net, FX['want'])
    assert calls[0] == FX['calls'] and pool.shape == (FX['n'], FX['D']) and probs.shape == (FX['n'], FX['classes'])
    _check_summary(seen[0], FX['preprocessed_first_batch'], rel, 'normalise x2 + resize')
    _check_summary(pool, FX['pool'], 10 * rel, 'pool')
    _check_summary(probs, FX['probs'], 10 * rel, 'softmax(logits)')
    # WrapInception alone on a non-square source
    seen.clear()
    net(blocky_images(3, 40, 901)[:, :, :, :27].contiguous().to(device))
    assert seen[0].shape == (3, 3, 299, 299)
    _check_summary(seen[0], FX['wrap_only_40x27'], rel, 'normalise + resize 40x27 -> 299x299')
    # a network without forward_samples gets the transform alone, at the sample's own size
    got = []
    IU.accumulate_inception_activations(sample, lambda x: (got.append(x) or (x.mean((2, 3)), x.mean((2, 3)))), 8)
    want = (blocky_images(FX['batch'], FX['size'], 700 + calls[0]) + 1) / 2
    want = (want - IU.VGG_MEAN) / IU.VGG_STD
    assert torch.allclose(got[0].cpu(), want, rtol=1e-5, atol=1e-6)
    # prepare_inception_metrics -> get_inception_metrics: the whole pipeline
    calls[0] = 0
    metrics = IU.prepare_inception_metrics(_moments_file(tmp_path), device, False, net=net)
    is_mean, is_std, fid = metrics(sample, FX['want'], num_splits=5, prints=False)
    assert abs(is_mean - FX['is_mean']) <= 1e-4 * FX['is_mean'] and abs(is_std - FX['is_std']) <= 1e-3 * FX['is_std'] + 1e-6
    assert abs(fid - FX['fid']) <= 3 * rel * FX['fid'], (fid, FX['fid'])


def test_frontend_host_logic_on_the_emulator(tmp_path):
    from tartangan_amd import backend
    prev = backend._set_backend_for_testing(Emulator())
    try:
        _run('cpu', str(tmp_path), 1e-5)
    finally:
        backend._set_backend_for_testing(prev)


@pytest.mark.gpu
def test_frontend_on_the_hip_kernels(tmp_path):
    from tartangan_amd import backend
    backend._set_backend_for_testing(None)
    _run('cuda', str(tmp_path), 1e-4)


@pytest.mark.gpu
def test_inception_preprocess_kernel_shapes():
    """tg_inception_preprocess against the emulator: up- and down-sampling, non-square, 0 / 1 / 2 normalisation stages."""
    from tartangan_amd import backend
    backend._set_backend_for_testing(None)
    K, E = backend.get(), Emulator()
    mean, std = torch.tensor([0.485, 0.456, 0.406]), torch.tensor([0.229, 0.224, 0.225])
    for (B, H, W, OH, OW) in [(4, 128, 128, 299, 299), (2, 32, 32, 299, 299), (2, 40, 27, 299, 299), (3, 299, 299, 299, 299),
                              (2, 512, 384, 299, 299), (1, 5, 7, 1, 1), (2, 64, 64, 64, 64)]:
        for stages in (0, 1, 2):
            x = blocky_images(B, max(H, W), 31)[:, :, :H, :W].contiguous()
            want = torch.empty(B, 3, OH, OW)
            E.inception_preprocess(x, mean, std, want, B, 3, H, W, OH, OW, stages)
            got = torch.empty(B, 3, OH, OW, device='cuda')
            K.inception_preprocess(x.cuda(), mean.cuda(), std.cuda(), got, B, 3, H, W, OH, OW, stages)
            assert torch.allclose(got.cpu(), want, rtol=2e-5, atol=2e-5), (B, H, W, OH, OW, stages, float((got.cpu() - want).abs().max()))


# ---------------------------------------------------------------------------------------------- rank-sharded moments
def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _shard_worker(rank, world, port, path, out):
    from tartangan_amd import backend, inception_utils as IU
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    backend._set_backend_for_testing(Emulator())
    net = IU.WrapInception(tiny_inception(FX['D'], FX['classes'], 5))
    k = [0]

    def sample():                       # rank r draws batches r, r + world, ...
        k[0] += 1
        return blocky_images(FX['batch'], FX['size'], 700 + rank + 1 + (k[0] - 1) * world)
    metrics = IU.prepare_inception_metrics(path, 'cpu', False, net=net)
    res = metrics(sample, 48, num_splits=2, prints=False)
    if rank == 0:
        out.put(res)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_moments_two_ranks_equal_one_process(tmp_path):
    """Each of two ranks samples its share (24 of 48 images) and the moments are reduced: same FID as one process over the
    union; the Inception score equals the single-process score on the gathered order (rank 0's samples, then rank 1's)."""
    from tartangan_amd import backend, inception_utils as IU
    path = _moments_file(str(tmp_path))
    ctx = mp.get_context('spawn')
    out, port = ctx.Queue(), _free_port()
    procs = [ctx.Process(target=_shard_worker, args=(r, 2, port, path, out)) for r in range(2)]
    for p in procs:
        p.start()
    is_mean, is_std, fid = out.get(timeout=300)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    prev = backend._set_backend_for_testing(Emulator())
    try:
        net = IU.WrapInception(tiny_inception(FX['D'], FX['classes'], 5))
        order = [1, 3, 5, 2, 4, 6]                       # rank 0's three batches, then rank 1's
        k = [0]

        def sample():
            k[0] += 1
            return blocky_images(FX['batch'], FX['size'], 700 + order[k[0] - 1])
        want = IU.prepare_inception_metrics(path, 'cpu', False, net=net)(sample, 48, num_splits=2, prints=False)
    finally:
        backend._set_backend_for_testing(prev)
    assert abs(fid - want[2]) <= 1e-4 * want[2], (fid, want[2])
    assert abs(is_mean - want[0]) <= 1e-5 * want[0] and abs(is_std - want[1]) <= 1e-4 * want[1] + 1e-7
