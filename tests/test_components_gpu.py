"""SURVEY.md 8f-3 on the device: checkpoint / resume and the progress sampler next to a step that replays from HIP graphs
(device-resident flat buckets, pickled CUDA modules, graph re-capture after a resume, draws outside the step).

The loop driven here is the reference's ``Trainer.train`` (trainers/trainer.py:88-107) reduced to its calls:
``on_train_begin`` -> per batch ``train_batch`` + ``on_batch_end(steps, logs)``."""
import json
import os

import pytest
import torch

from oracle.procedural import procedural_state, synthetic_images

pytestmark = pytest.mark.gpu


def _trainer(tmp_path, kind, run_id, init_seed=0, **extra):
    from tartangan_amd import backend
    from tartangan_amd.models.pluggan import GAN_CONFIGS
    from tartangan_amd.trainers.cnn import CNNTrainer
    from tartangan_amd.trainers.iqn import IQNTrainer
    backend._set_backend_for_testing(None)
    cls = {'cnn': CNNTrainer, 'iqn': IQNTrainer}[kind]
    cfg = GAN_CONFIGS['32']._replace(attention=(2,))
    args = cls.default_args(config=cfg, batch_size=8, device='cuda', output=str(tmp_path), run_id=run_id, checkpoint_freq=2,
                            gen_freq=2, resume_training_step=None, resume_training_latest=False, **extra)
    tr = cls(args)
    torch.manual_seed(init_seed)
    tr.build_models()
    return tr


def _load_procedural(tr):
    tr.g.load_state_dict(procedural_state(tr.g.state_dict(), 7))
    tr.target_g.load_state_dict(procedural_state(tr.target_g.state_dict(), 8))
    tr.d.load_state_dict(procedural_state(tr.d.state_dict(), 9))


def _loop(tr, components, batches, seeds):
    """One ``torch.manual_seed`` per step, so that a run that starts later draws the same stream for the same step."""
    logs = []
    for imgs, seed in zip(batches, seeds):
        torch.manual_seed(seed)
        out = tr.train_batch(imgs)
        for c in components:
            c.on_batch_end(tr.steps, out)
        logs.append(out)
    return logs


@pytest.mark.parametrize('kind', ['cnn', 'iqn'])
def test_graph_replay_with_sampler_and_checkpoints_equals_eager(tmp_path, kind):
    """enable_graphs() with ImageSamplerComponent (sample_z(32) at train begin, sample_z(4) at the first output: both drawn
    BETWEEN steps) and ModelCheckpointComponent attached: same losses, same parameters, same RNG consumption and the same
    files as the eager run."""
    from tartangan_amd.trainers.components import ImageSamplerComponent, ModelCheckpointComponent
    batches = [synthetic_images(8, 32, 100 + k).cuda() for k in range(4)]
    results = {}
    for mode in ('eager', 'graphs'):
        tr = _trainer(tmp_path, kind, mode)
        _load_procedural(tr)
        if mode == 'graphs':
            tr.enable_graphs()
        comps = [ImageSamplerComponent(tr.args), ModelCheckpointComponent(tr.args)]
        tr.attach(*comps)
        torch.manual_seed(3)
        for c in comps:
            c.on_train_begin(0, {})
        logs = _loop(tr, comps, batches, seeds=[11, 12, 13, 14])
        results[mode] = (logs, tr.optimizer_g.flat.clone(), tr.optimizer_d.flat.clone(), float(torch.rand(1)),
                         getattr(tr, '_graphs', None) is not None)
        root = f'{tmp_path}/{mode}'
        assert sorted(os.listdir(f'{root}/samples')) == ['grid_sample_2.png', 'grid_sample_4.png', 'sample_2.png', 'sample_4.png']
        assert sorted(os.listdir(f'{root}/checkpoints')) == ['2', '4']
        assert json.load(open(f'{root}/checkpoints/4/trainer.json')) == dict(epoch=1, steps=4)
    eager, graphs = results['eager'], results['graphs']
    assert graphs[4] and not eager[4]
    assert eager[0] == graphs[0]
    assert torch.equal(eager[1], graphs[1]) and torch.equal(eager[2], graphs[2])
    assert eager[3] == graphs[3]


@pytest.mark.parametrize('kind', ['cnn', 'iqn'])
def test_resume_into_a_fresh_graphed_trainer_continues_bit_for_bit(tmp_path, kind):
    """Two graphed steps, checkpoint, then: (a) the same process takes steps 3 and 4; (b) a FRESH trainer (other initial
    weights) resumes from the files with enable_graphs() and takes steps 3 (eager, records), 4 (captures + replays).  Same
    losses, same parameters and optimiser state, bit for bit (model_checkpoint.py:32-68: whole pickled modules / optimisers
    on the device, state_dict round trip into the flat buckets)."""
    from tartangan_amd.trainers.components import ModelCheckpointComponent
    batches = [synthetic_images(8, 32, 200 + k).cuda() for k in range(4)]
    tr = _trainer(tmp_path, kind, 'run')
    _load_procedural(tr)
    tr.enable_graphs()
    ck = ModelCheckpointComponent(tr.args)
    tr.attach(ck)
    ck.on_train_begin(0, {})
    _loop(tr, [ck], batches[:2], seeds=[21, 22])
    assert os.path.isdir(f'{tmp_path}/run/checkpoints/2') and tr._graphs is not None
    want = _loop(tr, [], batches[2:], seeds=[23, 24])

    tr2 = _trainer(tmp_path, kind, 'run', init_seed=99)
    tr2.args.resume_training_latest = True
    tr2.enable_graphs()
    ck2 = ModelCheckpointComponent(tr2.args)
    tr2.attach(ck2)
    ck2.on_train_begin(0, {})
    assert tr2.steps == 2 and tr2.optimizer_d.step_count == 2
    got = _loop(tr2, [], batches[2:], seeds=[23, 24])
    assert tr2._graphs is not None
    assert got == want
    for a, b in ((tr.optimizer_g, tr2.optimizer_g), (tr.optimizer_d, tr2.optimizer_d)):
        assert torch.equal(a.flat, b.flat) and torch.equal(a.exp_avg, b.exp_avg) and torch.equal(a.exp_avg_sq, b.exp_avg_sq)
        assert a.step_count == b.step_count == 4
    assert torch.equal(tr.optimizer_g.flat, tr2.optimizer_g.flat)
    for (k, v), (_, w) in zip(tr.d.state_dict().items(), tr2.d.state_dict().items()):
        assert torch.equal(v, w), k
    for (k, v), (_, w) in zip(tr.target_g.state_dict().items(), tr2.target_g.state_dict().items()):
        assert torch.equal(v, w), k


def test_checkpoint_written_on_the_device_loads_on_the_cpu_emulator(tmp_path):
    """The files are ordinary pickles of modules with the reference's state_dict keys: what a GPU run wrote loads with
    map_location='cpu' into modules of the same architecture."""
    from tartangan_amd.trainers.components import ModelCheckpointComponent
    tr = _trainer(tmp_path, 'cnn', 'dev')
    ck = ModelCheckpointComponent(tr.args)
    tr.attach(ck)
    tr.train_batch(synthetic_images(8, 32, 1).cuda())
    ck.save_checkpoint(tr.steps)
    g = torch.load(f'{tmp_path}/dev/checkpoints/1/g.pt', map_location='cpu', weights_only=False)
    assert all(v.device.type == 'cpu' for v in g.state_dict().values())
    for (k, v), (k2, w) in zip(g.state_dict().items(), tr.g.state_dict().items()):
        assert k == k2 and torch.equal(v, w.cpu()), k


def test_garbage_collection_during_graph_capture_is_harmless(tmp_path):
    """An earlier graphed trainer that is only cyclic garbage by the time the next trainer captures its step (components and RNG
    hooks make trainers cyclic): with the collector's thresholds at 1 a collection would run inside the capture and free the old
    trainer's graphs there -- an illegal driver call under stream capture, the process aborts.  ``_capture`` collects first and
    holds the collector off while it captures."""
    import gc
    batches = [synthetic_images(8, 32, 300 + k).cuda() for k in range(3)]
    old = gc.get_threshold()
    try:
        for run in range(2):
            tr = _trainer(tmp_path, 'iqn', f'gc{run}')
            _load_procedural(tr)
            tr.enable_graphs()
            tr._self_cycle = tr                      # (whatever else: this trainer can only be freed by the cyclic collector)
            gc.set_threshold(1, 1, 1)
            logs = _loop(tr, [], batches, [5, 6, 7])
            gc.set_threshold(*old)
            assert tr._graphs is not None and all(torch.isfinite(torch.as_tensor(list(map(float, l.values())))).all() for l in logs)
            del tr
    finally:
        gc.set_threshold(*old)
        gc.collect()
