"""the device-resident ImageBytesDataset + transform kernel, checkpoint / resume and the sampler.

The expected pixel values are the operations of the reference's transform chain restated with plain torch ops
(torchvision itself is not installed anywhere): ``ToTensor`` = HWC uint8 -> CHW float32 ``.div(255)``, ``Normalize`` =
``.sub_(mean).div_(std)`` (trainers/trainer.py:69-78).  uint8 has 256 values: the GPU test is exhaustive and bit-exact."""
import json
import os

import numpy as np
import pytest
import torch

from emulator import Emulator
from oracle.procedural import procedural_state, synthetic_images


def reference_transform(u8_hwc, y0=0, x0=0, size=None):
    size = size or u8_hwc.shape[0]
    crop = torch.as_tensor(u8_hwc)[y0:y0 + size, x0:x0 + size]
    t = crop.permute(2, 0, 1).contiguous().to(torch.float32).div(255)
    return t.sub_(0.5).div_(0.5)


def make_archive(tmp_path, n=40, size=32, seed=0):
    """a synthetic ImageBytesDataset file in the reference's on-disk format."""
    images = np.random.default_rng(seed).integers(0, 256, (n, size, size, 3), dtype=np.uint8)
    path = os.path.join(tmp_path, 'images.npz')
    np.savez_compressed(path, images=images)
    return path, images


@pytest.fixture
def emulated():
    from tartangan_amd import backend
    prev = backend._set_backend_for_testing(Emulator())
    yield
    backend._set_backend_for_testing(prev)


class _IndexAndCrop(torch.utils.data.Dataset):
    """Stand-in for ImageBytesDataset + RandomCrop under a REAL DataLoader: returns (index, y0, x0) and consumes the default
    generator like torchvision's RandomCrop.get_params (two randint draws per image unless the crop is the whole image)."""

    def __init__(self, n, stored, crop):
        self.n, self.stored, self.crop = n, stored, crop

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        y0 = x0 = 0
        if self.stored != self.crop:
            y0 = int(torch.randint(0, self.stored - self.crop + 1, size=(1,)).item())
            x0 = int(torch.randint(0, self.stored - self.crop + 1, size=(1,)).item())
        return torch.tensor([i, y0, x0])


def _real_dataloader_stream(n, batch, stored, crop, seed, epochs=1, shuffle=True):
    """-> (image order, crop offsets, next default-generator draw) of a real torch DataLoader over ``epochs`` epochs."""
    torch.manual_seed(seed)
    order, crops = [], []
    dl = torch.utils.data.DataLoader(_IndexAndCrop(n, stored, crop), batch_size=batch, shuffle=shuffle, drop_last=True)
    for _ in range(epochs):
        for rows in dl:
            order += rows[:, 0].tolist()
            crops += rows[:, 1:].tolist()
    return order, crops, float(torch.rand(1))


@pytest.mark.parametrize('shuffle', [True, False])
def test_loader_consumes_the_default_generator_like_a_real_dataloader(tmp_path, emulated, shuffle):
    """Two epochs, with random crops: same image order, same crop offsets, same generator state afterwards as
    ``torch.utils.data.DataLoader(shuffle, drop_last=True)`` (whose iterator draws a base seed before the sampler's)."""
    from tartangan_amd.image_bytes_dataset import ImageBytesDataset
    path, images = make_archive(tmp_path, n=22, size=40)
    ds = ImageBytesDataset.from_path(path, crop_size=32, device='cpu')
    order, crops, after = _real_dataloader_stream(22, 8, 40, 32, seed=7, epochs=2, shuffle=shuffle)
    torch.manual_seed(7)
    got = [b for _ in range(2) for b in ds.loader(8, shuffle=shuffle)]
    assert float(torch.rand(1)) == after
    assert len(got) == 4
    want = [reference_transform(images[i], y0, x0, 32) for i, (y0, x0) in zip(order, crops)]
    assert torch.equal(torch.cat(got), torch.stack(want))


def test_loader_order_sharding_and_rng_consumption(tmp_path, emulated):
    from tartangan_amd.image_bytes_dataset import ImageBytesDataset
    path, images = make_archive(tmp_path)
    ds = ImageBytesDataset.from_path(path, device='cpu')
    assert len(ds) == 40 and ds.image_size == 32
    assert torch.equal(ds[3], reference_transform(images[3]))
    # what a real DataLoader(shuffle=True, drop_last=True) (trainers/trainer.py:84-86) draws from the default generator
    order, _, after = _real_dataloader_stream(40, 16, 32, 32, seed=5)
    torch.manual_seed(5)
    batches = list(ds.loader(16))
    assert float(torch.rand(1)) == after                          # same consumption of the default generator
    assert len(batches) == 2                                      # drop_last
    for k, got in enumerate(batches):
        want = torch.stack([reference_transform(images[i]) for i in order[16 * k:16 * k + 16]])
        assert torch.equal(got, want)
    # two ranks: each global batch of 16 split into rows [0, 8) and [8, 16)
    for rank in range(2):
        torch.manual_seed(5)
        shards = list(ds.loader(8, rank=rank, world=2))
        for k, got in enumerate(shards):
            assert torch.equal(got, batches[k][8 * rank:8 * rank + 8])


def test_random_crop_draws_like_torchvision(tmp_path, emulated):
    from tartangan_amd.image_bytes_dataset import ImageBytesDataset
    path, images = make_archive(tmp_path, n=6, size=40)
    ds = ImageBytesDataset.from_path(path, crop_size=32, device='cpu')
    torch.manual_seed(9)
    got = ds.batch([4, 1, 5])
    torch.manual_seed(9)
    for b, i in enumerate([4, 1, 5]):                            # RandomCrop.get_params: i (row) then j (column), per image
        y0 = int(torch.randint(0, 9, size=(1,)).item())
        x0 = int(torch.randint(0, 9, size=(1,)).item())
        assert torch.equal(got[b], reference_transform(images[i], y0, x0, 32))
    with pytest.raises(IndexError):
        ds.batch([6])


@pytest.mark.gpu
def test_image_bytes_kernel_is_bit_exact_for_every_byte_value(tmp_path):
    from tartangan_amd import backend
    from tartangan_amd.image_bytes_dataset import ImageBytesDataset
    backend._set_backend_for_testing(None)
    # every uint8 value in every channel, on image sizes that do / do not divide into 4-pixel groups, with crops
    for (size, crop) in [(32, 32), (128, 128), (40, 32), (21, 18)]:
        n = 5
        images = np.random.default_rng(size).integers(0, 256, (n, size, size, 3), dtype=np.uint8)
        images[0].reshape(-1)[:768] = np.repeat(np.arange(256, dtype=np.uint8), 3)
        ds = ImageBytesDataset(images, crop_size=crop, device='cuda')
        idx = [0, 4, 2, 0, 1]
        torch.manual_seed(3)
        got = ds.batch(idx).cpu()
        torch.manual_seed(3)
        for b, i in enumerate(idx):
            y0 = x0 = 0
            if crop != size:
                y0 = int(torch.randint(0, size - crop + 1, size=(1,)).item())
                x0 = int(torch.randint(0, size - crop + 1, size=(1,)).item())
            assert torch.equal(got[b], reference_transform(images[i], y0, x0, crop)), (size, crop, b)
    vals = torch.arange(256, dtype=torch.uint8).view(1, 16, 16, 1).expand(1, 16, 16, 3).contiguous()
    out = ImageBytesDataset(vals.numpy(), device='cuda').batch([0]).cpu()
    assert torch.equal(out[0], reference_transform(vals[0]))
    assert float(out.min()) == -1.0 and float(out.max()) == 1.0


def _trainer(tmp_path, seed=0):
    from tartangan_amd.models.pluggan import GAN_CONFIGS
    from tartangan_amd.trainers.iqn import IQNTrainer
    cfg = GAN_CONFIGS['32']._replace(attention=(2,))
    args = IQNTrainer.default_args(config=cfg, batch_size=4, device='cpu', output=str(tmp_path), run_id='r',
                                   checkpoint_freq=2, gen_freq=2, resume_training_step=None, resume_training_latest=False)
    tr = IQNTrainer(args)
    torch.manual_seed(seed)
    tr.build_models()
    return tr


def test_checkpoint_resume_round_trip(tmp_path, emulated):
    """components/model_checkpoint.py:32-68: whole-object pickles + trainer.json; a resumed run continues bit-identically."""
    from tartangan_amd.trainers.components import ModelCheckpointComponent
    tr = _trainer(tmp_path)
    ck = ModelCheckpointComponent(tr.args)
    tr.attach(ck)
    ck.on_train_begin(0, {})
    imgs = synthetic_images(4, 32, 1)
    torch.manual_seed(1)
    for _ in range(2):
        tr.train_batch(imgs)
        ck.on_batch_end(tr.steps, {})
    root = f'{tmp_path}/r/checkpoints/2'
    assert sorted(os.listdir(root)) == ['d.pt', 'g.pt', 'g_target.pt', 'opt_d.pt', 'opt_g.pt', 'trainer.json']
    assert json.load(open(f'{root}/trainer.json')) == dict(epoch=1, steps=2)
    torch.manual_seed(2)
    want = tr.train_batch(imgs)                                   # step 3 of the original run

    tr2 = _trainer(tmp_path, seed=123)                            # different init: everything must come from the files
    tr2.args.resume_training_latest = True
    ck2 = ModelCheckpointComponent(tr2.args)
    tr2.attach(ck2)
    ck2.on_train_begin(0, {})
    assert tr2.steps == 2 and ck2.latest_checkpoint_id() == 2
    torch.manual_seed(2)
    got = tr2.train_batch(imgs)
    assert got == want
    for a, b in ((tr.g, tr2.g), (tr.d, tr2.d), (tr.target_g, tr2.target_g)):
        for (k, v), (_, w) in zip(a.state_dict().items(), b.state_dict().items()):
            assert torch.equal(v, w), k
    assert tr2.optimizer_d.step_count == tr.optimizer_d.step_count == 3


def test_stock_module_checkpoint_loads_and_round_trips(tmp_path, emulated):
    """A state_dict with the reference modules' keys (the fixtures hold the reference's own key lists) loads into the HIP
    modules, survives torch.save(module) / torch.load, and comes back out under the same keys with the same values; the
    unpickled module trains on (its parameters are re-homed into fresh flat buckets)."""
    from conftest import load_golden
    fx = load_golden('c32a2_iqn_b8')
    tr = _trainer(tmp_path)
    for net, keys, seed in ((tr.g, fx['state_keys']['g'], 3), (tr.d, fx['state_keys']['d'], 4)):
        assert list(net.state_dict().keys()) == keys
        state = procedural_state(net.state_dict(), seed)
        net.load_state_dict(state)
        path = f'{tmp_path}/m.pt'
        torch.save(net, path)
        back = torch.load(path, weights_only=False)
        assert list(back.state_dict().keys()) == keys
        for k in keys:
            assert torch.equal(back.state_dict()[k], state[k].to(back.state_dict()[k].dtype)), k
    from tartangan_amd.optim import FusedAdam, flatten_parameters
    back_d = torch.load(f'{tmp_path}/m.pt', weights_only=False)
    opt = FusedAdam(back_d, lr=1e-3)
    flat, grads = flatten_parameters(back_d)
    assert all(p.grad is not None for p in back_d.parameters()) and flat is opt.flat
    for p in back_d.parameters():
        p.grad = None                                            # zero_grad(set_to_none=True) of a foreign loop
    opt.zero_grad()
    assert all(p.grad is not None and p.grad.data_ptr() >= opt.grads.data_ptr() for p in back_d.parameters())
    sd = FusedAdam(back_d, lr=5e-4, betas=(0., 0.9)).state_dict()
    opt.load_state_dict(sd)
    assert opt.lr == 5e-4 and opt.betas == (0., 0.9)


def test_sampler_consumes_rng_like_the_reference(tmp_path, emulated):
    from tartangan_amd.trainers.components import ImageSamplerComponent
    from tartangan_amd.trainers.components.image_sampler import image_grid_uint8
    tr = _trainer(tmp_path)
    sm = ImageSamplerComponent(tr.args)
    tr.attach(sm)
    torch.manual_seed(7)
    sm.on_train_begin(0, {})
    tr.g.train(); tr.target_g.train()
    sm.on_batch_end(0, {})
    after = float(torch.rand(1))
    torch.manual_seed(7)
    z32 = torch.randn(32, tr.gan_config.latent_dims)            # image_sampler.py:15
    corners = torch.randn(4, tr.gan_config.latent_dims)         # :47-49, at the first output
    assert float(torch.rand(1)) == after
    assert torch.equal(sm.progress_samples, z32)
    g = sm._latent_grid_samples
    assert g.shape == (25, tr.gan_config.latent_dims) and g.dtype == torch.float32
    assert torch.allclose(g[0], corners[0]) and torch.allclose(g[4], corners[1]) and torch.allclose(g[24], corners[3])
    files = sorted(os.listdir(sm.sample_root))
    assert files == ['grid_sample_0.png', 'sample_0.png']
    from PIL import Image
    assert Image.open(f'{sm.sample_root}/sample_0.png').size == (8 * 34 + 2, 4 * 34 + 2)      # 32 images, 8 per row, padding 2
    x = torch.tensor([-2.0, -1.0, 0.0, 1.0, 3.0]).view(1, 1, 1, 5)
    assert image_grid_uint8(x, padding=0).reshape(-1).tolist() == [0, 0, 128, 255, 255]


def test_reference_adam_checkpoint_loads_into_the_flat_optimiser(tmp_path, emulated):
    """opt_d.pt / opt_g.pt of a reference run are pickled ``torch.optim.Adam`` objects (model_checkpoint.py:39-45);
    ``load_checkpoint`` hands their ``state_dict()`` ({'state', 'param_groups'}) to ``FusedAdam.load_state_dict``: moments
    are packed into the flat buckets, step / lr / betas / eps taken over, and the next step equals torch's."""
    from tartangan_amd.optim import FusedAdam, param_offsets
    tr = _trainer(tmp_path)
    net = tr.d
    stock_params = [torch.nn.Parameter(p.detach().clone()) for p in net.parameters()]
    stock = torch.optim.Adam(stock_params, lr=3e-4, betas=(0., 0.999))
    gen = torch.Generator().manual_seed(3)
    for _ in range(2):
        for p in stock_params:
            p.grad = torch.randn(p.shape, generator=gen)
        stock.step()
    mine = FusedAdam(net, lr=1e-3, betas=(0.5, 0.9))
    mine.load_state_dict(torch.optim.Adam(stock_params).state_dict() | stock.state_dict())
    assert mine.step_count == 2 and mine.lr == 3e-4 and mine.betas == (0., 0.999)
    offs, _ = param_offsets(list(net.parameters()))
    for k, (p, o) in enumerate(zip(stock_params, offs)):
        st = stock.state[p]
        assert torch.equal(mine.exp_avg[o:o + p.numel()].view(p.shape), st['exp_avg']), k
        assert torch.equal(mine.exp_avg_sq[o:o + p.numel()].view(p.shape), st['exp_avg_sq']), k
    # one more step on both from the same weights and gradients
    with torch.no_grad():
        for p, q in zip(net.parameters(), stock_params):
            p.copy_(q)
    for p, q in zip(net.parameters(), stock_params):
        q.grad = torch.randn(q.shape, generator=gen)
        p.grad.copy_(q.grad)
    stock.step()
    mine.step()
    for k, (p, q) in enumerate(zip(net.parameters(), stock_params)):
        assert torch.allclose(p, q, rtol=1e-6, atol=1e-8), k
    # a layout the flat optimiser cannot take leaves it untouched
    bad = stock.state_dict()
    bad['param_groups'][0]['amsgrad'] = True
    with pytest.raises(ValueError):
        mine.load_state_dict(bad)
    assert mine.step_count == 3


def test_draws_outside_a_step_bypass_the_recorded_rng_plan(tmp_path, emulated):
    """ImageSamplerComponent.on_train_begin calls sample_z(32) and the first output sample_z(4) BETWEEN steps; with the
    step's draws routed through RngFeed (graph replay, data parallel) those must not enter the recorded plan, must not hand
    out the step's static buffers and must not be rank-sliced."""
    from tartangan_amd.trainers.components import ImageSamplerComponent
    tr = _trainer(tmp_path)
    sm = ImageSamplerComponent(tr.args)
    tr.attach(sm)
    torch.manual_seed(11)
    sm.on_train_begin(0, {})
    assert tr.rng_feed.plan == []
    imgs = synthetic_images(4, 32, 1)
    logs = [tr.train_batch(imgs)]
    plan = list(tr.rng_feed.plan)
    assert plan and ('z', 32, tr.gan_config.latent_dims) not in plan and len(plan) == 5
    torch.save(tr.d, f'{tmp_path}/d_with_hook.pt')    # the tau hook the trainer installed on the IQN head does not travel
    assert torch.load(f'{tmp_path}/d_with_hook.pt', weights_only=False).to_output.iqn.tau_source is None
    sm.on_batch_end(0, {})                            # sample_z(4) == the batch size: must not alias the step's z buffer
    assert tr.rng_feed.plan == plan
    assert all(sm._latent_grid_samples.data_ptr() != b.data_ptr() for b in tr.rng_feed.static)
    logs.append(tr.train_batch(imgs))                 # (second step: the plan is served, refilled by train_batch itself)
    assert tr.rng_feed.mode == 'serve'
    after = float(torch.rand(1))

    ref = _trainer(tmp_path)                          # the same run with plain inline draws in every step (no plan, no pairing)
    ref.args.pair_g = False
    ref.train_batch = lambda imgs, _tb=ref.train_batch: (ref.rng_feed.plan.clear(), ref.rng_feed.static.clear(), ref.rng_feed.host.clear(), _tb(imgs))[-1]
    sm2 = ImageSamplerComponent(ref.args)
    ref.attach(sm2)
    torch.manual_seed(11)
    sm2.on_train_begin(0, {})
    want = [ref.train_batch(imgs)]
    sm2.on_batch_end(0, {})
    want.append(ref.train_batch(imgs))
    assert float(torch.rand(1)) == after
    for a, b in zip(logs, want):        # (the second step of the first run shares one generator pass between its phases)
        # (g_loss is taken after D's Adam step, whose +-lr moves amplify last-bit differences: bounded loosely here)
        assert all(abs(a[k] - b[k]) <= (1e-3 if k == 'g_loss' else 2e-5) * max(abs(b[k]), 1e-6) for k in b), (a, b)
    assert torch.equal(sm.progress_samples, sm2.progress_samples)


def test_fid_component_hooks_and_flags(tmp_path, emulated):
    """metrics/fid.py:13-46: every fid_freq batches (IS mean, IS std, FID) from trainer.sample_g batches are appended to
    the step's logs; same flags as the reference.  The network is a stand-in with torchvision's attribute names."""
    import argparse
    from oracle.fid_features import procedural_features, tiny_inception
    from tartangan_amd.trainers.components import FIDComponent
    p = argparse.ArgumentParser()
    FIDComponent.add_args_to_parser(p)
    flags = p.parse_args(['--n-inception-imgs', '12', '--fid-freq', '2'])
    assert flags.inception_moments is None and not flags.cleanup_inception_model
    data = procedural_features(60, 16, 3).double().numpy()
    path = os.path.join(tmp_path, 'moments.npz')
    np.savez(path, mu=data.mean(0), sigma=np.cov(data, rowvar=False))
    tr = _trainer(tmp_path)
    for k, v in vars(flags).items():
        setattr(tr.args, k, v)
    tr.args.inception_moments = path
    fid = FIDComponent(tr.args, net=tiny_inception(16, 5, 1))
    tr.attach(fid)
    fid.on_train_begin(0, {})
    logs = {}
    fid.on_batch_end(1, logs)
    assert logs == {}
    torch.manual_seed(4)
    fid.on_batch_end(2, logs)                          # 12 images = 3 sample_g() batches of 4
    assert set(logs) == {'fid', 'inception_score_mean', 'inception_score_std'} and len(logs['fid']) == 1
    assert np.isfinite(logs['fid'][0]) and logs['inception_score_mean'][0] >= 1.0
    torch.manual_seed(4)
    z = torch.randn(4, tr.gan_config.latent_dims)      # the sampling consumed the default generator batch by batch
    torch.randn(4, tr.gan_config.latent_dims); torch.randn(4, tr.gan_config.latent_dims)
    after = float(torch.rand(1))
    torch.manual_seed(4)
    fid.on_batch_end(4, logs)
    assert float(torch.rand(1)) == after and len(logs['fid']) == 2
    fid.on_train_end(4, logs)
    bare = FIDComponent(tr.args)                       # no network handed in: the pretrained weights cannot be fetched here
    tr.attach(bare)
    with pytest.raises(RuntimeError, match='pretrained|Inception'):
        bare.on_train_begin(0, {})
