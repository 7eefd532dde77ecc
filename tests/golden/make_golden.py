#!/usr/bin/env python3
"""Generate golden fixtures by running the REFERENCE implementation.

Runs only in the build container (needs /root/reference); the fixtures it
writes (plain numbers, tests/golden/*.json) are what travels to the GPU box.

What is executed is the reference's own code, unmodified:
  * ``tartangan.trainers.cnn.CNNTrainer`` / ``tartangan.trainers.iqn.IQNTrainer``
    ``build_models()`` and ``train_batch()`` (reference trainers/cnn.py:29-156,
    trainers/iqn.py:29-147)
  * ``tartangan.models.pluggan`` Generator / Discriminator / IQNDiscriminator.

The trainer modules import three third-party packages that are absent from
this image and are never called by build_models/train_batch (torchvision,
smart_open, boto3).  They are registered as empty modules so
the import statements succeed; no reference code is altered or copied.

Usage:  python tests/golden/make_golden.py [case ...]
"""
import argparse
import copy
import json
import os
import sys
import time
import types

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REFERENCE = os.environ.get('TARTANGAN_REFERENCE', '/root/reference')
sys.path.insert(0, REPO)
sys.path.insert(0, REFERENCE)

import torch  # noqa: E402

from oracle.procedural import procedural_state, summarize, synthetic_images  # noqa: E402


class _Inert(types.ModuleType):
    """Empty stand-in for an absent third-party package: any attribute is an
    inert callable, any submodule import yields another stand-in."""
    __path__ = []

    def __getattr__(self, item):
        if item.startswith('__'):
            raise AttributeError(item)
        if item[0].isupper() and not item.isupper():
            return type(item, (), {})        # usable as a base class
        return lambda *a, **k: None


class _InertFinder:
    ROOTS = ('torchvision', 'smart_open', 'boto3')

    def find_spec(self, name, path=None, target=None):
        import importlib.machinery
        if name.split('.')[0] in self.ROOTS:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        return _Inert(spec.name)

    def exec_module(self, module):
        pass


def _install_absent_third_party():
    sys.meta_path.append(_InertFinder())
    import tqdm._utils
    if not hasattr(tqdm._utils, '_unicode'):
        tqdm._utils._unicode = str


_install_absent_third_party()

from tartangan.models.pluggan import GAN_CONFIGS  # noqa: E402
from tartangan.trainers.cnn import CNNTrainer  # noqa: E402
from tartangan.trainers.iqn import IQNTrainer  # noqa: E402

# The author's commented-out attention placements (pluggan.py:227,240), via
# the public namedtuple API.
GAN_CONFIGS['64:1'] = GAN_CONFIGS['64']._replace(attention=(1,))
GAN_CONFIGS['128:3'] = GAN_CONFIGS['128']._replace(attention=(3,))
GAN_CONFIGS['32:2'] = GAN_CONFIGS['32']._replace(attention=(2,))
GAN_CONFIGS['256:3'] = GAN_CONFIGS['256']._replace(attention=(3,))

CASES = {
    # name: (config, trainer, batch, steps)
    'c32_cnn_b16': ('32', 'cnn', 16, 3),
    'c32_iqn_b16': ('32', 'iqn', 16, 3),
    'c32a2_cnn_b8': ('32:2', 'cnn', 8, 3),
    'c32a2_iqn_b8': ('32:2', 'iqn', 8, 3),
    'c64_cnn_b8': ('64', 'cnn', 8, 2),
    'c64a1_cnn_b8': ('64:1', 'cnn', 8, 3),
    'c64a1_iqn_b8': ('64:1', 'iqn', 8, 3),
    'c128a3_cnn_b4': ('128:3', 'cnn', 4, 2),
    'c128a3_iqn_b4': ('128:3', 'iqn', 4, 2),
    'c64a1_cnn_b64': ('64:1', 'cnn', 64, 1),
    'c64a1_iqn_b64': ('64:1', 'iqn', 64, 1),
    # the benched workload and the per-rank batches of BASELINE.json configs 4 / 5
    'c128a3_cnn_b64': ('128:3', 'cnn', 64, 1),
    'c128a3_cnn_b32': ('128:3', 'cnn', 32, 1),
    'c128a3_iqn_b64': ('128:3', 'iqn', 64, 1),
    # option variants: trainer flags as a 5th element
    'c32_cnn_b8_selu': ('32', 'cnn', 8, 2, dict(activation='selu')),
    'c32_cnn_b8_elu': ('32', 'cnn', 8, 2, dict(activation='elu')),
    'c32_iqn_b8_selu': ('32', 'iqn', 8, 2, dict(activation='selu')),
    'c32_cnn_b8_tiledz': ('32', 'cnn', 8, 2, dict(g_base='tiledz')),
    'c64a1_cnn_b4_scale075': ('64:1', 'cnn', 4, 2, dict(model_scale=0.75)),
    'c32_cnn_b8_id': ('32', 'cnn', 8, 2, dict(norm='id')),
    'c512thin_test_cnn_b2': ('512thin-test', 'cnn', 2, 1),
    # the remaining families of pluggan.GAN_CONFIGS: the smallest, a 256 px one (with and without attention), the wide one
    'c16_cnn_b16': ('16', 'cnn', 16, 2),
    'c256_iqn_b2': ('256', 'iqn', 2, 1),
    # (batch 8: at batch 2 the BatchNorm statistics of these two sat on a knife edge, tools/knife_edge.py)
    'c256a3_cnn_b8': ('256:3', 'cnn', 8, 1),
    'c128big_cnn_b8': ('128big', 'cnn', 8, 1),
    # BASELINE.json config 4 at its GLOBAL batch (256 = 8 GPUs x 32; replayed as 4 ranks x 64 with SyncBN on one GPU)
    'c128a3_cnn_b256': ('128:3', 'cnn', 256, 1),
    # BASELINE.json config 5 at its GLOBAL batch (512 = 8 GPUs x 64; replayed as 4 ranks x 128 with SyncBN on one GPU)
    'c128a3_iqn_b512': ('128:3', 'iqn', 512, 1),
}

WEIGHT_SEED = 7
RNG_SEED = 1234
IMG_SEED = 4321


def make_args(config, batch, **flags):
    ns = argparse.Namespace(
        config=config, model_scale=1., norm='bn', g_base='mlp', activation='relu',
        lr_g=1e-4, lr_d=4e-4, lr_target_g=1e-3, batch_size=batch,
        grad_penalty=5., device='cpu', run_id='golden', output='/tmp/golden_out',
    )
    for k, v in flags.items():
        assert hasattr(ns, k), k
        setattr(ns, k, v)
    return ns


def build_trainer(config, kind, batch, init_seed=0, **flags):
    cls = {'cnn': CNNTrainer, 'iqn': IQNTrainer}[kind]
    t = object.__new__(cls)          # skip Trainer.__init__ (filesystem side effects only)
    t.args = make_args(config, batch, **flags)
    torch.manual_seed(init_seed)
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):   # build_models prints the nets
        t.build_models()
    return t


def net_summary(module, grads=False, n_samples=4):
    out = {}
    if grads:
        for name, p in module.named_parameters():
            out[name] = summarize(p.grad, n_samples) if p.grad is not None else None
    else:
        for name, v in module.state_dict().items():
            out[name] = summarize(v, n_samples)
    return out


def total_l2(module, grads=False):
    s = 0.0
    for p in module.parameters():
        t = p.grad if grads else p
        if t is not None:
            s += float(t.detach().double().pow(2).sum())
    return s ** 0.5


def run_case(name):
    config, kind, batch, steps = CASES[name][:4]
    flags = CASES[name][4] if len(CASES[name]) > 4 else {}
    t0 = time.time()
    tr = build_trainer(config, kind, batch, **flags)
    size = tr.g.max_size
    fixture = dict(
        case=name, config=config.split(':')[0], flags=flags,
        attention=list(GAN_CONFIGS[config].attention), trainer=kind, batch=batch,
        size=size, weight_seed=WEIGHT_SEED, rng_seed=RNG_SEED, img_seed=IMG_SEED,
        torch_version=torch.__version__, num_threads=torch.get_num_threads(),
        source='reference tartangan v0.4.0 code under torch %s CPU fp32' % torch.__version__,
        blocks=list(tr.gan_config.blocks), latent_dims=tr.gan_config.latent_dims,     # after --model-scale
    )
    # default-init pin (same seed -> same initial parameters, incl. the
    # lr-ignoring update_target_generator(1.) quirk, cnn.py:95,158-165)
    fixture['default_init'] = dict(
        g_l2=total_l2(tr.g), target_g_l2=total_l2(tr.target_g), d_l2=total_l2(tr.d),
        g_first=summarize(next(iter(tr.g.parameters())), 4),
        d_last=summarize(list(tr.d.parameters())[-1], 4),
        target_g_first=summarize(next(iter(tr.target_g.parameters())), 4),
    )
    # procedural weights
    tr.g.load_state_dict(procedural_state(tr.g.state_dict(), WEIGHT_SEED))
    tr.target_g.load_state_dict(procedural_state(tr.target_g.state_dict(), WEIGHT_SEED + 1))
    tr.d.load_state_dict(procedural_state(tr.d.state_dict(), WEIGHT_SEED + 2))
    fixture['n_params'] = dict(
        g=sum(p.numel() for p in tr.g.parameters()),
        d=sum(p.numel() for p in tr.d.parameters()))
    fixture['state_keys'] = dict(g=list(tr.g.state_dict().keys()), d=list(tr.d.state_dict().keys()))

    # model-level forward pins on deep copies (train mode, like the trainers)
    with torch.no_grad():
        g2, d2 = copy.deepcopy(tr.g), copy.deepcopy(tr.d)
        gz = torch.Generator().manual_seed(99)
        z = torch.randn(batch, tr.gan_config.latent_dims, generator=gz)
        imgs0 = synthetic_images(batch, size, IMG_SEED)
        g_out = g2(z)
        fwd = dict(g_out=summarize(g_out, 8))
        if kind == 'cnn':
            fwd['d_real'] = [float(v) for v in d2(imgs0).reshape(-1)]
            fwd['d_fake'] = [float(v) for v in d2(g_out).reshape(-1)]
        else:
            torch.manual_seed(555)
            labels = torch.ones(batch, 1)
            p, loss = d2(imgs0, targets=labels)
            fwd['d_real'] = [float(v) for v in p.reshape(-1)]
            fwd['d_real_loss'] = float(loss)
            torch.manual_seed(555)
            fwd['taus_head'] = [float(v) for v in torch.rand(8 * batch, 1).reshape(-1)[:8]]
        g2.eval()
        fwd['g_out_eval'] = summarize(g2(z), 8)
    fixture['forward'] = fwd

    torch.manual_seed(RNG_SEED)
    fixture['steps'] = []
    for k in range(steps):
        imgs = synthetic_images(batch, size, IMG_SEED + k)
        logs = tr.train_batch(imgs)
        entry = dict(logs)
        entry['g_l2'] = total_l2(tr.g)
        entry['d_l2'] = total_l2(tr.d)
        entry['target_g_l2'] = total_l2(tr.target_g)
        entry['g_grad_l2'] = total_l2(tr.g, grads=True)
        entry['d_grad_l2'] = total_l2(tr.d, grads=True)
        fixture['steps'].append(entry)
        print(f'  {name} step {k + 1}: {logs}  ({time.time() - t0:.1f}s)', flush=True)
        if k == 0:
            fixture['after_step1'] = dict(
                d_grad=net_summary(tr.d, grads=True), g_grad=net_summary(tr.g, grads=True))
    fixture['final'] = dict(g=net_summary(tr.g), d=net_summary(tr.d), target_g=net_summary(tr.target_g))
    fixture['rng_after'] = float(torch.rand(1))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), f'{name}.json')
    with open(path, 'w') as f:
        json.dump(fixture, f, indent=None, separators=(',', ':'))
    print(f'wrote {path} ({os.path.getsize(path) / 1024:.0f} KB, {time.time() - t0:.1f}s)')


def main():
    torch.set_num_threads(1)    # bit-repeatable oracle (SURVEY.md §8c "Determinism")
    names = sys.argv[1:] or list(CASES)
    for n in names:
        run_case(n)


if __name__ == '__main__':
    main()
