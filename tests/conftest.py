import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN_DIR = os.path.join(REPO, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'slow: long-running CPU test')


def golden_cases():
    # trainer fixtures are named c<size>...; other fixture files (fid_math.json, ...) have their own loaders
    return sorted(f[:-5] for f in os.listdir(GOLDEN_DIR) if f.endswith('.json') and f[0] == 'c' and f[1].isdigit())


def load_golden(name):
    import json
    with open(os.path.join(GOLDEN_DIR, name + '.json')) as f:
        return json.load(f)


@pytest.fixture
def single_thread():
    import torch
    n = torch.get_num_threads()
    torch.set_num_threads(1)
    yield
    torch.set_num_threads(n)


def trainer_from_fixture(fx, device, seed=0):
    """The HIP-engine trainer a fixture describes (config, attention, trainer kind, batch and the option flags
    make_golden.py passed to the reference trainer), built from ``seed`` like the reference was."""
    import torch
    from tartangan_amd.models.pluggan import GAN_CONFIGS
    from tartangan_amd.trainers.cnn import CNNTrainer
    from tartangan_amd.trainers.iqn import IQNTrainer
    cls = {'cnn': CNNTrainer, 'iqn': IQNTrainer}[fx['trainer']]
    cfg = GAN_CONFIGS[fx['config']]._replace(attention=tuple(fx['attention']))
    tr = cls(cls.default_args(config=cfg, batch_size=fx['batch'], device=device, **fx.get('flags', {})))
    torch.manual_seed(seed)
    tr.build_models()
    return tr
