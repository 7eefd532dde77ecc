"""Dev tool (GPU box): step-1 parity of the HIP trainers against the CPU oracle for ANY config (incl. the ones without a
committed reference fixture: odd channel counts, 8-channel layers), from identical procedural weights and RNG."""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from oracle import sagan_cpu as O
from oracle.procedural import procedural_state, synthetic_images
from tartangan_amd.models.pluggan import GAN_CONFIGS
from tartangan_amd.trainers.cnn import CNNTrainer
from tartangan_amd.trainers.iqn import IQNTrainer

for spec in sys.argv[1:]:
    name, kind, batch = spec.split(',')
    batch = int(batch)
    cfg = GAN_CONFIGS[name]
    size = cfg.base_size * 2 ** len(cfg.blocks)
    cls = {'cnn': CNNTrainer, 'iqn': IQNTrainer}[kind]
    tr = cls(cls.default_args(config=cfg, batch_size=batch, device='cuda'))
    torch.manual_seed(0)
    tr.build_models()
    torch.manual_seed(0)
    O.CONFIGS.setdefault(name, O.GANConfig(cfg.base_size, cfg.latent_dims, cfg.data_dims, tuple(cfg.blocks), tuple(cfg.attention)))
    ref = O.OracleTrainer(name, kind, batch, attention=tuple(cfg.attention) if cfg.attention else None)
    for mine, theirs, seed in ((tr.g, ref.g, 7), (tr.target_g, ref.target_g, 8), (tr.d, ref.d, 9)):
        mine.load_state_dict(procedural_state(theirs, seed))
    ref.load(g=procedural_state(ref.g, 7), target_g=procedural_state(ref.target_g, 8), d=procedural_state(ref.d, 9))
    imgs = synthetic_images(batch, size, 4321)
    torch.manual_seed(1234)
    got = tr.train_batch(imgs)
    torch.manual_seed(1234)
    want = ref.train_batch(imgs)
    worst = max(abs(got[k] - want[k]) / max(abs(want[k]), 1e-12) for k in want)
    print(f'{name:14s} {kind} b{batch}: ' + '  '.join(f'{k} {got[k]:.6f}/{want[k]:.6f}' for k in want) + f'   worst rel {worst:.2e}', flush=True)
