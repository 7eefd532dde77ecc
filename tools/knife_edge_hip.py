"""Dev tool (GPU box): step-1 losses of the HIP trainer for a fixture with the input images perturbed by 1e-7 relative
(trial 0: unperturbed) -- is a deviation from the fixture systematic, or does it move with the rounding pattern?"""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'tests'))
import torch
from conftest import load_golden, trainer_from_fixture
from oracle.procedural import procedural_state, synthetic_images
from tartangan_amd.models.blocks import SelfAttention2d
fx = load_golden(sys.argv[1])
ref = fx['steps'][0]
for fused in (True, False):
    SelfAttention2d.fuse_projections = fused
    for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 5):
        tr = trainer_from_fixture(fx, 'cuda')
        tr.g.load_state_dict(procedural_state(tr.g.state_dict(), fx['weight_seed']))
        tr.target_g.load_state_dict(procedural_state(tr.target_g.state_dict(), fx['weight_seed'] + 1))
        tr.d.load_state_dict(procedural_state(tr.d.state_dict(), fx['weight_seed'] + 2))
        imgs = synthetic_images(fx['batch'], fx['size'], fx['img_seed'])
        if trial:
            g = torch.Generator().manual_seed(trial)
            imgs = imgs * (1 + 1e-7 * torch.randn(imgs.shape, generator=g))
        torch.manual_seed(fx['rng_seed'])
        logs = tr.train_batch(imgs)
        print('fused' if fused else 'sep  ', trial, {k: f'{(logs[k] - ref[k]) / abs(ref[k]):+.2e}' for k in logs}, flush=True)
