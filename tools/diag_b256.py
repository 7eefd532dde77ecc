"""Dev tool (GPU box): the c128a3_cnn_b256 fixture on one GPU, paired and unpaired D pass."""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'tests'))
import torch
from conftest import load_golden, trainer_from_fixture
from oracle.procedural import procedural_state, synthetic_images
fx = load_golden(sys.argv[1] if len(sys.argv) > 1 else 'c128a3_cnn_b256')
from tartangan_amd.models.blocks import SelfAttention2d
for pair in (True, False):
    SelfAttention2d.fuse_projections = pair         # (argv[2] == 'qkv': toggle the fused attention projections instead of the D pairing)
    tr = trainer_from_fixture(fx, 'cuda')
    if len(sys.argv) < 3 or sys.argv[2] != 'qkv':
        SelfAttention2d.fuse_projections = True
        tr.args.pair_d = pair
    tr.g.load_state_dict(procedural_state(tr.g.state_dict(), fx['weight_seed']))
    tr.target_g.load_state_dict(procedural_state(tr.target_g.state_dict(), fx['weight_seed'] + 1))
    tr.d.load_state_dict(procedural_state(tr.d.state_dict(), fx['weight_seed'] + 2))
    torch.manual_seed(fx['rng_seed'])
    logs = tr.train_batch(synthetic_images(fx['batch'], fx['size'], fx['img_seed']))
    ref = fx['steps'][0]
    print('pair' if pair else 'sep ', {k: (logs[k], ref[k], abs(logs[k] - ref[k]) / abs(ref[k])) for k in logs}, flush=True)
