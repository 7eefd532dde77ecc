import sys, json
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch
from conftest import load_golden
from oracle import sagan_cpu as O
from oracle.procedural import procedural_state, synthetic_images
torch.set_num_threads(8)
for case in sys.argv[1:]:
    fx = load_golden(case)
    vals = []
    for trial in range(int(__import__("os").environ.get("KNIFE_TRIALS", "6"))):
        flags = dict(fx.get('flags', {})); flags.pop('model_scale', None)
        torch.manual_seed(0)
        tr = O.OracleTrainer(fx['config'], fx['trainer'], fx['batch'], attention=fx['attention'], blocks=fx.get('blocks'), latent_dims=fx.get('latent_dims'), **flags)
        tr.load(g=procedural_state(tr.g, fx['weight_seed']), target_g=procedural_state(tr.target_g, fx['weight_seed'] + 1),
                d=procedural_state(tr.d, fx['weight_seed'] + 2))
        torch.manual_seed(fx['rng_seed'])
        imgs = synthetic_images(fx['batch'], fx['size'], fx['img_seed'])
        if trial:
            g = torch.Generator().manual_seed(trial)
            imgs = imgs * (1 + 1e-7 * torch.randn(imgs.shape, generator=g))
        logs = tr.train_batch(imgs)
        vals.append((logs['g_loss'], logs['d_loss'], logs['gp']))
        print(case, trial, logs, flush=True)
    ref = fx['steps'][0]
    for i, name in enumerate(('g_loss', 'd_loss', 'gp')):
        v = [x[i] for x in vals]
        print(f'{case} {name}: reference {ref[name]:.8g}; oracle unperturbed {v[0]:.8g}; spread under 1e-7 input noise {(max(v) - min(v)) / abs(ref[name]):.2e} rel')
