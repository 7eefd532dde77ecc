"""Dev tool (GPU box): per-step deviation of the HIP trainer from every golden fixture, and
per-tensor G-phase gradient deviation from the oracle."""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'tests'))
import torch
from conftest import golden_cases, load_golden
from oracle import sagan_cpu as O
from oracle.procedural import procedural_state, synthetic_images
import test_parity_gpu as T

rel = lambda a, b: abs(a - b) / max(abs(b), 1e-30)
cases = sys.argv[1:] or golden_cases()
for case in cases:
    fx = load_golden(case)
    tr = T.make_trainer(fx)
    tr.g.load_state_dict(procedural_state(tr.g.state_dict(), fx['weight_seed']))
    tr.target_g.load_state_dict(procedural_state(tr.target_g.state_dict(), fx['weight_seed'] + 1))
    tr.d.load_state_dict(procedural_state(tr.d.state_dict(), fx['weight_seed'] + 2))
    torch.manual_seed(fx['rng_seed'])
    for k, ref in enumerate(fx['steps']):
        logs = tr.train_batch(synthetic_images(fx['batch'], fx['size'], fx['img_seed'] + k))
        print(case, k, {n: f"{rel(logs[n], ref[n]):.1e}" for n in logs},
              'g_grad %.1e d_grad %.1e' % (rel(T._total_l2(tr.g, True), ref['g_grad_l2']), rel(T._total_l2(tr.d, True), ref['d_grad_l2'])), flush=True)
        if k == 0:
            worst = []
            for name, p in tr.d.named_parameters():
                r = fx['after_step1']['d_grad'][name]
                got = float(p.grad.double().pow(2).sum().sqrt())
                worst.append((abs(got - r['l2']) / max(ref['d_grad_l2'], 1e-30), name, got, r['l2']))
            worst.sort(reverse=True)
            print('   worst d_grad tensors (err / total l2):', [(f'{w[0]:.1e}', w[1]) for w in worst[:3]])
