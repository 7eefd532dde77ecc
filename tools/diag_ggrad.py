"""Dev tool: per-parameter G-phase gradient error vs the oracle for one fixture."""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'tests'))
import torch
from conftest import load_golden
from oracle import sagan_cpu as O
from oracle.procedural import procedural_state
import test_parity_gpu as T
case = sys.argv[1] if len(sys.argv) > 1 else 'c64a1_cnn_b8'
fx = load_golden(case)
tr = T.make_trainer(fx) if hasattr(T, 'make_trainer') else None
torch.manual_seed(0)
ref = O.OracleTrainer(fx['config'], fx['trainer'], fx['batch'], attention=fx['attention'])
gs, ds = procedural_state(ref.g, fx['weight_seed']), procedural_state(ref.d, fx['weight_seed'] + 2)
ref.load(g=gs, d=ds)
tr.g.load_state_dict(gs); tr.d.load_state_dict(ds)
tr.g.train(); tr.d.train()
torch.manual_seed(77)
g_loss = float(tr._g_phase(fx['batch']))
torch.manual_seed(77)
ref._toggle(ref.g, True); ref._toggle(ref.d, False)
fake = O.g_forward(ref.g, ref.sample_z(fx['batch']), ref.cfg)
ones = torch.ones(fx['batch'], 1)
want = torch.nn.functional.binary_cross_entropy_with_logits(ref._d(fake), ones)
want.backward()
print('g_loss', g_loss, float(want))
for name, p in tr.g.named_parameters():
    w = ref.g[name].grad
    err = float((p.grad.cpu() - w).abs().max())
    g = p.grad.cpu().flatten(); wf = w.flatten()
    top = wf.abs().topk(min(8, wf.numel())).indices
    ratio = (g[top] / wf[top])
    print(f'{name:40s} err {err:.3e}  max {float(w.abs().max()):.3e}  rel {err/float(w.abs().max()):.2e}  ratio(top) min {float(ratio.min()):.5f} max {float(ratio.max()):.5f}')
