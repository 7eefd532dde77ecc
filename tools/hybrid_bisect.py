"""Dev tool (GPU box): which kernel family costs end-to-end accuracy?  Runs the D phase of a fixture
case with selected C-ABI entry points re-routed to the CPU fp32 emulator (device<->host copies) and
reports gp / d_loss / d_grad error against the fp64 oracle."""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'tests'))
import torch
import torch.nn.functional as F
from conftest import load_golden
from emulator import Emulator
from oracle import sagan_cpu as O
from oracle.procedural import procedural_state, synthetic_images
import test_parity_gpu as T
from tartangan_amd import backend, functional as TF
from tartangan_amd.models.losses import gradient_penalty
from tartangan_amd.trainers.utils import toggle_grad

case = sys.argv[1]
fx = load_golden(case); B = fx['batch']
torch.manual_seed(0)
ref = O.OracleTrainer(fx['config'], fx['trainer'], B, attention=fx['attention'])
gs, ds = procedural_state(ref.g, fx['weight_seed']), procedural_state(ref.d, fx['weight_seed'] + 2)
imgs = synthetic_images(B, fx['size'], fx['img_seed'])
z = torch.randn(B, ref.cfg.latent_dims, generator=torch.Generator().manual_seed(5))

def truth():
    dt = torch.float64
    S = {k: (v.clone().to(dt) if v.is_floating_point() else v.clone()) for k, v in ds.items()}
    G = {k: (v.clone().to(dt) if v.is_floating_point() else v.clone()) for k, v in gs.items()}
    for k, v in S.items():
        if O.is_param(k): v.requires_grad_(True)
    with torch.no_grad():
        fake = O.g_forward(G, z.to(dt), ref.cfg)
    real = imgs.to(dt).clone().requires_grad_()
    labels = torch.zeros(2 * B, 1, dtype=dt); labels[:B] = 1
    p_real = O.d_forward(S, real, ref.cfg); p_fake = O.d_forward(S, fake, ref.cfg)
    bce = F.binary_cross_entropy_with_logits(torch.cat([p_real, p_fake]), labels)
    gp = 5.0 * O.gradient_penalty(p_real, real)
    (bce + gp).backward()
    return fake.float(), float(bce), float(gp), {k: v.grad for k, v in S.items() if O.is_param(k)}

FAKE, BCE, GP, GRADS = truth()
HIP = backend.get()
EMU = Emulator()
ALL = [n for n in dir(EMU) if not n.startswith('_') and callable(getattr(EMU, n)) and hasattr(HIP, n)]

class Hybrid:
    name = 'hybrid'
    def __init__(self, cpu_ops):
        for n in ALL:
            setattr(self, n, self._cpu(n) if n in cpu_ops else getattr(HIP, n))
    def _cpu(self, n):
        def call(*args):
            host = [a.detach().cpu() if torch.is_tensor(a) else a for a in args]
            before = [h.clone() if torch.is_tensor(h) else None for h in host]
            rc = getattr(EMU, n)(*host)
            for a, h, b in zip(args, host, before):
                if torch.is_tensor(a) and not torch.equal(h, b): a.data.copy_(h)
            return rc
        return call

def run(cpu_ops, label):
    backend._set_backend_for_testing(Hybrid(set(cpu_ops)))
    tr = T.make_trainer(fx)
    tr.g.load_state_dict(gs); tr.d.load_state_dict(ds)
    tr.g.train(); tr.d.train()
    toggle_grad(tr.g, False); toggle_grad(tr.d, True); tr.optimizer_d.zero_grad()
    real = imgs.cuda().requires_grad_()
    labels = torch.zeros(2 * B, 1, device='cuda'); labels[:B] = 1
    p_real = tr.d(real); p_fake = tr.d(FAKE.cuda())
    bce = TF.bce_with_logits(torch.cat([p_real, p_fake]), labels)
    gp = TF.scale(gradient_penalty(p_real, real), 5.0)
    TF.add(bce, gp).backward()
    tl2 = sum(float(v.double().pow(2).sum()) for v in GRADS.values()) ** 0.5
    diff = sum(float((p.grad.cpu().double() - GRADS[n]).pow(2).sum()) for n, p in tr.d.named_parameters()) ** 0.5
    rel = lambda a, b: abs(a - b) / abs(b)
    print('%-34s bce %.1e  gp %.1e  d_grads %.1e' % (label, rel(float(bce), BCE), rel(float(gp), GP), diff / tl2), flush=True)

CONV = ['conv2d_fwd', 'conv2d_dgrad', 'conv2d_wgrad']
BN = ['bn_train_stats', 'bn_act_fwd', 'bn_act_bwd', 'bn_act_dbwd']
ATT = ['gemm', 'softmax_fwd', 'softmax_bwd', 'softmax_dbwd', 'maxpool2_fwd', 'maxpool2_bwd', 'maxpool2_gather']
RES = ['up2x', 'pool2', 'bilinear_half_fwd', 'bilinear_half_bwd']
run([], 'all HIP')
for n in RES:
    run([n], n + ' on CPU')
x = torch.randn(12, 128, 128)
y_cpu = torch.zeros(12, 64, 64); EMU.bilinear_half_fwd(x, y_cpu, 12, 128, 128)
y_hip = torch.zeros(12, 64, 64).cuda(); HIP.bilinear_half_fwd(x.cuda(), y_hip, 12, 128, 128)
y64 = torch.nn.functional.interpolate(x.double()[None], scale_factor=0.5, mode='bilinear', align_corners=True)[0]
print('bilinear fwd 128: hip-vs-cpu32 %.2e  hip-vs-f64 %.2e  cpu32-vs-f64 %.2e' % (float((y_hip.cpu() - y_cpu).abs().max()), float((y_hip.cpu().double() - y64).abs().max()), float((y_cpu.double() - y64).abs().max())))
g = torch.randn(12, 64, 64)
gx_cpu = torch.zeros(12, 128, 128); EMU.bilinear_half_bwd(g, None, gx_cpu, 12, 128, 128)
gx_hip = torch.zeros(12, 128, 128).cuda(); HIP.bilinear_half_bwd(g.cuda(), None, gx_hip, 12, 128, 128)
gx64 = torch.zeros(12, 128, 128, dtype=torch.float64); EMU.bilinear_half_bwd(g.double(), None, gx64, 12, 128, 128)
print('bilinear bwd 128: hip-vs-cpu32 %.2e  hip-vs-f64 %.2e  cpu32-vs-f64 %.2e' % (float((gx_hip.cpu() - gx_cpu).abs().max()), float((gx_hip.cpu().double() - gx64).abs().max()), float((gx_cpu.double() - gx64).abs().max())))
