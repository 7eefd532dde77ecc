"""Dev tool (GPU box): run one D phase (+G phase) of a fixture case on the HIP backend and, for
EVERY C-ABI call, recompute the outputs on the CPU in float64 from the same inputs (tests/emulator.py
semantics).  Prints the calls with the largest relative error -> pinpoints an inaccurate kernel/shape."""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'tests'))
import torch
from conftest import load_golden
from emulator import Emulator
from oracle.procedural import procedural_state, synthetic_images
import test_parity_gpu as T
from tartangan_amd import backend

OUTS = {  # name -> indices of output tensor args
    'conv2d_fwd': [3], 'conv2d_dgrad': [2], 'conv2d_wgrad': [2, 3], 'channel_sum': [1], 'channel_bcast': [1], 'gemm': [2],
    'bn_train_stats': [1, 2], 'bn_act_fwd': [6], 'bn_act_bwd': [8, 9, 10], 'bn_act_dbwd': [10, 11, 12],
    'up2x': [1], 'pool2': [1], 'bilinear_half_fwd': [1], 'bilinear_half_bwd': [2], 'maxpool2_fwd': [1], 'maxpool2_bwd': [2],
    'maxpool2_gather': [2], 'row_sum': [1], 'row_bcast': [1], 'add': [2], 'mul': [2], 'scale': [2], 'scale_dev': [3],
    'scale_add_dev': [3], 'dot': [3], 'lrelu_bwd': [3], 'tanh_fwd': [1], 'tanh_bwd': [2], 'softmax_fwd': [1],
    'softmax_bwd': [2], 'softmax_dbwd': [3], 'bce_logits': [2, 3], 'sumsq': [2], 'iqn_loss': [4, 5], 'sum_reps': [1], 'repeat_rows': [1],
}
E = Emulator()
K = backend.get()
records = []

def wrap(name, fn):
    def call(*args):
        pre = [a.detach().cpu().double() if torch.is_tensor(a) and a.dtype == torch.float32 else (a.detach().cpu() if torch.is_tensor(a) else a) for a in args]
        rc = fn(*args)
        if name in ('bn_train_stats',):       # running stats are in/out; only check mean/invstd
            pre[3] = pre[4] = None
        getattr(E, name)(*pre)
        if name == 'bn_act_bwd' and args[8] is not None:
            gz, x, mean, invstd, gamma, beta, slope = pre[:7]
            B_, C_, HW_ = args[-3:]
            xv = x.view(B_, C_, HW_)
            y64 = (xv - mean.view(1, C_, 1)) * invstd.view(1, C_, 1) * gamma.view(1, C_, 1) + beta.view(1, C_, 1)
            a32 = (gamma.float() * invstd.float()); b32 = beta.float() - mean.float() * a32
            y32 = torch.addcmul(b32.view(1, C_, 1), xv.float(), a32.view(1, C_, 1))
            flips = int(((y64 >= 0) != (y32 >= 0)).sum())
            got = args[8].detach().cpu().double().view(B_, C_, HW_); want = pre[8].view(B_, C_, HW_)
            d = (got - want).abs()
            if float(d.max()) > 1e-4 * float(want.abs().max()):
                idx = torch.nonzero(d > 0.5 * d.max())[:5].tolist()
                print('BN_BWD', [B_, C_, HW_], 'mask flips', flips, 'n_bad', int((d > 1e-4 * want.abs().max()).sum()), 'worst at', idx,
                      'per-channel max err (top3):', torch.topk(d.amax((0, 2)), 3), 'y64 near zero count', int((y64.abs() < 1e-5).sum()))
        for i in OUTS[name]:
            if args[i] is None: continue
            want, got = pre[i], args[i].detach().cpu().double()
            scale = float(want.abs().max())
            err = float((want - got).abs().max())
            rms = float((want - got).pow(2).mean().sqrt()) / max(float(want.pow(2).mean().sqrt()), 1e-30)
            dims = [a for a in args if isinstance(a, int)]
            records.append((err / max(scale, 1e-30), rms, name, i, dims))
        return rc
    return call

for name in OUTS:
    setattr(K, name, wrap(name, getattr(K, name)))

case = sys.argv[1]
fx = load_golden(case)
tr = T.make_trainer(fx)
tr.g.load_state_dict(procedural_state(tr.g.state_dict(), fx['weight_seed']))
tr.d.load_state_dict(procedural_state(tr.d.state_dict(), fx['weight_seed'] + 2))
tr.g.train(); tr.d.train()
torch.manual_seed(fx['rng_seed'])
imgs = synthetic_images(fx['batch'], fx['size'], fx['img_seed']).cuda()
tr._d_phase(imgs)
n_d = len(records)
tr._g_phase(fx['batch'])
print(f'{len(records)} checked outputs ({n_d} in the D phase)')
records = [r for r in records if r[2] != 'channel_sum']
for r in sorted(records, reverse=True)[:25]:
    print('max-rel %.2e  rms-rel %.2e  %-18s out%d dims %s' % r)
print('--- by rms-rel')
for r in sorted(records, key=lambda t: -t[1])[:15]:
    print('max-rel %.2e  rms-rel %.2e  %-18s out%d dims %s' % r)
