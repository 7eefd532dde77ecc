"""Dev tool (GPU box): per-shape timings of the conv family and BatchNorm INSIDE the real step (eager, HIP events)."""
import sys, os, collections
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import bench
from tartangan_amd import backend
K = backend.get()
cfg_name = sys.argv[1] if len(sys.argv) > 1 else '128:3'
kind = sys.argv[2] if len(sys.argv) > 2 else 'cnn'
tr, cfg = bench.make_trainer(cfg_name, kind, 64, 'cuda')
S = int(cfg_name.split(':')[0])
imgs = (torch.rand(64, 3, S, S) * 2 - 1).cuda()
for _ in range(2): tr.train_batch(imgs)
with bench.KernelTimer(K) as kt:
    tr.train_batch(imgs)
torch.cuda.synchronize()
S2 = {'upconv3x3_fwd': 5, 'poolconv3x3_fwd': 5, 'upconv3x3_dgrad': 3, 'poolconv3x3_dgrad': 3, 'upconv3x3_wgrad': 5, 'poolconv3x3_wgrad': 5}
BN = {'bn_train_fwd': 13, 'bn_act_bwd': 12, 'bn_act_dbwd': 14, 'bn_act_fwd': 7, 'channel_sum': 3}
BNG = {'bn_train_fwd_groups': 13, 'bn_act_bwd_groups': 12}      # (groups, B per group, C, HW) from there
agg = collections.defaultdict(lambda: [0.0, 0, 0.0])
other = collections.defaultdict(lambda: [0.0, 0])
for name, args, a, b in kt.records:
    ms = a.elapsed_time(b)
    if name == 'conv2d_fwd_up2res':
        Bb, Cin, Cout, H, W = args[5:10]
        key, fl = (name, Cin, Cout, H, 3, Bb), 2.0 * Bb * Cin * Cout * H * W * 9
    elif name in bench.CONV_DIMS:
        Bb, Cin, Cout, H, W, ks = args[bench.CONV_DIMS[name]]
        key, fl = (name, Cin, Cout, H, ks, Bb), 2.0 * Bb * Cin * Cout * H * W * ks * ks
    elif name in S2:
        Bb, Cin, Cout, H, W = args[S2[name]:S2[name] + 5]
        key, fl = (name, Cin, Cout, H, 4, Bb), 2.0 * Bb * Cin * Cout * H * W * 16
    elif name in BN:
        Bb, C, HW = args[BN[name]:BN[name] + 3]
        key, fl = (name, C, C, int(HW ** 0.5), 0, Bb), 0.0
    elif name in BNG:
        G, Bb, C, HW = args[BNG[name]:BNG[name] + 4]
        key, fl = (name, C, C, int(HW ** 0.5), 0, G * Bb), 0.0
    else:
        other[name][0] += ms; other[name][1] += 1
        continue
    agg[key][0] += ms; agg[key][1] += 1; agg[key][2] += fl
tot = 0
for key, (ms, n, fl) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    tot += ms
    print(f'{key[0]:22s} B{key[5]:4d} {key[1]:4d}->{key[2]:4d} @{key[3]:3d} k{key[4]}  calls {n:2d}  total {ms:6.3f} ms  avg {ms/n*1e3:7.1f} us  {fl/ms/1e9:6.1f} TF')
print('total conv+bn ms', tot)
for name, (ms, n) in sorted(other.items(), key=lambda kv: -kv[1][0]):
    print(f'{name:26s} calls {n:3d} total {ms:6.3f} ms')
print('total other ms', sum(v[0] for v in other.values()))
