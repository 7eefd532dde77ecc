"""Dev tool (GPU box): per-shape timings of the bandwidth-bound kernels INSIDE the real step, with achieved GB/s."""
import sys, os, collections
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import bench
from tartangan_amd import backend
K = backend.get()
tr, cfg = bench.make_trainer('128:3', 'cnn', 64, 'cuda')
imgs = (torch.rand(64, 3, 128, 128) * 2 - 1).cuda()
for _ in range(2): tr.train_batch(imgs)
with bench.KernelTimer(K) as kt:
    tr.train_batch(imgs)
torch.cuda.synchronize()
PASSES = {'bn_train_stats': 1, 'bn_train_fwd': 3, 'maxpool2_fwd': 1.25, 'maxpool2_bwd': 1.25, 'lrelu_bwd': 3, 'tanh_fwd': 2, 'tanh_bwd': 3, 'scale_add_dev': 3, 'scale_dev': 2, 'bn_act_fwd': 2, 'bn_act_bwd': 5, 'bn_act_dbwd': 8, 'up2x': 1.25, 'pool2': 1.25, 'bilinear_half_fwd': 1.25,
          'bilinear_half_bwd': 1.25, 'add': 3, 'channel_sum': 1}
agg = collections.defaultdict(lambda: [0.0, 0, 0.0])
for name, args, a, b in kt.records:
    if name not in PASSES: continue
    ints = [x for x in args if isinstance(x, int) and not isinstance(x, bool)]
    if name.startswith('bn_') or name == 'channel_sum':
        Bb, C, HW = ints[-4:-1] if name in ('bn_act_bwd', 'channel_sum') else ints[-3:]
        n = Bb * C * HW; key = (name, C, HW)
    elif name in ('add', 'lrelu_bwd', 'tanh_fwd', 'tanh_bwd', 'scale_add_dev', 'scale_dev'):
        n = ints[-1]; key = (name, n, 0)
    else:
        BC, H, W = ints[-3:]
        n = BC * H * W * (4 if name == 'up2x' else 1); key = (name, BC, H)
    ms = a.elapsed_time(b)
    agg[key][0] += ms; agg[key][1] += 1; agg[key][2] += n * 4.0 * PASSES[name]
for key, (ms, n, by) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:45]:
    print(f'{key[0]:18s} {key[1]:8d} {key[2]:6d}  calls {n:2d}  total {ms:6.3f} ms  avg {ms/n*1e3:7.1f} us  {by/ms/1e6:7.0f} GB/s')
