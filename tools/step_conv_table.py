"""Dev tool (GPU box): per-shape conv timings INSIDE the real step (eager, HIP events)."""
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
agg = collections.defaultdict(lambda: [0.0, 0, 0.0])
for name, args, a, b in kt.records:
    if not name.startswith('conv2d'): continue
    Bb, Cin, Cout, H, W, ks = args[bench.CONV_DIMS[name]]
    key = (name, Cin, Cout, H, ks)
    ms = a.elapsed_time(b)
    agg[key][0] += ms; agg[key][1] += 1; agg[key][2] += 2.0 * Bb * Cin * Cout * H * W * ks * ks
tot = 0
for key, (ms, n, fl) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    tot += ms
    print(f'{key[0]:13s} {key[1]:4d}->{key[2]:4d} @{key[3]:3d} k{key[4]}  calls {n:2d}  total {ms:6.3f} ms  avg {ms/n*1e3:7.1f} us  {fl/ms/1e9:6.1f} TF')
print('total conv ms', tot)
