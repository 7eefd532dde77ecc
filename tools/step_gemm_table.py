import sys, os, collections
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch, bench
from tartangan_amd import backend
K = backend.get()
tr, cfg = bench.make_trainer('128:3', 'cnn', 64, 'cuda')
imgs = (torch.rand(64, 3, 128, 128) * 2 - 1).cuda()
for _ in range(2): tr.train_batch(imgs)
with bench.KernelTimer(K) as kt:
    tr.train_batch(imgs)
torch.cuda.synchronize()
base = kt._empty_pair_ms()
agg = collections.defaultdict(lambda: [0.0, 0])
for name, args, a, b in kt.records:
    if name not in ('gemm', 'softmax_fwd', 'softmax_bwd', 'softmax_dbwd', 'attn_fwd', 'attn_bwd', 'maxpool2_fwd', 'maxpool2_bwd', 'maxpool2_gather', 'scale_add_dev', 'dot', 'scale_dev'): continue
    ints = tuple(x for x in args if isinstance(x, int) and not isinstance(x, bool))
    key = (name,) + (ints[:3] + ints[6:9] if name == 'gemm' else ints[-5:])
    agg[key][0] += max(a.elapsed_time(b) - base, 0); agg[key][1] += 1
for key, (ms, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:30]:
    print(f'{str(key):70s} calls {n:2d} total {ms:6.3f} ms avg {ms/n*1e3:7.1f} us')
