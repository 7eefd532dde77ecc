"""Dev tool (GPU box): which ATen kernels still run inside one eager step, with shapes and counts."""
import sys, os, collections
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import bench
tr, cfg = bench.make_trainer('128:3', sys.argv[1] if len(sys.argv) > 1 else 'cnn', 64, 'cuda')
imgs = (torch.rand(64, 3, 128, 128) * 2 - 1).cuda()
for _ in range(2): tr.train_batch(imgs)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr.train_batch(imgs)
    torch.cuda.synchronize()
rows = []
for ev in prof.key_averages(group_by_input_shape=True):
    dev = getattr(ev, 'self_device_time_total', 0) or getattr(ev, 'self_cuda_time_total', 0)
    if ev.key.startswith('aten::') and dev > 0:
        rows.append((dev, ev.count, ev.key, str(ev.input_shapes)[:100]))
tot = 0
for dev, n, name, shapes in sorted(rows, reverse=True):
    tot += dev
    print(f'{name:28s} x{n:3d} {dev:9.1f} us  {shapes}')
print('total aten self device us', tot)
