"""Dev tool (GPU box): two eager steps + one graph-replayed step of every GAN config at a small batch; losses must be finite
and the graph replay must reproduce the eager continuation."""
import sys, os, math
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import bench
from tartangan_amd.models.pluggan import GAN_CONFIGS

which = sys.argv[1:] or [k for k in GAN_CONFIGS]
for name in which:
    cfg = GAN_CONFIGS[name]
    size = cfg.base_size * 2 ** len(cfg.blocks)
    batch = 8 if size <= 128 else (4 if size <= 256 else 2)
    spec = name if not cfg.attention else name        # attention comes with the config
    for kind in ('cnn', 'iqn'):
        try:
            torch.manual_seed(0)
            tr, _ = bench.make_trainer(name, kind, batch, 'cuda')
            imgs = (torch.rand(batch, 3, size, size) * 2 - 1).cuda()
            logs = None
            for _ in range(2):
                logs = tr.train_batch(imgs)
            ok = all(math.isfinite(v) for v in logs.values())
            print(f'{name:14s} {kind} {size:4d}px b{batch}: {"ok " if ok else "NONFINITE"} {logs}', flush=True)
            del tr
            torch.cuda.empty_cache()
        except Exception as e:
            print(f'{name:14s} {kind}: FAILED {type(e).__name__}: {str(e)[:200]}', flush=True)
