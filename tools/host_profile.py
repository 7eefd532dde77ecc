"""Dev tool (GPU box): where the HOST time of a graph-replayed step goes (cProfile over N steps) and how long the GPU idles
between steps (events around consecutive steps vs. rocprof GPU-busy)."""
import sys, os, time, cProfile, pstats
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import bench
tr, cfg = bench.make_trainer('128:3', sys.argv[1] if len(sys.argv) > 1 else 'cnn', 64, 'cuda')
tr.enable_graphs()
imgs = (torch.rand(64, 3, 128, 128) * 2 - 1).cuda()
for _ in range(5):
    tr.train_batch(imgs)
torch.cuda.synchronize()
N = 100
t0 = time.perf_counter()
for _ in range(N):
    tr.train_batch(imgs)
torch.cuda.synchronize()
print(f'{(time.perf_counter() - t0) / N * 1e3:.3f} ms/step wall')
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    tr.train_batch(imgs)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(22)
# host time from the loss read-back of one step to the first replay of the next = GPU idle
import tartangan_amd.trainers.cnn as C
marks = []
orig_replay = torch.cuda.CUDAGraph.replay
def replay(self):
    marks.append(('replay', time.perf_counter()))
    return orig_replay(self)
torch.cuda.CUDAGraph.replay = replay
for _ in range(20):
    out = tr.train_batch(imgs)
    marks.append(('done', time.perf_counter()))
gaps = []
for (k0, t0), (k1, t1) in zip(marks, marks[1:]):
    if k0 == 'done' and k1 == 'replay':
        gaps.append(t1 - t0)
print('host gap between a step\'s read-back and the next step\'s first replay: median %.1f us' % (sorted(gaps)[len(gaps) // 2] * 1e6))
