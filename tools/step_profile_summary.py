"""Dev tool: per-kernel GPU time of steady-state graph replay from a rocprofv3 --kernel-trace CSV of bench.py."""
import csv, glob, os, re, collections, sys
p = max(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(p)))
names = [r['Kernel_Name'] for r in rows]
idx = [i for i, n in enumerate(names) if 'ema_kernel' in n]
nsteps = min(10, len(idx) - 2)
a, b = idx[-nsteps - 1], idx[-1]
seg = rows[a + 1:b + 1]
d = collections.defaultdict(lambda: [0, 0.0])
for r in seg:
    n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']); n = re.sub(r'^void ', '', n)[:int(sys.argv[3]) if len(sys.argv) > 3 else 64]
    d[n][0] += 1; d[n][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
tot = sum(v[1] for v in d.values())
print('launches/step', len(seg) / nsteps, 'GPU busy ms/step', round(tot / 1e3 / nsteps, 3),
      'wall ms/step', round((int(seg[-1]['End_Timestamp']) - int(seg[0]['Start_Timestamp'])) / 1e6 / nsteps, 3))
for n, (c, t) in sorted(d.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print(f'{t/1e3/nsteps:7.3f} ms  {c/nsteps:6.1f}/step  avg {t/c:7.1f} us  {n}')
