"""Dev tool: per-step summary of a rocprofv3 --kernel-trace --stats kernel_stats.csv (steps inferred from the Adam launches)."""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
adam = [r for r in rows if 'adam' in r['Name']]
nsteps = int(adam[0]['Calls']) / 2 if adam else 1
tot = sum(float(r['TotalDurationNs']) for r in rows)
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n); n = re.sub(r'^void ', '', n)
    return re.sub(r'\(.*', '', n)[:76]
print(f'steps profiled {nsteps:.0f}  GPU-busy ms/step {tot / nsteps / 1e6:.3f}  launches/step {sum(int(r["Calls"]) for r in rows) / nsteps:.0f}')
fam = collections.defaultdict(float)
for r in rows:
    n = short(r['Name'])
    key = ('conv fwd/dgrad' if re.match(r'conv_(dma|fwd|upfwd|upT|upfwd_dma|upT_dma|wino_dma|poolwino_dma)_kernel|conv1x1', n) else
           'conv wgrad' if 'wgrad' in n else 'batchnorm' if re.match(r'planes::|bn_|chan_stage', n) else
           'attention' if n.startswith('attn') else 'aten' if n.startswith('at::') else 'other')
    fam[key] += float(r['TotalDurationNs']) / nsteps / 1e6
print('  '.join(f'{k} {v:.2f} ms' for k, v in sorted(fam.items(), key=lambda kv: -kv[1])))
for r in rows[:top]:
    print(f"{short(r['Name']):78s} x{int(r['Calls']) / nsteps:5.1f} avg {float(r['AverageNs']) / 1e3:7.1f} us {float(r['TotalDurationNs']) / nsteps / 1e6:6.3f} ms/step")
