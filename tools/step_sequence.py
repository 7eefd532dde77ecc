"""Dev tool: the launches of ONE replayed steady-state step in order -- index, duration, idle gap before it, grid, kernel
(rocprofv3 --kernel-trace CSV of bench.py).  usage: step_sequence.py <dir with *kernel_trace.csv> [steps back from the end]"""
import csv, glob, os, re, sys
p = max(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True), key=os.path.getmtime)
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = list(csv.DictReader(open(p)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
short = lambda n: re.sub(r'\(anonymous namespace\)::|^void |at::native::|\(.*$', '', n)[:90]
idx = [i for i, r in enumerate(rows) if 'ema_kernel' in r['Kernel_Name']]
a, b = idx[-back - 1], idx[-back]
busy = gap_total = 0.0
for i in range(a + 1, b + 1):
    r = rows[i]
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - int(rows[i - 1]['End_Timestamp'])) / 1e3
    busy += (e - s) / 1e3
    gap_total += max(gap, 0.0)
    wg = r.get('Workgroup_Size', r.get('Workgroup_Size_X', '?'))
    print(f"{i - a:4d} {(e - s) / 1e3:7.1f} us  gap {gap:6.1f}  grid {r.get('Grid_Size', r.get('Grid_Size_X', '?')):>9} wg {wg:>5}  {short(r['Kernel_Name'])}")
print(f'# launches {b - a}  busy {busy / 1e3:.3f} ms  gaps {gap_total / 1e3:.3f} ms  span {(int(rows[b]["End_Timestamp"]) - int(rows[a]["End_Timestamp"])) / 1e6:.3f} ms')
