"""Dev tool: where does a kernel sit in the steady-state step?  Lists each launch of kernels matching a pattern in ONE
replayed step with its grid size, duration and neighbours (rocprofv3 --kernel-trace CSV of bench.py)."""
import csv, glob, os, re, sys
p = max(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True), key=os.path.getmtime)
pat = sys.argv[2]
rows = list(csv.DictReader(open(p)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
short = lambda n: re.sub(r'\(anonymous namespace\)::|^void |at::native::', '', n)[:46]
idx = [i for i, r in enumerate(rows) if 'ema_kernel' in r['Kernel_Name']]
a, b = idx[-2], idx[-1]
for i in range(a + 1, b + 1):
    r = rows[i]
    if re.search(pat, r['Kernel_Name']):
        dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        print(f"{i - a:4d} {dur:6.1f} us grid {r.get('Grid_Size', r.get('Grid_Size_X', '?')):>9}  prev: {short(rows[i-1]['Kernel_Name'])} | next: {short(rows[i+1]['Kernel_Name'])}")
