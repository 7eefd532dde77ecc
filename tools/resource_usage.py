"""Dev tool: table of per-kernel register / LDS / occupancy figures from `hipcc -Rpass-analysis=kernel-resource-usage` output."""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
rows, cur = [], None
for line in txt.splitlines():
    m = re.search(r'remark: (.*?) \[-Rpass-analysis', line)
    if not m:
        continue
    body = m.group(1).strip()
    if body.startswith('Function Name:'):
        name = body.split(':', 1)[1].strip()
        try:
            name = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
        except Exception:
            pass
        cur = dict(name=name)
        rows.append(cur)
    elif cur is not None and ':' in body:
        k, v = body.split(':', 1)
        cur[k.strip()] = v.strip()
flt = sys.argv[2] if len(sys.argv) > 2 else ''
for r in rows:
    n = re.sub(r'\(anonymous namespace\)::', '', r['name'])
    n = re.sub(r'\(float.*', '', n).replace('void ', '')
    if flt and not re.search(flt, n):
        continue
    print(f"{n[:78]:78s} V{r.get('VGPRs','?'):>4} A{r.get('AGPRs','?'):>4} S{r.get('TotalSGPRs', r.get('SGPRs','?')):>4} occ {r.get('Occupancy [waves/SIMD]','?'):>2} lds {r.get('LDS Size [bytes/block]','?'):>6} scr {r.get('ScratchSize [bytes/lane]','?')}")
