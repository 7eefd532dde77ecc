"""Summarise tools/pmc_traffic.sh: per kernel family, HBM bytes per launch from FETCH_SIZE / WRITE_SIZE.

Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): both counters are in KiB; on gfx950 FETCH_SIZE
tallies the 128-byte requests of wide (16 B / lane) streaming reads at 64 bytes, i.e. reports half the bytes -> x2.
WRITE_SIZE is exact for 16 B / lane streaming stores; the conv epilogue stores 4 B / lane (uncalibrated width, taken as is).
"""
import csv, glob, json, re, sys, collections

def load(d, counter):
    f = glob.glob(f'{d}/**/*counter_collection.csv', recursive=True)
    agg = collections.defaultdict(lambda: [0, 0.0])
    for p in f:
        for r in csv.DictReader(open(p)):
            if r['Counter_Name'] != counter:
                continue
            n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
            n = re.sub(r'^void ', '', n).split('(')[0]
            fam = re.sub(r'<.*', '', n)
            if fam == 'conv_fwd_kernel':
                ks = re.search(r'conv_fwd_kernel<Geo<[^>]*>, (\d)', n)
                fam += ' 1x1' if ks and ks.group(1) == '1' else ''
                fam += ' (dgrad)' if re.search(r', true>$', n) else ' (fwd)'
            elif fam == 'conv_dma_kernel':
                fam += ' (dgrad)' if re.search(r', true, (true|false)>$', n) else ' (fwd)'
            elif fam == 'conv_wino_dma_kernel':
                fam += ' (dgrad)' if re.search(r'conv_wino_dma_kernel<Geo<[^>]*>, \d+, true', n) else ' (fwd)'
            agg[fam][0] += 1
            agg[fam][1] += float(r['Counter_Value'])
    return agg

root = sys.argv[1]
fetch, write = load(f'{root}/fetch', 'FETCH_SIZE'), load(f'{root}/write', 'WRITE_SIZE')
out = {}
for fam in sorted(set(fetch) | set(write), key=lambda k: -(fetch.get(k, [0, 0])[1] + write.get(k, [0, 0])[1])):
    nf, sf = fetch.get(fam, [0, 0.0]); nw, sw = write.get(fam, [0, 0.0])
    if not nf or not nw:
        continue
    rd = 2.0 * sf * 1024 / nf          # gfx950 correction
    wr = sw * 1024 / nw
    out[fam] = {'launches_sampled': nf, 'fetch_bytes_per_launch': round(rd), 'write_bytes_per_launch': round(wr),
                'hbm_bytes_per_launch': round(rd + wr)}
import hashlib, os
here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sha = hashlib.sha256(b''.join(open(os.path.join(here, 'tartangan_amd', 'csrc', f), 'rb').read() for f in ('conv.hip', 'wino.h'))).hexdigest()
# the roofline family of bench.py: every kernel that produces activations / activation gradients of a 3x3 or stride-2 conv
FAMILY = ('conv_fwd_kernel (fwd)', 'conv_fwd_kernel (dgrad)', 'conv_dma_kernel (fwd)', 'conv_dma_kernel (dgrad)', 'conv_upfwd_kernel',
          'conv_upT_kernel', 'conv_upfwd_dma_kernel', 'conv_upT_dma_kernel', 'conv_wino_dma_kernel (fwd)', 'conv_wino_dma_kernel (dgrad)',
          'conv_poolwino_dma_kernel')          # (1x1 kernels: priced against HBM, listed apart)
fl = sum(out[k]['launches_sampled'] for k in FAMILY if k in out)
fb = sum(out[k]['launches_sampled'] * out[k]['hbm_bytes_per_launch'] for k in FAMILY if k in out)
family = {'kernels': [k for k in FAMILY if k in out], 'launches_sampled': fl, 'hbm_bytes_per_launch': round(fb / max(fl, 1))}
print(json.dumps({'conv_hip_sha256': sha, 'conv_family_fwd_dgrad': family, 'source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on bench.py --steps 2 --warmup 1',
                  'corrections': 'KiB -> bytes; FETCH_SIZE x2 on gfx950 (128-B requests tallied at 64 B)',
                  'kernels': dict(list(out.items())[:40])}, indent=1))
