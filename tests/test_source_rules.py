"""Source-level rules that a compile or a numerics test cannot see until they bite.

LDS-DMA (`dma16`, buffer_load ... lds) takes its LDS destination from a wave-uniform register; ROCm 7.2 may merge two calls
across a DIVERGENT branch and then reads that register from the first active lane for all lanes (filter
rows half loaded in conv_dma_kernel<G16, 32, 1, 8>).  The rule in csrc/conv.hip: a `dma16` call is guarded by nothing, or by a
condition made of compile-time constants and the scalar wave index only -- never by a per-lane value."""
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
CONV = os.path.join(os.path.dirname(HERE), 'tartangan_amd', 'csrc', 'conv.hip')
SCALAR_TOKEN = re.compile(r'^(i|v|wave|CT_THREADS|GB_T|PCHP|WCHP|GCHP|ACH|BCH|\d+)$')


FID = os.path.join(os.path.dirname(HERE), 'tartangan_amd', 'csrc', 'fid.hip')
WINO = os.path.join(os.path.dirname(HERE), 'tartangan_amd', 'csrc', 'wino.h')       # (included by conv.hip)


def _guards(path=CONV):
    lines = open(path).read().split('\n')
    out = []
    for n, line in enumerate(lines):
        if 'dma16(' not in line or '__device__' in line:
            continue
        stmt = line.strip()
        if not stmt.startswith('if'):                      # the guard sits on the line above
            stmt = re.sub(r'\s*//.*$', '', lines[n - 1].strip()) + ' ' + stmt
        m = re.match(r'if \((.*)\)\s+(?:gb_)?dma16\(', stmt)
        out.append((n + 1, m.group(1) if m else None, stmt))
    return out


def test_every_lds_dma_sits_under_a_scalar_condition_only():
    guards = _guards()
    assert len(guards) >= 10                               # conv_dma, wgrad, upT, upfwd, s2 wgrad: two regions each
    fid = _guards(FID)
    assert len(fid) == 2                                   # gemm_big_kernel: the A and the B panel
    wino = _guards(WINO)
    assert len(wino) == 2                                  # conv_wino_dma_kernel, conv_poolwino_dma_kernel: the patch
    for lineno, cond, stmt in guards + fid + wino:
        assert cond is not None, f'csrc line {lineno}: unguarded or unparsable dma16 call: {stmt}'
        assert 'threadIdx' not in cond and 'lane' not in cond, f'csrc line {lineno}: per-lane guard on an LDS-DMA: {cond}'
        for tok in re.findall(r'[A-Za-z_]\w*|\d+', cond):
            assert SCALAR_TOKEN.match(tok), f'csrc line {lineno}: `{tok}` in the guard of an LDS-DMA is not known to be wave-uniform: {cond}'


def test_wave_index_is_scalar_where_it_guards_a_dma():
    src = open(CONV).read()
    assert '__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)' in src
    # every kernel that issues DMAs takes `wave` from wave_index(), not from threadIdx directly
    src = src + open(WINO).read()
    for m in re.finditer(r'__global__[^{]*?(\w+_dma_kernel)\(', src):
        body = src[m.end():src.index('\n}\n', m.end())]
        assert 'wave = wave_index()' in body, f'{m.group(1)}: wave must be the scalar wave_index()'
