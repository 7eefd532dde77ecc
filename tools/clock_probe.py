"""Dev tool (GPU box): the shader clock the chip holds under the LDS-DMA conv kernel (MI355X_MICROARCH.md, DVFS give-back item 6).

  make -C tartangan_amd/csrc diag
  TG_LIBRARY=tartangan_amd/csrc/libtartangan_amd_diag.so python tools/clock_probe.py [seconds per arm]

The diagnostic build stamps s_memtime / s_memrealtime around the main loop of conv_dma_kernel (one pair per workgroup, written
to a buffer nothing else reads).  Each arm launches one layer back to back for >= 2 s, then reads the stamps of the LAST launch:
clock = d(memtime) / d(memrealtime) x 100 MHz, median over workgroups.  Arms: random operands, all-zero operands.
`cyc frac` = MFMA cycles the layer needs (FLOPs / 256 per CU per cycle / 256 CUs) / (wall x clock): the share of the cycles that
elapsed on the chip in which its MFMA pipes were needed -- the clock-independent view of the roofline fraction.
"""
import ctypes
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch
from tartangan_amd import backend

DIAG = 'diag' in os.path.basename(backend.LIBRARY)      # product library: timings and the output check only
K = backend.get()
lib = ctypes.CDLL(backend.LIBRARY)
SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 2.5
B = 64
SHAPES = [(16, 16, 128), (32, 16, 128), (32, 32, 64), (64, 32, 64), (64, 64, 32), (128, 64, 32), (128, 128, 16)]
SLOTS = 8192


def stamps():
    buf = (ctypes.c_ulonglong * (2 * SLOTS))()
    rc = lib.tg_diag_read_stamps(buf, SLOTS)
    assert rc == 0, rc
    a = np.frombuffer(buf, dtype=np.uint64).reshape(SLOTS, 2).astype(np.float64)
    a = a[a[:, 1] > 0]
    return a


def phases(n_wg):
    buf = (ctypes.c_ulonglong * (16 * SLOTS))()
    assert lib.tg_diag_read_phases(buf, SLOTS) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(SLOTS, 4, 4).astype(np.float64)[:n_wg]
    return a.reshape(-1, 4).mean(axis=0)           # mean cycles per wave in (vmcnt wait, barrier, DMA issue, MFMA phase)


def life(n_wg):
    buf = (ctypes.c_ulonglong * (16 * SLOTS))()
    assert lib.tg_diag_read_life(buf, SLOTS) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(SLOTS, 4, 4)[:n_wg].astype(np.float64)
    total = a[:, :, 0] + a[:, :, 1] + a[:, :, 2]
    return dict(setup=round(a[:, :, 3].mean()), pre_loop=round(a[:, :, 0].mean()), loop_to_stores_issued=round(a[:, :, 1].mean()),
                stores_ack=round(a[:, :, 2].mean()), life=round(total.mean()))


def arm(fn, flops):
    torch.cuda.synchronize()
    t_end = time.time() + SECONDS
    while time.time() < t_end:                     # keep the queue full: ~1000 launches between host checks
        for _ in range(1000):
            fn()
        torch.cuda.synchronize()
    if DIAG:
        assert lib.tg_diag_clear_stamps() == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 200 * 1e3
    if not DIAG:
        return dict(phase_cycles=dict(wait=0, barrier=0, issue=0, mfma=0), us=round(us, 2), tf=round(flops / us / 1e6, 1), clock_ghz=0.0,
                    loop_us=0.0, workgroups=1, cyc_frac=0.0)
    a = stamps()
    clk = np.median(a[:, 0] / a[:, 1]) * 0.1       # GHz (100 MHz reference counter)
    loop_us = np.median(a[:, 1]) / 100.0
    tf = flops / us / 1e6
    cyc_need = flops / 256.0 / 256.0
    frac = cyc_need / (us * 1e-6 * clk * 1e9)
    ph = phases(len(a))
    return dict(phase_cycles=dict(wait=round(ph[0]), barrier=round(ph[1]), issue=round(ph[2]), mfma=round(ph[3])), life=life(len(a)),
                us=round(us, 2), tf=round(tf, 1), clock_ghz=round(float(clk), 3), loop_us=round(float(loop_us), 2),
                workgroups=int(len(a)), cyc_frac=round(float(frac), 3))


out = []
for Cin, Cout, H in SHAPES:
    g = torch.Generator(device='cuda').manual_seed(Cin * 1000 + Cout * 10 + H)
    flops = 2.0 * B * Cin * Cout * H * H * 9
    row = dict(layer=f'{Cin}->{Cout}@{H}')
    for name in ('random', 'zeros'):
        if name == 'random':
            x = torch.randn(B, Cin, H, H, device='cuda', generator=g)
            w = torch.randn(Cout, Cin, 3, 3, device='cuda', generator=g)
        else:
            x = torch.zeros(B, Cin, H, H, device='cuda')
            w = torch.zeros(Cout, Cin, 3, 3, device='cuda')
        bias = torch.zeros(Cout, device='cuda')
        y = torch.empty(B, Cout, H, H, device='cuda')
        row[name] = arm(lambda: K.conv2d_fwd(x, w, bias, None, y, B, Cin, Cout, H, H, 3), flops)
        if name == 'random':
            want = torch.nn.functional.conv2d(x.cpu(), w.cpu(), bias.cpu(), padding=1).cuda()     # CPU fp32
            row['max_abs_err_vs_miopen'] = float((y - want).abs().max())
            bad = ((y - want).abs() > 1e-3 * want.abs().max()).nonzero()
            if len(bad):
                print('   BAD elements:', len(bad), 'first', bad[:4].tolist(), 'last', bad[-2:].tolist())
            row['ref_max'] = float(want.abs().max())
            print(f"{'':>14}  max |y - cpu conv| {row['max_abs_err_vs_miopen']:.3e} (max |y| {row['ref_max']:.1f})")
    r, z = row['random'], row['zeros']
    print(f"{row['layer']:>14}  random: {r['us']:7.1f} us {r['tf']:6.1f} TF  clock {r['clock_ghz']:.3f} GHz  cyc frac {r['cyc_frac']:.3f}"
          f"   zeros: {z['us']:7.1f} us {z['tf']:6.1f} TF  clock {z['clock_ghz']:.3f} GHz  cyc frac {z['cyc_frac']:.3f}", flush=True)
    p = r['phase_cycles']
    tot = sum(p.values())
    mfma_cycles_wave = flops / 256.0 / 256.0 * 256 * 4 / (r['workgroups'] * 4)      # MFMA-pipe cycles one wave's MFMAs occupy
    print(f"{'':>14}  per wave (random): wait {p['wait']} barrier {p['barrier']} issue {p['issue']} mfma-phase {p['mfma']} cycles"
          f" (sum {tot}; its MFMAs alone occupy the pipe {mfma_cycles_wave:.0f})", flush=True)
    print(f"{'':>14}  workgroup life (random): {r['life']}", flush=True)
    out.append(row)
os.makedirs(os.path.join(REPO, 'gpurun_out'), exist_ok=True)
with open(os.path.join(REPO, 'gpurun_out', 'clock_probe.json'), 'w') as f:
    json.dump(dict(seconds_per_arm=SECONDS, batch=B, peak_f32_mfma_tf_at_2p4ghz=157.3, layers=out), f, indent=1)
