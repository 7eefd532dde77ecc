"""Summarise tools/pmc_conv.sh (rocprofv3 --pmc passes over tools/bench_conv_one.py) per conv kernel:
instruction mix per wave, MFMA-pipe busy fraction, wait fractions, LDS bank conflicts, mean occupancy, L2 hit rate.
Units per /opt/skills/guides/MI355X_MICROARCH.md: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles,
SQ_VALU_MFMA_BUSY_CYCLES cycles (issue cycles x instructions, summed over SIMDs), GRBM_GUI_ACTIVE the sum over the 8 XCDs."""
import csv, glob, collections, json, re, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        kn = r['Kernel_Name']
        if 'conv_' not in kn:
            continue
        name = re.sub(r'\(anonymous namespace\)::', '', kn)
        name = re.sub(r'^void ', '', name).split('(')[0]
        key = (name, r['Grid_Size'], r['LDS_Block_Size'], r['VGPR_Count'], r['Accum_VGPR_Count'])
        agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
out = {}
for (name, grid, lds, vgpr, agpr), c in agg.items():
    m = {k: sum(v) / len(v) for k, v in c.items()}
    waves = m.get('SQ_WAVES', 0) or 1
    cyc = m.get('GRBM_GUI_ACTIVE', 0) / 8.0                      # kernel duration in shader cycles
    wc = m.get('SQ_WAVE_CYCLES', 0) or 1
    d = dict(grid_threads=int(grid), lds_bytes=int(lds), vgpr=int(vgpr), agpr=int(agpr), waves=int(waves),
             kernel_cycles=round(cyc),
             per_wave=dict(mfma=round(m.get('SQ_INSTS_MFMA', 0) / waves, 1), valu_incl_mfma=round(m.get('SQ_INSTS_VALU', 0) / waves, 1),
                           salu=round(m.get('SQ_INSTS_SALU', 0) / waves, 1), lds=round(m.get('SQ_INSTS_LDS', 0) / waves, 1),
                           vmem_rd=round(m.get('SQ_INSTS_VMEM_RD', 0) / waves, 1), vmem_wr=round(m.get('SQ_INSTS_VMEM_WR', 0) / waves, 1)),
             mfma_busy_frac=round(m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (1024.0 * cyc), 4) if cyc else None,
             wait_any_frac=round(m.get('SQ_WAIT_ANY', 0) / wc, 4), wait_inst_any_frac=round(m.get('SQ_WAIT_INST_ANY', 0) / wc, 4),
             active_inst_frac=round(m.get('SQ_ACTIVE_INST_ANY', 0) / wc, 4),
             lds_bank_conflict_frac=round(m.get('SQ_LDS_BANK_CONFLICT', 0) / max(m.get('SQ_LDS_IDX_ACTIVE', 0), 1), 4),
             mean_waves_per_simd=round(wc * 4.0 / (1024.0 * cyc), 2) if cyc else None,
             l2_hit_rate=round(m.get('TCC_HIT_sum', 0) / max(m.get('TCC_HIT_sum', 0) + m.get('TCC_MISS_sum', 0), 1), 4))
    out[f'{name} grid={grid}'] = d
print(json.dumps(dict(source='rocprofv3 --kernel-trace --pmc (4 passes, tools/pmc_conv.sh) on tools/bench_conv_one.py, batch 64',
                      units='quad-cycle counters normalised by SQ_WAVE_CYCLES; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles)',
                      kernels=out), indent=1))
