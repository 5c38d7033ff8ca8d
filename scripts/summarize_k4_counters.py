"""gpurun_out/k4c_<tag> (scripts/collect_k4_counters.sh) -> profiles/<tag>_pmc_k4.json: instructions and cycles per tile-wave of the K4 tile
kernel (score_topk_bf16_kernel<refine>), both benchmark shapes.

A tile-wave = one wave's 32 users x 32 items of one tile (8 v_mfma_f32_32x32x16_f16 at k = 128).  SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_*
count quad-cycles (MI355X_MICROARCH.md), SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE cycles; counters are summed over the 8 XCCs."""
import glob, json, os, sys
import pandas as pd
src, tag = sys.argv[1], sys.argv[2]
SHAPES = {'ml': dict(users=69878, items=10380, name='69,878 x 10,380, k = 128, top-30'), 'nf': dict(users=480189, items=17770, name='480,189 x 17,770, k = 128, top-30')}
res = {'source': 'rocprofv3 --kernel-trace --pmc <group of 4> around scripts/probe_topk.py (scripts/collect_k4_counters.sh); kernel score_topk_bf16_kernel<8, unsigned short, true, true>',
       'units': 'per_tile_wave: counter / (ceil(users/32) * ceil(items/32)); *_cycles in shader cycles (quad-cycle counters x 4)', 'shapes': {}}
for key, sh in SHAPES.items():
    vals, dur = {}, {}
    for d in sorted(glob.glob(os.path.join(src, key + '_*'))):
        if not os.path.isdir(d):
            continue
        f = glob.glob(d + '/**/b_counter_collection.csv*', recursive=True)
        if not f:
            continue
        t = pd.read_csv(f[0])
        t = t[t.Kernel_Name.str.contains(os.environ.get('K4_KERNEL', 'score_topk_bf16_kernel'))]
        g = t.groupby(['Dispatch_Id', 'Counter_Name']).agg(v=('Counter_Value', 'sum'), s=('Start_Timestamp', 'first'), e=('End_Timestamp', 'first')).reset_index()
        g['dur'] = g.e - g.s
        big = g[g.dur > 0.5 * g.dur.max()]
        for c, gg in big.groupby('Counter_Name'):
            vals[c] = float(gg.v.mean())
            dur[c] = float(gg.dur.mean()) / 1e3
    if not vals:
        continue
    tw = -(-sh['users'] // 32) * -(-sh['items'] // 32)
    wc = vals['SQ_WAVE_CYCLES']
    out = {'shape': sh['name'], 'tile_waves': tw, 'kernel_us_under_counters': dur['SQ_WAVE_CYCLES'], 'counters_per_dispatch': vals}
    pt = lambda c: vals[c] / tw
    out['per_tile_wave'] = {
        'wave_cycles': 4 * pt('SQ_WAVE_CYCLES'), 'waiting_cycles (s_waitcnt / barrier)': 4 * pt('SQ_WAIT_ANY'), 'issue_stall_cycles': 4 * pt('SQ_WAIT_INST_ANY'),
        'issuing_cycles': 4 * pt('SQ_ACTIVE_INST_ANY'), 'mfma_pipe_cycles': pt('SQ_VALU_MFMA_BUSY_CYCLES'),
        'insts_valu (MFMA included)': pt('SQ_INSTS_VALU'), 'insts_salu': pt('SQ_INSTS_SALU'), 'insts_lds': pt('SQ_INSTS_LDS'), 'insts_smem': pt('SQ_INSTS_SMEM'),
        'insts_vmem_rd': pt('SQ_INSTS_VMEM_RD'), 'insts_vmem_wr': pt('SQ_INSTS_VMEM_WR'), 'insts_branch': pt('SQ_INSTS_BRANCH'),
        'lds_bank_conflict_cycles': pt('SQ_LDS_BANK_CONFLICT'), 'lds_active_cycles': pt('SQ_LDS_IDX_ACTIVE')}
    out['fractions_of_wave_cycles'] = {'waiting': vals['SQ_WAIT_ANY'] / wc, 'issue_stall': vals['SQ_WAIT_INST_ANY'] / wc, 'issuing': vals['SQ_ACTIVE_INST_ANY'] / wc,
                                       'issuing_valu': vals['SQ_ACTIVE_INST_VALU'] / wc, 'issuing_scalar': vals['SQ_ACTIVE_INST_SCA'] / wc, 'issuing_lds': vals['SQ_ACTIVE_INST_LDS'] / wc}
    elapsed = vals['GRBM_GUI_ACTIVE'] / 8
    out['mfma_busy_frac'] = vals['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * elapsed)
    out['valu_issue_frac_of_simd_time'] = 4 * vals['SQ_ACTIVE_INST_VALU'] / (1024 * elapsed)
    out['clock_GHz_from_counters'] = elapsed / (dur['GRBM_GUI_ACTIVE'] * 1e3)
    out['resident_waves_per_simd'] = 4 * wc / (1024 * elapsed)
    res['shapes'][key] = out
json.dump(res, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', tag + '_pmc_k4.json'), 'w'), indent=1)
for k, v in res['shapes'].items():
    print(k, json.dumps({a: (round(b, 1) if isinstance(b, float) else b) for a, b in v['per_tile_wave'].items()}))
    print('  ', {a: round(b, 3) for a, b in v['fractions_of_wave_cycles'].items()}, 'mfma busy %.3f' % v['mfma_busy_frac'], 'valu issue %.3f' % v['valu_issue_frac_of_simd_time'], 'waves/SIMD %.2f' % v['resident_waves_per_simd'])
