"""gpurun_out/mfma_<tag> (scripts/collect_mfma.sh) -> profiles/<tag>_pmc_mfma.json: matrix-pipe busy fraction of the MFMA kernels.

busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x elapsed shader cycles), elapsed shader cycles = GRBM_GUI_ACTIVE of
the same dispatch (per XCC: the counter is summed over the 8 XCCs) -- cross-checked against the kernel's duration in the trace and
against the MFMA instruction count the kernel must issue (MI355X_MICROARCH.md: the counter counts cycles, 32 per 32x32x16 MFMA)."""
import glob, json, os, re, sys
import pandas as pd
src, tag = sys.argv[1], sys.argv[2]
SIMDS, XCCS = 256 * 4, 8


def table(leg, C):
    hits = glob.glob(os.path.join(src, '%s_%s' % (leg, C), '**', 'b_counter_collection.csv*'), recursive=True)
    return pd.read_csv(hits[0]) if hits else None


def per_kernel(leg, C, pat):
    d = table(leg, C)
    if d is None:
        return None
    d = d[d.Kernel_Name.str.contains(pat)]
    if not len(d):
        return None
    g = d.groupby('Dispatch_Id').agg(v=('Counter_Value', 'sum'), s=('Start_Timestamp', 'first'), e=('End_Timestamp', 'first'))
    return g


LEGS = (('topk', r'score_topk_refine2_kernel', 'score_topk_refine2_kernel<refine>, 69,878 x 10,380, k = 128',
         dict(users=69878, items=10380, k=128, mfma='v_mfma_f32_32x32x16_f16', cyc=32, per_block=8)),
        ('topknf', r'score_topk_refine2_kernel', 'score_topk_refine2_kernel<refine>, 480,189 x 17,770, k = 128',
         dict(users=480189, items=17770, k=128, mfma='v_mfma_f32_32x32x16_f16', cyc=32, per_block=8)),
        ('topk32', r'score_topk_kernel<', 'score_topk_kernel (fp32 MFMA), 69,878 x 10,380, k = 128',
         dict(users=69878, items=10380, k=128, mfma='v_mfma_f32_32x32x2_f32', cyc=64, per_block=64)),
        ('vbprd', r'vbpr_project_kernel', 'vbpr_project_kernel (fp32 MFMA), B = 256, d = 20,000, kh = 64', None),
        ('vbprd', r'vbpr_dense_kernel', 'vbpr_dense_kernel (fp32 MFMA), B = 256, d = 20,000, kh = 64', None))
res = {'source': 'rocprofv3 --kernel-trace --pmc <one counter per pass> around scripts/probe_topk.py / scripts/probe_vbpr.py (scripts/collect_mfma.sh)',
       'formula': 'mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (%d SIMDs x GRBM_GUI_ACTIVE / %d XCCs); the largest dispatches of a leg (the real passes, '
                  'not the 5-8 us no-op fallback launches) are averaged' % (SIMDS, XCCS), 'kernels': {}}
for leg, pat, name, shape in LEGS:
    out = {}
    vals = {}
    for C in ('SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_BUSY_CYCLES', 'GRBM_GUI_ACTIVE', 'SQ_INSTS_VALU_MFMA_MOPS_F16', 'SQ_INSTS_VALU_MFMA_MOPS_F32', 'SQ_WAVE_CYCLES'):
        g = per_kernel(leg, C, pat)
        if g is None:
            continue
        g = g.assign(dur=g.e - g.s)
        big = g[g.dur > 0.5 * g.dur.max()]                 # the real passes
        vals[C] = float(big.v.mean())
        out[C] = {'per_dispatch': vals[C], 'dispatches': int(len(big)), 'duration_us': float(big.dur.mean()) / 1e3}
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in vals and 'GRBM_GUI_ACTIVE' in vals:
        elapsed = vals['GRBM_GUI_ACTIVE'] / XCCS
        out['mfma_busy_frac'] = vals['SQ_VALU_MFMA_BUSY_CYCLES'] / (SIMDS * elapsed)
        out['clock_GHz_from_counters'] = elapsed / (out['GRBM_GUI_ACTIVE']['duration_us'] * 1e3)
    if shape and 'SQ_VALU_MFMA_BUSY_CYCLES' in vals:
        blocks = -(-shape['users'] // 32) * -(-shape['items'] // 32) * (shape['k'] // 128 if shape['k'] >= 128 else 1)
        n = blocks * shape['per_block']
        out['expected'] = {'mfma_instructions': n, 'instruction': shape['mfma'], 'cycles_each': shape['cyc'], 'busy_cycles': n * shape['cyc'],
                           'counter_over_expected': vals['SQ_VALU_MFMA_BUSY_CYCLES'] / (n * shape['cyc'])}
    if out:
        res['kernels'][name] = out
json.dump(res, open('profiles/%s_pmc_mfma.json' % tag, 'w'), indent=1)
print(json.dumps(res, indent=1))
