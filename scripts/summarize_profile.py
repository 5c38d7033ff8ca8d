"""Turn a gpurun_out/prof_<tag> directory (rocprofv3 trace + PMC passes of bench.py) into the tracked summaries
under profiles/: <round>_bench_kernel_stats.csv, <round>_kernel_summary.csv, <round>_pmc_traffic.json."""
import csv, json, os, sys
import pandas as pd
src, tag = sys.argv[1], sys.argv[2]
KERN = ('bpr_step_kernel', 'sample_plan_kernel', 'resolve_kernel', 'commit_kernel', 'score_topk_kernel', 'merge_topk_kernel',
        'build_mask_kernel', 'vbpr_project_kernel', 'vbpr_reduce_kernel', 'vbpr_occur_kernel', 'vbpr_rows_kernel', 'vbpr_dense_kernel',
        'calib_rowcopy_kernel')
def short(n):
    for k in KERN:
        if k in n:
            return 'tkr::' + k
    return None
rows = list(csv.reader(open(os.path.join(src, 'trace', 'bench_kernel_stats.csv'))))
csv.writer(open('profiles/%s_bench_kernel_stats.csv' % tag, 'w')).writerows([rows[0]] + [[r[0][:110]] + r[1:] for r in rows[1:]])
t = pd.read_csv(os.path.join(src, 'trace', 'bench_kernel_trace.csv'))
t['dur'] = t.End_Timestamp - t.Start_Timestamp
t['k'] = t.Kernel_Name.map(short)
out = []
for (k, g), d in t[t.k.notna()].groupby(['k', 'Grid_Size_X']):
    out.append((k, g, d.Workgroup_Size_X.iloc[0], len(d), d.dur.mean() / 1e3, d.dur.median() / 1e3, d.dur.min() / 1e3, d.dur.max() / 1e3,
                d.VGPR_Count.iloc[0], d.SGPR_Count.iloc[0], d.LDS_Block_Size.iloc[0]))
df = pd.DataFrame(out, columns=['kernel', 'grid_threads', 'wg', 'calls', 'avg_us', 'median_us', 'min_us', 'max_us', 'vgpr', 'sgpr', 'lds'])
df.to_csv('profiles/%s_kernel_summary.csv' % tag, index=False)
print(df.to_string())
def counter(tagdir, C, pat):
    d = pd.read_csv(os.path.join(src, '%s_%s' % (tagdir, C), '%s_counter_collection.csv' % tagdir[0]))
    return d[d.Kernel_Name.str.contains(pat)]
known_r, known_w = 541065216, 536870912
cf = counter('calib', 'FETCH_SIZE', 'calib_rowcopy').Counter_Value.mean()
cw = counter('calib', 'WRITE_SIZE', 'calib_rowcopy').Counter_Value.mean()
fr, fw = known_r / (cf * 1024), known_w / (cw * 1024)
res = {'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) around `python bench.py --no-cpu-baseline --steps 2048 '
                 '--warmup 256`; corrected by the factors measured with scripts/pmc_calibrate.py (known-byte gather copy, same access '
                 'pattern); bytes = KB * 1024 * factor',
       'calibration': {'known_read_bytes': known_r, 'known_write_bytes': known_w, 'FETCH_SIZE_KB': cf, 'WRITE_SIZE_KB': cw,
                       'read_factor': fr, 'write_factor': fw}}
f = counter('bench', 'FETCH_SIZE', 'bpr_step_kernel|score_topk_kernel')
w = counter('bench', 'WRITE_SIZE', 'bpr_step_kernel|score_topk_kernel')
for name, pat in (('bpr_step', 'bpr_step_kernel'), ('score_topk', 'score_topk_kernel')):
    for g, ff in f[f.Kernel_Name.str.contains(pat)].groupby('Grid_Size'):
        ww = w[(w.Kernel_Name.str.contains(pat)) & (w.Grid_Size == g)]
        key = '%s_grid%d' % (name, g)
        res[key] = {'FETCH_SIZE_KB': float(ff.Counter_Value.mean()), 'WRITE_SIZE_KB': float(ww.Counter_Value.mean()), 'launches': int(len(ff)),
                    'hbm_bytes_per_launch_corrected': float(ff.Counter_Value.mean() * 1024 * fr + ww.Counter_Value.mean() * 1024 * fw)}
        print(key, res[key])
json.dump(res, open('profiles/%s_pmc_traffic_raw.json' % tag, 'w'), indent=1)
