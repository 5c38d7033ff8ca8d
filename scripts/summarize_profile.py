"""Turn a gpurun_out/prof_<tag> directory (rocprofv3 trace + PMC passes of bench.py) into the tracked summaries
under profiles/: <round>_bench_kernel_stats.csv, <round>_kernel_summary.csv, <round>_pmc_traffic.json."""
import csv, json, os, re, sys
import pandas as pd
src, tag = sys.argv[1], sys.argv[2]
KERN = ('bpr_step_kernel', 'sample_plan_kernel', 'resolve_kernel', 'commit_kernel', 'score_topk_bf16_kernel', 'score_topk_kernel', 'merge_topk_kernel',
        'raw_rank_kernel', 'count_hits_rr_kernel',
        'build_mask_kernel', 'vbpr_sproject_kernel', 'vbpr_sdense_kernel', 'vbpr_project_kernel', 'vbpr_reduce_kernel', 'vbpr_occur_kernel', 'vbpr_rows_kernel', 'vbpr_dense_kernel',
        'calib_rowcopy_kernel')
def short(n):
    for k in KERN:
        if k in n:
            if k == 'bpr_step_kernel' and re.search(r'bpr_step_kernel<\d+, (?:true|false), \d+, true>', n):
                return 'tkr::bpr_step_kernel<SGD>'
            return 'tkr::' + k
    return None
import glob
def find(sub, name):
    hits = glob.glob(os.path.join(src, sub, '**', name), recursive=True)
    assert hits, (sub, name)
    return hits[0]
rows = list(csv.reader(open(find('trace', 'bench_kernel_stats.csv'))))
csv.writer(open('profiles/%s_bench_kernel_stats.csv' % tag, 'w')).writerows([rows[0]] + [[r[0][:110]] + r[1:] for r in rows[1:]])
t = pd.read_csv(find('trace', 'bench_kernel_trace.csv*'))
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
    d = pd.read_csv(find('%s_%s' % (tagdir, C), '%s_counter_collection.csv*' % tagdir[0]))
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
f = counter('bench', 'FETCH_SIZE', 'bpr_step_kernel|score_topk')
w = counter('bench', 'WRITE_SIZE', 'bpr_step_kernel|score_topk')
f = f.assign(k=f.Kernel_Name.map(short))
w = w.assign(k=w.Kernel_Name.map(short))
raw = {}
for (k, g), ff in f.groupby(['k', 'Grid_Size']):
    ww = w[(w.k == k) & (w.Grid_Size == g)]
    raw[(k, int(g))] = {'FETCH_SIZE_KB': float(ff.Counter_Value.mean()), 'WRITE_SIZE_KB': float(ww.Counter_Value.mean()), 'launches': int(len(ff)),
                        'hbm_bytes_per_launch_corrected': float(ff.Counter_Value.mean() * 1024 * fr + ww.Counter_Value.mean() * 1024 * fw)}
    print(k, g, raw[(k, int(g))])
def pick(kernel, which):
    grids = sorted(g for (k, g) in raw if k == kernel)
    return raw[(kernel, grids[0] if which == 'small' else grids[-1])] if grids else None
topk = 'tkr::score_topk_bf16_kernel' if any(k == 'tkr::score_topk_bf16_kernel' for k, _ in raw) else 'tkr::score_topk_kernel'
for key, val in (('bpr_step_B256', pick('tkr::bpr_step_kernel', 'small')), ('bpr_step_B8192', pick('tkr::bpr_step_kernel', 'large')),
                 ('bpr_step_sgd_B256', pick('tkr::bpr_step_kernel<SGD>', 'small')), ('bpr_step_sgd_B8192', pick('tkr::bpr_step_kernel<SGD>', 'large')),
                 ('score_topk_ml10m_k128', pick(topk, 'small')), ('score_topk_netflix_k128', pick(topk, 'large'))):
    if val is not None:
        res[key] = val
res['by_kernel_and_grid'] = {'%s grid %d' % k: v for k, v in raw.items()}
json.dump(res, open('profiles/%s_pmc_traffic.json' % tag, 'w'), indent=1)
if os.path.exists(os.path.join(src, 'bench_under_rocprof.json')):
    import shutil
    shutil.copy(os.path.join(src, 'bench_under_rocprof.json'), 'profiles/%s_bench_under_rocprof.json' % tag)
