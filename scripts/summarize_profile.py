"""Turn a gpurun_out/prof_<tag> directory (rocprofv3 trace + PMC passes of bench.py) into the tracked summaries
under profiles/: <round>_bench_kernel_stats.csv, <round>_kernel_summary.csv, <round>_pmc_traffic.json."""
import csv, json, os, re, sys
import pandas as pd
src, tag = sys.argv[1], sys.argv[2]
KERN = ('own_loss_kernel', 'step_loss_kernel', 'loss_slots_kernel', 'loss_slots_sum_kernel', 'bpr_own_kernel', 'bpr_flow_kernel', 'bpr_step_kernel', 'sample_plan_kernel', 'resolve_flow_wide_kernel', 'resolve_flow_kernel', 'score_topk_slab_kernel', 'resolve_kernel', 'commit_kernel', 'rollback_kernel',
        'mid_draw_kernel', 'mid_count_kernel', 'mid_build_wide_kernel', 'mid_build_kernel', 'mid_prefix_kernel', 'mid_resolve_kernel', 'mid_zero_kernel',
        'big_draw_kernel', 'big_flag_kernel', 'big_emit_kernel', 'big_count_kernel', 'big_fill_kernel', 'big_parity_kernel', 'big_record_kernel',
        'DeviceRadixSort', 'radix_sort', 'DeviceScan', 'scan', 'vbpr_pair_kernel', 'score_topk_refine2_kernel', 'topk_finish2_kernel', 'score_topk_wide_kernel', 'bpr_wide_kernel', 'score_topk_bf16_kernel', 'score_topk_kernel', 'merge_topk_kernel', 'topk_bounds_kernel',
        'raw_rank_kernel', 'count_hits_rr_kernel',
        'build_mask_kernel', 'vbpr_sproject_kernel', 'vbpr_sdense_kernel', 'vbpr_project_kernel', 'vbpr_reduce_kernel', 'vbpr_occur_kernel', 'vbpr_rows_kernel', 'vbpr_dense_kernel',
        'vbpr_tproject_kernel', 'vbpr_pairsum_kernel', 'vbpr_update_kernel', 'vbpr_colplan_kernel', 'topk_image_kernel',
        'sync_flow_pack_kernel', 'sync_flow_unpack_kernel', 'sync_flow_snapshot_kernel', 'calib_rowcopy_kernel')
def short(n):
    for k in KERN:
        if k in n:
            if k == 'bpr_step_kernel' and re.search(r'bpr_step_kernel<\d+, (?:true|false), \d+, true>', n):
                return 'tkr::bpr_step_kernel<SGD>'
            if k == 'score_topk_bf16_kernel':           # <KS, IdT, REFINE>: the bound-and-refine arithmetic is its own line
                return 'tkr::score_topk_bf16_kernel<refine>' if re.search(r'score_topk_bf16_kernel<\d+, [a-z ]+, true', n) else 'tkr::score_topk_bf16_kernel<bf16x3>'
            if k == 'score_topk_kernel':
                return 'tkr::score_topk_kernel (fp32 MFMA; ~5-8 us calls: the no-op fallback pass behind a refine launch)'
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
    d = pd.read_csv(find('%s_%s' % (tagdir, C), '%s_counter_collection.csv*' % ('c' if tagdir == 'calib' else 'b')))
    return d[d.Kernel_Name.str.contains(pat)]
known_r, known_w = 541065216, 536870912
cf = counter('calib', 'FETCH_SIZE', 'calib_rowcopy').Counter_Value.mean()
cw = counter('calib', 'WRITE_SIZE', 'calib_rowcopy').Counter_Value.mean()
fr, fw = known_r / (cf * 1024), known_w / (cw * 1024)
res = {'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, kernel trace only) around `python bench.py --no-cpu-baseline '
                 '--no-extras --steps 2048 --warmup 256` (headline: the persistent kernel; with the epoch_mode leg of the same process 11,718 batches of 256) and `... --batch-size 8192 --steps 256 '
                 '--warmup 128` (one launch per batch); corrected by the factors measured with scripts/pmc_calibrate.py (known-byte gather copy); '
                 'bytes = KB * 1024 * factor',
       'calibration': {'known_read_bytes': known_r, 'known_write_bytes': known_w, 'FETCH_SIZE_KB': cf, 'WRITE_SIZE_KB': cw,
                       'read_factor': fr, 'write_factor': fw}}
def total(tagdir, pat):
    f, w = counter(tagdir, 'FETCH_SIZE', pat), counter(tagdir, 'WRITE_SIZE', pat)
    return float(f.Counter_Value.sum()) * 1024 * fr, float(w.Counter_Value.sum()) * 1024 * fw, len(f)
rd, wr, n = total('headline', 'bpr_own_kernel|bpr_flow_kernel')      # the persistent step: K2o where the item table fits the CUs' LDS, else K2f
batches = 2048 + 256
batches += (3906 - batches % 3906) % 3906 + 2 * 3906          # bench.py's epoch_mode leg of the same process: to the epoch boundary, then two reference epochs
res['bpr_flow_B256'] = {'launches': n, 'batches': batches, 'hbm_read_bytes_per_batch': rd / batches, 'hbm_write_bytes_per_batch': wr / batches,
                        'hbm_bytes_per_launch_corrected': (rd + wr) / batches,
                        'note': 'per BATCH (a launch of the persistent kernel covers up to 512 batches); algorithmic 1,587,200 B; the granule '
                                'tables move 8 bytes per fp32 (value + version tag)'}
res['bpr_step_B256'] = res['bpr_own_B256'] = res['bpr_flow_B256']          # the keys bench.py looks up for the headline
rd, wr, n = total('b8192', 'bpr_step_kernel')
res['bpr_step_B8192'] = {'launches': n, 'hbm_bytes_per_launch_corrected': (rd + wr) / max(n, 1), 'hbm_read_bytes_per_launch': rd / max(n, 1),
                         'hbm_write_bytes_per_launch': wr / max(n, 1)}
# round 3: more legs (each optional: a pass that did not run is skipped)
def leg(tagdir, pat, key, per, note):
    try:
        rd, wr, n = total(tagdir, pat)
    except AssertionError:
        return
    if n:
        res[key] = {'launches': n, 'hbm_read_bytes_per_launch': rd / n, 'hbm_write_bytes_per_launch': wr / n,
                    'hbm_bytes_per_launch_corrected': (rd + wr) / n, 'per': per, 'note': note}
leg('topk', r'score_topk_refine2_kernel|topk_finish2_kernel', 'score_topk_ml10m_k128', 'pass over 69,878 users x 10,380 items, k = 128, top-30',
    'the tile kernel + the finish kernel of bound-and-refine (the bounds / image / merge launches are a few hundred KB); launches = 2 per pass: '
    'hbm_bytes_per_launch_corrected x 2 = bytes per pass')
leg('topknf', r'score_topk_refine2_kernel|topk_finish2_kernel', 'score_topk_netflix_k128', 'pass over 480,189 users x 17,770 items, k = 128, top-30',
    'as above: two launches per pass')
leg('b65536', 'bpr_step_kernel', 'bpr_step_B65536', 'batch of 65,536 triplets', 'algorithmic 406 MB: no credit for the duplicates of 131,072 item draws over 10,380 items')
for pat, key in (('vbpr_tproject_kernel', 'vbpr_tproject_B256'), ('vbpr_pairsum_kernel', 'vbpr_pairsum_B256'), ('vbpr_update_kernel', 'vbpr_update_B256')):
    leg('vbpr', pat, key, 'batch of 256 triplets, d = 20,000, ~100 nonzeros per feature row', '')
for key in ('bpr_flow_B256', 'bpr_step_B8192', 'score_topk_ml10m_k128', 'score_topk_netflix_k128', 'bpr_step_B65536', 'vbpr_tproject_B256', 'vbpr_update_B256'):
    print(key, res.get(key))
json.dump(res, open('profiles/%s_pmc_traffic.json' % tag, 'w'), indent=1)
if os.path.exists(os.path.join(src, 'bench_under_rocprof.json')):
    import shutil
    shutil.copy(os.path.join(src, 'bench_under_rocprof.json'), 'profiles/%s_bench_under_rocprof.json' % tag)
# the driver's exact command: which HIP calls and kernels the process made (the timed 20 steps must be launches only)
try:
    hs = pd.read_csv(find('driver', 'd_hip_api_stats.csv'))
    hs.to_csv('profiles/%s_driver_cmd_hip_api_stats.csv' % tag, index=False)
    ks = pd.read_csv(find('driver', 'd_kernel_stats.csv'))
    ks['Name'] = ks['Name'].str.slice(0, 110)
    ks.to_csv('profiles/%s_driver_cmd_kernel_stats.csv' % tag, index=False)
    shutil.copy(os.path.join(src, 'driver_cmd.json'), 'profiles/%s_driver_cmd.json' % tag)
    ht = pd.read_csv(find('driver', 'd_hip_api_trace.csv*'))
    kt = pd.read_csv(find('driver', 'd_kernel_trace.csv*'))
    flow = kt[kt.Kernel_Name.str.contains('bpr_own_kernel|bpr_flow_kernel')].sort_values('Start_Timestamp')
    # the warm-up (5 single-batch calls since round 4's short-call work), then the timed 20 batches; the epoch_mode leg follows
    n_warm = 5 if (flow.iloc[:5].Grid_Size_X == flow.iloc[0].Grid_Size_X).all() and \
        (flow.iloc[:5].End_Timestamp - flow.iloc[:5].Start_Timestamp).max() < 0.7 * (flow.iloc[5].End_Timestamp - flow.iloc[5].Start_Timestamp) else 1
    last = flow.iloc[n_warm]
    prev_end = flow.iloc[n_warm - 1].End_Timestamp
    win = ht[(ht.Start_Timestamp > prev_end) & (ht.Start_Timestamp < last.End_Timestamp)]
    calls = win.Function.value_counts().to_dict()
    json.dump({'window': 'HIP API calls between the end of the warm-up launch and the end of the timed launch of `bench.py --steps 20 --warmup 5`',
               'calls': calls, 'timed_kernel_us': float(last.End_Timestamp - last.Start_Timestamp) / 1e3},
              open('profiles/%s_driver_cmd_timed_region.json' % tag, 'w'), indent=1)
    print('timed region HIP calls:', calls)
except Exception as e:       # noqa
    print('driver trace not summarised:', repr(e))
