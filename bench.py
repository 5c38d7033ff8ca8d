#!/usr/bin/env python
"""bench.py -- BPR training triplets/s on synthetic MovieLens-10M-shaped data (BASELINE.json
configs[1]: ~70k users x ~10k items, k = 128), one process per GPU.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one mini-batch: (u,i,j) draw + batch plan (K1) and the
loss/gradient/RMSProp update (K2), reference semantics (single/bpr.py:136-147), default batch
256 like the reference's train.py.  Inputs (training CSR, model tables) are resident in HBM
before the timed region.  N > 1: users are sharded, item tables replicated, one RCCL all-reduce
of the item-side state per epoch (every (limit//B)//N steps); weak scaling (K steps per rank).

Prints ONE JSON line (rank 0).  Extra objects beside the contract fields:
  roofline        dominant kernel (K2) against the HBM roofline: algorithmic bytes per launch
                  (B x (48k+56) B) / average launch duration from HIP events;
                  `traffic` = HBM bytes per batch of the headline step measured in THIS run: two rocprofv3 counter
                  passes (FETCH_SIZE, WRITE_SIZE; kernel trace only) of scripts/pmc_leg.py in child processes
                  (live_traffic(); ~10 s; --no-live-traffic / --no-extras skip them; null with `traffic_how`
                  saying why where rocprofv3 cannot run); `traffic_from_profile` replays the committed
                  profiles/rNN_pmc_traffic.json (the other legs: only that)
  cpu_baseline    the numpy oracle with the reference's cost structure (per-element legacy
                  np.random sampler + numpy step), bounded sample, rank 0, N = 1 only
  throughput_mode the same path at batch_size 8192 (a legal train() argument; NOT the headline)
  sgd_mode / sgd_throughput_mode  the legacy plain-SGD optimiser (old/methods/bpr.py:57-61) at batch 256 / 8192
  bpr_netflix_shape  the headline path at BASELINE.json configs[3]'s shape (480,189 x 17,770) on one GPU
  topk            the other half of BASELINE.json's metric: full-catalogue top-30 scored users/s (K4)
                  with its own MFMA roofline (fp16 dense peak for the default bound-and-refine arithmetic; TKR_TOPK_MATH=bf16x3|fp32 the others) and cpu_baseline
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'top-k-rec_amd')]

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
MFMA_F32_PEAK_TF = 157.3       # MI355X_MICROARCH.md: fp32-input MFMA, dense
MFMA_BF16_PEAK_TF = 2500.0     # MI355X_MICROARCH.md: bf16 MFMA, dense (no sparsity)


MFMA_F16_PEAK_TF = 2500.0      # MI355X_MICROARCH.md: fp16 MFMA, dense (same rate as bf16)


def topk_math(k):
    """the score arithmetic K4 runs a width-k problem in (include/tkr.h tkr_topk_set_math)"""
    mode = os.environ.get('TKR_TOPK_MATH', 'refine')
    return 'fp32' if (k > 128 or mode == 'fp32') else ('bf16x3' if mode == 'bf16x3' else 'refine')


def topk_roofline(tf, k):
    """K4 against the pipe it runs on.  `achieved` is always ALGORITHMIC fp32 flops (2*k*n_items per user) per
    second.  refine (default, k <= 128): ONE fp16 product per element on the dense matrix pipe proposes the candidates (the
    fp32 rescoring of ~33 of them per user is 0.2 % of the flops), ceiling = the fp16 dense peak; bf16x3: six bf16 partial
    products per element, ceiling = bf16 dense peak / 6; fp32 (and k > 128): the fp32 MFMA peak."""
    math = topk_math(k)
    peak = {'refine': MFMA_F16_PEAK_TF, 'bf16x3': MFMA_BF16_PEAK_TF / 6.0, 'fp32': MFMA_F32_PEAK_TF}[math]
    what = {'refine': 'bound-and-refine: 1 x v_mfma_f32_32x32x16_f16 per 16 k (scaled fp16, rigorous margin) + exact fp32 fma-chain '
                      'rescoring of the candidates; results = the fp32 arithmetic bit for bit',
            'bf16x3': 'bf16x3 split, 6 x v_mfma_f32_32x32x16_bf16 per 16 k, fp32 accumulate: executed bf16 rate %.0f TFLOP/s of %.0f '
                      'dense' % (6 * tf, MFMA_BF16_PEAK_TF),
            'fp32': 'v_mfma_f32_32x32x2_f32'}[math]
    kernel = {'refine': 'tkr::score_topk_refine2_kernel (+ tkr::topk_finish2_kernel: exact rescoring and order, once per row)', 'bf16x3': 'tkr::score_topk_bf16_kernel',
              'fp32': 'tkr::score_topk_wide_kernel' if k > 768 else 'tkr::score_topk_slab_kernel' if k > 256 else 'tkr::score_topk_kernel'}[math]
    return {'kernel': kernel, 'bound': 'mfma', 'achieved': tf,
            'peak': peak, 'unit': 'TFLOP/s', 'frac': tf / peak, 'arithmetic': what, 'vs_fp32_mfma_peak': tf / MFMA_F32_PEAK_TF}


def topk_other_arithmetics(run, k):
    """ms per pass of the same launch under the other two score arithmetics (all on the bench line)"""
    import tkr_hip
    out = {}
    if k > 128:
        return out
    try:
        for mode in ('refine', 'bf16x3', 'fp32'):
            if mode == topk_math(k) or (mode == 'bf16x3' and not tkr_hip.lab()):      # (bf16x3: measured and dropped, `make LAB=1` only)
                continue
            tkr_hip.set_topk_math(mode)
            run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run(); run()
            e1.record()
            torch.cuda.synchronize()
            out[mode] = e0.elapsed_time(e1) / 2
    finally:
        tkr_hip.set_topk_math(topk_math(k))
    return out


def pmc_traffic(key):
    """HBM bytes per launch measured with rocprofv3 PMC counters in an earlier profiled run of this
    same command (profiles/rNN_pmc_traffic.json, newest round); None if that run does not exist"""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_traffic.json')))
    if not files:
        return None
    try:
        return json.load(open(files[-1])).get(key, {}).get('hbm_bytes_per_launch_corrected')
    except (OSError, ValueError):
        return None


ROUND = 'r06'                  # the round this bench.py belongs to: counter files of another round are replayed with "stale": true


def pmc_k4(shape):
    """counter summary of the K4 tile kernel at the ML-10M ('ml') or Netflix ('nf') shape from the newest profiles/rNN_pmc_k4.json
    (scripts/collect_k4_counters.sh + scripts/summarize_k4_counters.py): matrix-pipe busy fraction, cycles and instructions per tile-wave"""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_pmc_k4.json')))
    if not files:
        return None
    try:
        v = json.load(open(files[-1]))['shapes'][shape]
        src = os.path.basename(files[-1])
        return {'mfma_busy_frac': v['mfma_busy_frac'], 'valu_issue_frac_of_simd_time': v['valu_issue_frac_of_simd_time'],
                'resident_waves_per_simd': v['resident_waves_per_simd'], 'per_tile_wave': v['per_tile_wave'],
                'fractions_of_wave_cycles': v['fractions_of_wave_cycles'], 'source': src, 'stale': not src.startswith(ROUND)}
    except (OSError, ValueError, KeyError):
        return None


def pmc_mfma(match):
    """matrix-pipe busy fraction (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x elapsed shader cycles)) of the kernel whose name contains
    `match`, measured with rocprofv3 PMC passes in an earlier run of scripts/collect_mfma.sh (profiles/rNN_pmc_mfma.json, newest round)"""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_mfma.json')))
    if not files:
        return None
    try:
        for name, v in json.load(open(files[-1]))['kernels'].items():
            if match in name:
                return {'kernel': name, 'mfma_busy_frac': v.get('mfma_busy_frac'),
                        'SQ_VALU_MFMA_BUSY_CYCLES': v['SQ_VALU_MFMA_BUSY_CYCLES']['per_dispatch'],
                        'counter_over_expected_cycles': (v.get('expected') or {}).get('counter_over_expected'), 'source': os.path.basename(files[-1]),
                        'stale': not os.path.basename(files[-1]).startswith(ROUND)}       # a counter file of an earlier round: the kernel may have changed since
    except (OSError, ValueError, KeyError):
        pass
    return None


def live_traffic(k, B, shape, batches=4096):
    """HBM bytes per BATCH of the headline step, measured NOW: two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE: separate passes, kernel
    trace only, as MI355X_MICROARCH.md prescribes) of scripts/pmc_leg.py in child processes, bytes = KB x 1024 x the factors of the
    known-byte calibration copy (profiles/rNN_pmc_traffic.json: FETCH x 1.98, WRITE x 1.00 on gfx950).  -> (bytes per batch | None, note)"""
    import csv, glob, shutil, subprocess, tempfile
    if shutil.which('rocprofv3') is None:
        return None, 'rocprofv3 not on PATH'
    if any(v.startswith(('ROCPROF', 'ROCP_', 'ROCTX')) for v in os.environ):
        return None, 'this process runs under a profiler already'
    fr, fw = 1.9842944385544612, 1.0
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_traffic.json')))
    if files:
        try:
            cal = json.load(open(files[-1]))['calibration']
            fr, fw = float(cal['read_factor']), float(cal['write_factor'])
        except (OSError, ValueError, KeyError):
            pass
    got = {}
    for C in ('FETCH_SIZE', 'WRITE_SIZE'):
        d = tempfile.mkdtemp(prefix='tkr_pmc_', dir='/tmp')
        try:
            p = subprocess.run(['rocprofv3', '--kernel-trace', '--pmc', C, '--output-format', 'csv', '-d', d, '-o', 'b', '--', sys.executable,
                                os.path.join(ROOT, 'scripts', 'pmc_leg.py'), str(k), str(B), shape, str(batches)],
                               cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=240)
            info = json.loads(p.stdout.decode().strip().splitlines()[-1])
            hits = glob.glob(os.path.join(d, '**', 'b_counter_collection.csv'), recursive=True)
            name = info['kernel'].split('::')[-1]
            kb = 0.0
            for row in csv.DictReader(open(hits[0])):
                if name in row['Kernel_Name'] and row.get('Counter_Name', C) == C:
                    kb += float(row['Counter_Value'])
            got[C] = (kb * 1024.0, info['batches'])
        except Exception as e:          # a counter pass is evidence, never a reason to lose the bench line
            return None, '%s pass failed: %s' % (C, type(e).__name__)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    rd, nb = got['FETCH_SIZE']
    wr, _ = got['WRITE_SIZE']
    if rd <= 0 or wr <= 0:
        return None, 'no counter rows for the step kernel'
    return (rd * fr + wr * fw) / nb, ('rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, two child passes of scripts/pmc_leg.py (%d batches each), per BATCH '
                                    '(a launch of the persistent kernel covers up to 512); read x %.3f, write x %.3f (calibration copy)' % (nb, fr, fw))


def algorithmic_bytes_per_triplet(k):
    return 48 * k + 56          # 3 rows x (param+slot) x (read+write) + biases + ids


def build_problem(shape, k, rank, world, device, seed=42):
    import synth
    from single import _engine
    spec = dict(synth.ML10M if shape == 'ml10m' else synth.NETFLIX)
    r = synth.make_ratings(seed=seed, **spec)
    row_ptr, pos, _, tr_users = synth.positives_csr(r)
    n_users, n_items = r['n_users'], r['n_in'] + r['n_out']
    import dist as tdist
    hp = dict(lu=2.5e-3, li=2.5e-3, lj=2.5e-4, lb=0.0, lr=1e-4, mode='l2')       # single/bpr.py:20 defaults
    if world == 1:
        csr = _engine.TrainingCSR.from_arrays(row_ptr, pos, np.asarray(tr_users, dtype=np.int32), device)
        eng = _engine.BprEngine(n_users, n_items, k, hp, device, seed=1234)
    else:           # as BPR.train: the rank allocates only the rows of its users, the item tables are replicas (same seed)
        mine = np.asarray(tdist.shard_users(tr_users, rank, world), dtype=np.int64)
        csr = _engine.TrainingCSR.shard(row_ptr, pos, mine, device)
        eng = _engine.BprEngine(len(mine), n_items, k, hp, device, seed=1234, user_seed=99991 * (rank + 1))
        eng.ranks_on_device = tdist.ranks_sharing_device(device)       # ranks packed on one GPU split its CUs between their K2o launches (as BPR.train)
    return r, csr, eng, int(row_ptr[-1])


def _chunk_crossings(eng, steps, B):
    """upper bound of the extra step calls a run of `steps` batches makes because it crosses plan chunks"""
    return steps // eng._cap(B) + 1


# Every step also produces its batch's loss, as the reference's does (`_, loss = sess.run([solver, obj])`, single/bpr.py:141,
# vbpr.py:114) -- since round 4 (before: the step without it; TKR_BENCH_LOSS=0 brings that back for comparison)
WANT_LOSS = os.environ.get('TKR_BENCH_LOSS', '1') != '0'


class Loop:
    """The loop of single/bpr.py:136-147 as BPR.train runs it on this rank: batches in stream order and, at N > 1, the exchange
    of the item-side tables at its REAL cadence -- every `sync_every` batches, counted across run() calls (round 2 wrapped
    every call in begin()/end(): a 20-step timed call then paid one whole exchange that belongs to an epoch of 488 batches)."""

    def __init__(self, eng, csr, B, sync_every, world, names=None):
        import dist as tdist
        self.eng, self.csr, self.B, self.sync_every, self.world = eng, csr, B, sync_every, world
        eng.prepare(B)                   # the table layout of this batch size BEFORE the exchange binds to the tables (as BPR.train does)
        self.isync = tdist.ItemSync(eng, names) if world > 1 else None
        self.since = 0                   # batches since the last exchange
        self.exchanges = 0

    def run(self, n):
        done = 0
        while done < n:
            if self.isync is not None and self.since == 0:
                self.isync.begin()
            m = min(n - done, self.sync_every - self.since)
            ends_epoch = self.isync is not None and self.since + m == self.sync_every        # BPR.train: the exchange follows, then another epoch
            self.eng.run_batches(self.csr, m, self.B, want_loss=WANT_LOSS, then_exchange=self.sync_every if ends_epoch else 0)
            done += m
            self.since += m
            if self.since == self.sync_every:
                self.since = 0
                if self.isync is not None:
                    self.isync.end()
                    self.exchanges += 1


def _fence(world):
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()


def timed_run(eng, csr, B, steps, warmup, sync_every, world, names=None, loop=None, per_launch_events=True):
    """warmup untimed, then exactly `steps` batches between barriers; returns (wall_s, step_kernel_ms, loop).
    What the timed region contains: for each batch its (u, i, j) draw + plan (K1) AND its step -- settle() drops
    whatever an earlier call planned but did not run, and run_batches plans exactly what it is asked to run.
    step_kernel_ms: HIP events on the training stream.  ``per_launch_events``: one pair around every step launch (the steady
    state: launches of 512 batches); else ONE pair around the whole timed region -- the headline of a short run (`--steps 20`
    is ONE launch that plans and steps: a pair of event records inside the call is two more packets on the queue and ~10 us of
    host time in front of a ~90 us launch), which then also counts the host's way to the launch: an upper bound of the launch."""
    loop = loop or Loop(eng, csr, B, sync_every, world, names)
    singles = min(warmup, 8)
    if per_launch_events:
        eng.reserve_events(-(-(steps + warmup) // sync_every) + _chunk_crossings(eng, steps + warmup, B) + singles + 2)   # created now, not between the timed launches
    # The warm-up goes down the SAME host path as the timed call and its first batches as calls of their own: the first two or
    # three passes of the runtime through a launch path (kernel arguments, signals) cost tens of microseconds each, which a
    # 20-step timed call (~130 us) would otherwise carry.
    eng.step_events = [] if per_launch_events else None
    for w in range(singles):
        if w == singles - 1 and warmup == singles:
            eng.settle()         # the status word of the warm-up so far (a copy to the host) is looked at BEFORE the last warm-up launch:
                                 # the first launch behind a device-to-host copy costs the host ~4 us more (scripts/probe_short_host.py
                                 # PROBE_SETTLE: 112 -> 116 us per timed call), and in training no copy sits in front of every call
        loop.run(1)
        _fence(world)            # ... each onto an IDLE queue, as the timed call goes out (behind the fence below): the runtime's first
                                 # launches after an idle queue are slower than back-to-back ones (measured: timed call 148 -> 132 us)
    if warmup > singles:
        loop.run(warmup - singles)
    eng.step_events = None
    # nothing planned ahead: the timed batches sample and plan themselves (what a warm-up call planned beyond its own batches is
    # rolled back; the status check of everything that ran comes behind the timed region: eng.check() below)
    eng.settle(check=not (0 < singles == warmup))
    region = _recorded_pair() if not per_launch_events else None
    _fence(world)
    eng.step_events = [] if per_launch_events else None
    before = loop.exchanges
    if region is not None:
        region[0].record()
    t0 = time.perf_counter()
    loop.run(steps)
    t_host = time.perf_counter()
    if region is not None:
        region[1].record()
    _fence(world)
    wall = time.perf_counter() - t0
    if os.environ.get('TKR_BENCH_TRACE') == '1':
        print('timed region: host returned at %.1f us, fence at %.1f us' % ((t_host - t0) * 1e6, wall * 1e6), file=sys.stderr, flush=True)
    if region is not None:
        step_ms = region[0].elapsed_time(region[1])
        loop.last_step_events = []
    else:
        step_ms = sum(a.elapsed_time(b) for a, b, _ in eng.step_events)
        assert sum(n for _, _, n in eng.step_events) == steps
        loop.last_step_events = eng.step_events
    eng.step_events = None
    eng.check()
    loop.timed_exchanges = loop.exchanges - before
    return wall, step_ms, loop


def _recorded_pair():
    """two timing events that already exist on the device (a torch event is created by its first record())"""
    pair = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    for e in pair:
        e.record()
    return pair


def step_kernel(eng, B):
    """(kernel symbol, C-ABI entry, profile key) of the step that runs batch size B on this engine"""
    if eng.layout != 'flow':
        return 'tkr::bpr_step_kernel', 'K2 tkr_bpr_run', 'bpr_step_B%d' % B
    if eng._plan_owners(B):
        return 'tkr::bpr_own_kernel', 'K2o tkr_bpr_own_run', 'bpr_own_B%d' % B
    return 'tkr::bpr_flow_kernel', 'K2f tkr_bpr_flow_run', 'bpr_flow_B%d' % B


def max_over_ranks(x, device, world):
    if world == 1:
        return x
    t = torch.tensor([x], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def epoch_mode(eng, csr, B, k, world, device, loop, epochs=2):
    """Whole reference epochs: `epochs` x sync_every batches per rank (sync_every = (epoch_sample_limit // B) // N, what
    BPR.train gives a rank per epoch), each followed by its exchange at N > 1.  At N = 1 this is the steady state of the
    headline configuration (2 x 3906 = 7812 batches, K1 of every batch inside the timed region)."""
    per = loop.sync_every
    loop.run((per - loop.since) % per)           # to an epoch boundary
    if loop.isync is not None:
        loop.isync.timing = []
    wall, step_ms, _ = timed_run(eng, csr, B, epochs * per, 0, per, world, loop=loop)
    wall = max_over_ranks(wall, device, world)
    us = step_ms * 1e3 / (epochs * per)
    gbs = B * algorithmic_bytes_per_triplet(k) / (us * 1e-6) / 1e9
    out = {'epochs': epochs, 'batches_per_rank_per_epoch': per, 'steps': epochs * per, 'value': world * epochs * per * B / wall,
           'unit': 'triplets/s', 'ms_per_step': wall * 1e3 / (epochs * per), 'ms_per_epoch': wall * 1e3 / epochs,
           'timed_region': 'per batch: K1 (draw + plan) and the step; per epoch: the exchange of the item tables' if world > 1
                           else 'per batch: K1 (draw + plan, side stream behind the previous chunk) and the step',
           'roofline': {'kernel': step_kernel(eng, B)[0], 'bound': 'hbm', 'achieved': gbs,
                        'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': gbs / HBM_PEAK_GBS, 'launch_us': us,
                        'algorithmic_bytes_per_launch': B * algorithmic_bytes_per_triplet(k), 'traffic': None,
                        'traffic_from_profile': pmc_traffic(step_kernel(eng, B)[2]) if (k == 128 and eng.layout == 'flow') else None}}
    if loop.isync is not None:
        t = loop.isync.timing
        loop.isync.timing = None
        if t:
            parts = [(a.elapsed_time(b), b.elapsed_time(c), c.elapsed_time(d)) for a, b, c, d in t]
            n = len(parts)
            out['exchanges'] = n
            out['exchange_us'] = {'pack': 1e3 * sum(p[0] for p in parts) / n, 'collective': 1e3 * sum(p[1] for p in parts) / n,
                                  'unpack': 1e3 * sum(p[2] for p in parts) / n,
                                  'note': 'HIP events on the training stream around tkr_sync_*pack / all_reduce / tkr_sync_*unpack, mean per exchange, rank 0'}
            # what an epoch boundary exposes besides the exchange itself: from the end of the unpack to the first step launch of the
            # next epoch (K1 of its first chunk when it could not be planned ahead: the item counters restart at zero) -- VERDICT r3 #2b
            gaps, done, ev = [], 0, loop.last_step_events
            starts = {}
            for a, _, m in ev:
                if done % per == 0:
                    starts[done // per] = a
                done += m
            for e, marks in enumerate(t):
                if e + 1 in starts:
                    gaps.append(marks[3].elapsed_time(starts[e + 1]) * 1e3)
            if gaps:
                out['exchange_us']['exposed_after_exchange'] = sum(gaps) / len(gaps)
                out['exchange_us']['exposed_note'] = ('device time between the end of the unpack and the first step launch of the next epoch, mean of %d; the first '
                                                      'chunk of that epoch is planned AHEAD of the exchange (PlanMixin._next_chunk then_exchange: zeroed shadow of the '
                                                      'item counters, side stream, behind the last steps of the epoch before)' % len(gaps))
            out['ms_per_epoch_minus_collective'] = out['ms_per_epoch'] - out['exchange_us']['collective'] * 1e-3
            out['batches_x_launch_us_ms'] = per * us * 1e-3
    return out


def cpu_baseline(r, k, B, budget_s=12.0):
    """the oracle, shaped like the reference: legacy per-element sampler + numpy step, 1 core"""
    from oracle import ref_np as R
    keep = r['tr_l'] == 1
    users, items = r['tr_u'][keep], r['tr_i'][keep]
    cuts = np.flatnonzero(np.r_[True, users[1:] != users[:-1], True])
    tr_data = {int(users[a]): items[a:b].tolist() for a, b in zip(cuts[:-1], cuts[1:])}
    tr_users = list(tr_data.keys())
    n_users, n_items = r['n_users'], r['n_in'] + r['n_out']
    st = R.init_bpr_state(n_users, n_items, k, np.random.Generator(np.random.PCG64(0)))
    hp = dict(lu=2.5e-3, li=2.5e-3, lj=2.5e-4, lb=0.0, lr=1e-4, mode='l2')
    np.random.seed(0)
    gen = R.legacy_uniform_user_sampler(tr_users, tr_data, n_items, B)
    t0 = time.perf_counter()
    nb = 0
    while time.perf_counter() - t0 < budget_s:
        ub, ib, jb = next(gen)
        R.bpr_step(st, ub, ib, jb, hp)
        nb += 1
    dt = time.perf_counter() - t0
    return dict(value=nb * B / dt, unit='triplets/s', cores=1, host_cores=os.cpu_count(), kind='port',
                sample='%d batches of %d triplets (%.1f s): oracle/ref_np legacy np.random sampler + numpy step, '
                       'ML-10M shape k=%d' % (nb, B, dt, k))


def topk_problem(r, k, device, rank, world, seed=7):
    """scoring inputs: N(0,0.01) factors rounded like the '%f' export, every user's train history as the
    rated mask, all items as candidates (full catalogue); users block-sharded over ranks."""
    import synth
    import tkr_hip
    n_users, n_items = r['n_users'], r['n_in'] + r['n_out']
    lo, hi = rank * n_users // world, (rank + 1) * n_users // world
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    U = (torch.randn((n_users, k), device=device, generator=g) * 0.01 * 1e6).round() / 1e6
    V = (torch.randn((n_items, k), device=device, generator=g) * 0.01 * 1e6).round() / 1e6
    ptr, cols = synth.rated_csr(r)
    sub_ptr = torch.from_numpy(ptr[lo:hi + 1] - ptr[lo]).to(device)
    sub_cols = torch.from_numpy(cols[ptr[lo]:ptr[hi]]).to(device)
    mask, pitch = tkr_hip.build_rated_mask(sub_ptr, sub_cols, hi - lo, n_items)
    return U[lo:hi].contiguous(), V, mask, pitch


def topk_bench(r, k, device, rank, world, K=30, reps=5):
    import tkr_hip
    U, V, mask, pitch = topk_problem(r, k, device, rank, world)
    tkr_hip.score_topk(U, V, K, mask=mask, mask_pitch=pitch)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(reps):
        tkr_hip.score_topk(U, V, K, mask=mask, mask_pitch=pitch)
    e1.record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    wall = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([wall], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    launch_ms = e0.elapsed_time(e1) / reps
    others = topk_other_arithmetics(lambda: tkr_hip.score_topk(U, V, K, mask=mask, mask_pitch=pitch), k) if world == 1 else {}
    n_items = V.shape[0]
    flops = 2.0 * k * n_items * U.shape[0]
    tf = flops / (launch_ms * 1e-3) / 1e12
    return {'metric': 'full-catalogue top-%d scored users/sec' % K, 'value': r['n_users'] * reps / wall, 'unit': 'users/s',
            'config': {'workload': '%d users x %d items, k=%d, top-%d, train history masked' % (r['n_users'], n_items, k, K)},
            'ms_per_pass': wall * 1e3 / reps,
            'roofline': dict(topk_roofline(tf, k),
                             traffic=None,
                             traffic_from_profile=pmc_traffic('score_topk_ml10m_k128') if (k == 128 and n_items == 10380 and world == 1) else None,
                             mfma_busy_from_profile=(pmc_mfma('<refine>, 69,878') if topk_math(k) == 'refine' else pmc_mfma('(fp32 MFMA), 69,878'))
                             if (k == 128 and n_items == 10380 and world == 1) else None,
                             counters_from_profile=pmc_k4('ml') if (k == 128 and n_items == 10380 and world == 1 and topk_math(k) == 'refine') else None,
                             launch_ms=launch_ms, algorithmic_flops_per_launch=flops),
            'ms_per_pass_other_arithmetics': others}


def topk_bench_netflix(k, device, K=30, reps=3):
    """BASELINE.json configs[4] shape on ONE GPU: 480,189 users x 17,770 items, every user with 150 random
    train-rated items masked (the synthetic Netflix-size rating file is not generated for this leg)"""
    import tkr_hip
    n_users, n_items, deg = 480189, 17770, 150
    g = torch.Generator(device=device)
    g.manual_seed(11)
    U = (torch.randn((n_users, k), device=device, generator=g) * 0.01 * 1e6).round() / 1e6
    V = (torch.randn((n_items, k), device=device, generator=g) * 0.01 * 1e6).round() / 1e6
    ptr = torch.arange(0, (n_users + 1) * deg, deg, dtype=torch.int64, device=device)
    cols = torch.randint(0, n_items, (n_users * deg,), device=device, generator=g, dtype=torch.int32)
    mask, pitch = tkr_hip.build_rated_mask(ptr, cols, n_users, n_items)
    tkr_hip.score_topk(U, V, K, mask=mask, mask_pitch=pitch)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(reps):
        tkr_hip.score_topk(U, V, K, mask=mask, mask_pitch=pitch)
    e1.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ms = e0.elapsed_time(e1) / reps
    others = topk_other_arithmetics(lambda: tkr_hip.score_topk(U, V, K, mask=mask, mask_pitch=pitch), k)
    flops = 2.0 * k * n_items * n_users
    tf = flops / (ms * 1e-3) / 1e12
    return {'value': n_users * reps / wall, 'unit': 'users/s', 'ms_per_pass': wall * 1e3 / reps,
            'config': {'workload': '%d users x %d items, k=%d, top-%d, %d rated items per user masked' % (n_users, n_items, k, K, deg)},
            'roofline': dict(topk_roofline(tf, k), launch_ms=ms, traffic=None,
                             traffic_from_profile=pmc_traffic('score_topk_netflix_k128') if k == 128 else None,
                             mfma_busy_from_profile=pmc_mfma('<refine>, 480,189') if (k == 128 and topk_math(k) == 'refine') else None,
                             counters_from_profile=pmc_k4('nf') if (k == 128 and topk_math(k) == 'refine') else None),
            'ms_per_pass_other_arithmetics': others}


def vbpr_bench(r, csr, k, device, B=256, d=20000, steps=256, warmup=32):
    """BASELINE.json configs[2]: VBPR, ML-10M shape, content features d=20,000 (train.py:11), ~100 nnz/row, resident as a
    dense fp32 matrix like the reference's.  Default: the engine takes the CSR/CSC view of such features (S1/S3 kernels,
    HBM-bound); `dense_view` times the dense fp32-MFMA kernels (V1/V3) on the same data."""
    import synth
    from single import _engine
    n_users, n_items = r['n_users'], r['n_in'] + r['n_out']
    g = torch.Generator(device=device)
    g.manual_seed(7)
    feat = torch.zeros((n_items, d), device=device)                      # tf-idf-like: ~100 positive entries per row, L2-normalised
    cols = torch.randint(0, d, (n_items, 100), device=device, generator=g)
    feat.scatter_(1, cols, torch.rand((n_items, 100), device=device, generator=g) + 0.1)
    feat /= feat.norm(dim=1, keepdim=True)
    hp = dict(lu=2.5e-3, li=2.5e-3, lj=2.5e-4, lb=0.0, le=0.0, lr=1e-4, mode='l2')
    kh = k // 2
    nnz = int(torch.count_nonzero(feat))
    out = {}
    for key, sparse in (('sparse_view', None), ('dense_view', False)):
        eng = _engine.VbprEngine(n_users, n_items, k, d, feat, hp, device, seed=3, sparse=sparse)
        wall, step_ms, _ = timed_run(eng, csr, B, steps, warmup, 10 ** 9, 1, names=eng.replicated_names)
        step_s = step_ms * 1e-3 / steps
        if eng.sparse is not None:
            # dense optimizer traffic (cem and its slot, read + written: TF's dense RMSProp moves every element every batch) + the
            # column plan of the batch (a header per column, 16 B per nonzero of the 2B feature rows) + one cem row and one icb value per
            # such nonzero.  (Round 2 also walked the whole CSC of feat per batch, 8 B per nonzero of feat: gone with the column plan.)
            ent = 2.0 * B * (nnz / n_items)
            bytes_ = 16.0 * d * kh + 32.0 * d + 16.0 * ent + ent * (4.0 * kh + 12.0)
            gbs = bytes_ / step_s / 1e9
            roof = {'kernels': 'tkr::vbpr_tproject (project + alpha, beta) / pairsum / update (row tasks + column tasks): 3 launches per batch; '
                               'tkr::vbpr_colplan beside K1' if eng.wants_cols(B) else
                               'tkr::vbpr_sproject / pair / rows / sdense (4 launches per batch)', 'bound': 'hbm', 'achieved': gbs,
                    'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': gbs / HBM_PEAK_GBS, 'algorithmic_bytes_per_launch_chain': bytes_,
                    'step_us': step_s * 1e6, 'traffic': None,
                    'traffic_from_profile': (sum(pmc_traffic('vbpr_%s_B256' % n) or 0 for n in ('tproject', 'pairsum', 'update')) or None)
                                            if (eng.wants_cols(B) and B == 256 and d == 20000) else None}
        else:
            flops = 4.0 * d * kh * B                                     # project the difference once, fwd + dense gradient
            tf = flops / step_s / 1e12
            bytes_ = B * 2 * 4 * d * 2 + 16.0 * d * kh                   # feature rows (V1 + V3) + dense optimizer traffic
            roof = {'kernels': 'tkr::vbpr_project/reduce/occur/pair/rows/dense (6 launches per batch)', 'bound': 'mfma', 'achieved': tf,
                    'peak': MFMA_F32_PEAK_TF, 'unit': 'TFLOP/s', 'frac': tf / MFMA_F32_PEAK_TF,
                    'hbm_GBps_algorithmic': bytes_ / step_s / 1e9, 'hbm_frac': bytes_ / step_s / 1e9 / HBM_PEAK_GBS, 'step_us': step_s * 1e6,
                    'mfma_busy_from_profile': [pmc_mfma('vbpr_project_kernel'), pmc_mfma('vbpr_dense_kernel')] if (B == 256 and d == 20000) else None}
        out[key] = {'value': steps * B / wall, 'unit': 'triplets/s', 'steps': steps, 'ms_per_step': wall * 1e3 / steps, 'roofline': roof}
        del eng
    res = dict(out['sparse_view'])
    res['config'] = {'workload': 'VBPR ML-10M shape, k=%d (kh=%d), content features d=%d with %d nonzeros (%.2f %% dense), batch_size=%d'
                                 % (k, kh, d, nnz, 100.0 * nnz / (n_items * d), B)}
    res['dense_view'] = out['dense_view']
    # throughput mode of the same model: batch_size 8192 (a legal train() argument), sparse view
    eng = _engine.VbprEngine(n_users, n_items, k, d, feat, hp, device, seed=3)
    Bt, st_ = 8192, 48
    wall, step_ms, _ = timed_run(eng, csr, Bt, st_, st_, 10 ** 9, 1, names=eng.replicated_names)
    step_s = step_ms * 1e-3 / st_
    bytes_ = 16.0 * d * kh + 8.0 * nnz + 2.0 * Bt * (nnz / n_items) * (4.0 * kh + 12.0) + Bt * (48.0 * kh + 56)
    res['throughput_mode'] = {'batch_size': Bt, 'steps': st_, 'value': st_ * Bt / wall, 'unit': 'triplets/s', 'ms_per_step': wall * 1e3 / st_,
                              'roofline': {'bound': 'hbm', 'achieved': bytes_ / step_s / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                           'frac': bytes_ / step_s / 1e9 / HBM_PEAK_GBS, 'step_us': step_s * 1e6, 'traffic': None}}
    del eng
    # the literal reading of BASELINE.json configs[2] ("d=128"): DENSE content features of width d_c = 128
    dc = 128
    featd = torch.rand((n_items, dc), device=device, generator=g) + 0.1
    featd /= featd.norm(dim=1, keepdim=True)
    eng = _engine.VbprEngine(n_users, n_items, k, dc, featd, hp, device, seed=3)
    assert eng.wants_cols(B)                 # a narrow feat takes the gather view + column plan whatever its density (the fp32-MFMA kernels V1 / V3
                                             # tile d by 128 / 64 columns: 1-2 workgroups at d_c = 128, 55 us per batch in round 2); every column meets
                                             # every triplet, so a column's run (2B entries) is split over the 16 groups of its workgroup
    wall, step_ms, _ = timed_run(eng, csr, B, steps, warmup, 10 ** 9, 1, names=eng.replicated_names)
    step_s = step_ms * 1e-3 / steps
    bytes_ = B * (2 * 4 * dc * 2 + 48.0 * kh + 56) + 16.0 * dc * kh       # feature rows (V1 + V3) + the sparse rows + dense optimizer traffic
    res['dense_dc128'] = {'value': steps * B / wall, 'unit': 'triplets/s', 'steps': steps, 'ms_per_step': wall * 1e3 / steps,
                          'config': {'workload': 'VBPR ML-10M shape, k=%d, DENSE content features d_c=%d, batch_size=%d' % (k, dc, B)},
                          'roofline': {'kernels': 'tkr::vbpr_tproject / pairsum / update (3 launches per batch)', 'bound': 'hbm',
                                       'achieved': bytes_ / step_s / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': bytes_ / step_s / 1e9 / HBM_PEAK_GBS,
                                       'mfma_TFLOPs': 4.0 * dc * kh * B / step_s / 1e12, 'step_us': step_s * 1e6, 'traffic': None}}
    del eng
    return res


def netflix_train_bench(k, device, B=256, steps=4096, warmup=1536):
    """BASELINE.json configs[3] shape on ONE GPU (the 8-GPU run shards these users): 480,189 users x 17,770 items, ~3.5e7
    train positives generated straight as CSR (synth.train_csr_shape), BPR defaults, batch 256"""
    import synth
    from single import _engine
    n_users, n_items = 480189, 17770
    row_ptr, pos, _, tr_users = synth.train_csr_shape(n_users, n_items, mean_pos=90.0, seed=43)
    csr = _engine.TrainingCSR.from_arrays(row_ptr, pos, tr_users, device)
    hp = dict(lu=2.5e-3, li=2.5e-3, lj=2.5e-4, lb=0.0, lr=1e-4, mode='l2')
    eng = _engine.BprEngine(n_users, n_items, k, hp, device, seed=4321)
    wall, step_ms, _ = timed_run(eng, csr, B, steps, warmup, 10 ** 9, 1)
    us = step_ms * 1e3 / steps
    gbs = B * algorithmic_bytes_per_triplet(k) / (us * 1e-6) / 1e9
    return {'value': steps * B / wall, 'unit': 'triplets/s', 'steps': steps, 'ms_per_step': wall * 1e3 / steps,
            'config': {'workload': 'BPR Netflix shape (%d users x %d items, %d train positives), k=%d, batch_size=%d, one GPU'
                                   % (n_users, n_items, int(row_ptr[-1]), k, B)},
            'roofline': {'kernel': step_kernel(eng, B)[0], 'bound': 'hbm', 'achieved': gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': gbs / HBM_PEAK_GBS, 'launch_us': us, 'traffic': None}}


def topk_cpu_baseline(r, k, K=30, budget_s=12.0, slice_users=2000):
    """evaluate.py's operations on user slices until the time budget is spent:
    np.dot -> np.argsort -> python rank walk (oracle restatement of evaluate.py:78-105)"""
    import synth
    n_items = r['n_in'] + r['n_out']
    rng = np.random.Generator(np.random.PCG64(0))
    V = np.round(rng.standard_normal((n_items, k)) * 0.01, 6).astype(np.float32)
    ptr, cols = synth.rated_csr(r)
    done, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s and done + slice_users <= r['n_users']:
        U = np.round(rng.standard_normal((slice_users, k)) * 0.01, 6).astype(np.float32)
        order = np.argsort(np.dot(U, V.T), axis=1)
        for q in range(slice_users):
            u = done + q
            rated = set(cols[ptr[u]:ptr[u + 1]].tolist())
            kept = 0
            for c in order[q, ::-1]:
                if c not in rated:
                    kept += 1
                    if kept == K:
                        break
        done += slice_users
    dt = time.perf_counter() - t0
    return dict(value=done / dt, unit='users/s', cores=os.cpu_count(), kind='port',
                sample='%d users x %d items (%.1f s): np.dot (BLAS threads = host cores) + np.argsort + python rank walk '
                       '(single thread), oracle restatement of evaluate.py:78-105' % (done, n_items, dt))


def shards_on_one_gpu(r, k, device, B, single_value, limit=10 ** 6, epochs=2):
    """S user shards with replicated item tables on ONE GPU, each the persistent step of batch size B on CUs // S owners and a HIP
    stream of its own, reconciled once per epoch by dist.LocalShards (pack, sum, unpack): north_star's sharded semantics (one
    reconcile per epoch, /root/reference single/bpr.py:136-147 is the loop each shard runs) inside one device -- the rehearsal of the
    8-GPU run, and what a single sequential stream (10 % of HBM by construction) leaves idle.  Whole epochs WITH their exchanges."""
    import synth
    import dist as tdist
    from single import _engine
    row_ptr, pos, _, tr_users = synth.positives_csr(r)
    n_users, n_items = r['n_users'], r['n_in'] + r['n_out']
    hp = dict(lu=2.5e-3, li=2.5e-3, lj=2.5e-4, lb=0.0, lr=1e-4, mode='l2')
    out = {'what': 'S shards x (epoch of (limit // B) // S batches + exchange), aggregate over %d epochs after one warm-up epoch; every shard holds a full '
                   'user table and samples inside its own users (dist.shard_users)' % epochs, 'single_stream_value': single_value}
    for S in (2, 4, 8):
        try:
            nb = tdist.batches_per_rank(limit // B, S)
            engines, csrs, streams = [], [], []
            for q in range(S):
                e = _engine.BprEngine(n_users, n_items, k, hp, device, seed=1234)
                e.ranks_on_device, e.private_side_stream = S, True
                e.own_min_owners = torch.cuda.get_device_properties(device).multi_processor_count // 2      # as BPR._train_streams
                e.plan_in_order = S + 1 <= 4 < 2 * S + 1              # as BPR._train_streams: no planner streams where they would share hardware queues
                e.prepare(B)
                e.triplets_drawn = q * (epochs + 1) * nb * B
                engines.append(e)
                csrs.append(_engine.TrainingCSR.from_arrays(row_ptr, pos, np.asarray(tdist.shard_users(tr_users, q, S), dtype=np.int32), device))
                streams.append(torch.cuda.Stream(device=device))
            shards = tdist.LocalShards(engines, streams)

            def epoch(more):
                shards.begin()
                for e, c, st in zip(engines, csrs, streams):
                    with torch.cuda.stream(st):
                        e.run_batches(c, nb, B, want_loss=WANT_LOSS, then_exchange=nb if more else 0)
                shards.end()
            epoch(True)
            torch.cuda.synchronize(device)
            shards.timing = []
            t0 = time.perf_counter()
            for ep in range(epochs):
                epoch(ep + 1 < epochs)
            torch.cuda.synchronize(device)
            wall = time.perf_counter() - t0
            gave_up = shards.any_gave_up()
            parts = [(a.elapsed_time(b), b.elapsed_time(c), c.elapsed_time(d)) for a, b, c, d in shards.timing]
            x = {'pack': 1e3 * sum(p[0] for p in parts) / len(parts), 'sum': 1e3 * sum(p[1] for p in parts) / len(parts),
                 'unpack': 1e3 * sum(p[2] for p in parts) / len(parts)}
            x['total'] = x['pack'] + x['sum'] + x['unpack']
            value = S * epochs * nb * B / wall
            out['S%d' % S] = {'shards': S, 'batches_per_shard_per_epoch': nb, 'owners_per_shard': engines[0]._plan_owners(B), 'step': step_kernel(engines[0], B)[0],
                              'value': value, 'unit': 'triplets/s', 'over_single_stream': value / single_value if single_value else None,
                              'us_per_batch_per_shard': wall * 1e6 / (epochs * nb), 'ms_per_epoch': wall * 1e3 / epochs, 'exchange_us': x,
                              'gave_up': bool(gave_up),
                              # the same exchange beside an 8-rank epoch of the single-stream rate (488 batches): what it costs before the collective
                              'exchange_share_of_8_rank_epoch': x['total'] * 1e-6 / ((limit // B) // 8 * B / single_value) if single_value else None}
            del engines, csrs, shards
            torch.cuda.empty_cache()
        except Exception as ex:        # a leg of extras never takes the line down
            out['S%d' % S] = {'error': '%s: %s' % (type(ex).__name__, ex)}
    return out


def summary(out):
    """the numbers VERDICT tracks, compact, as the last key of the line (the driver's record keeps only the tail of stdout)"""
    def leg(key, *path):
        v = out.get(key)
        for q in path:
            v = v.get(q) if isinstance(v, dict) else None
        return round(v, 6) if isinstance(v, float) else v
    s = {'value_Mtps': round(out['value'] / 1e6, 3), 'ms_per_step': round(out['ms_per_step'], 6), 'kernel': out['roofline']['kernel'].split(' ')[0],
         'steady_Mtps': (leg('steady_state', 'value') or 0) / 1e6 or None, 'steady_us_per_batch': leg('steady_state', 'roofline', 'launch_us'),
         'steady_frac': leg('steady_state', 'roofline', 'frac'),
         'netflix_Mtps': (leg('bpr_netflix_shape', 'value') or 0) / 1e6 or None, 'netflix_us_per_batch': leg('bpr_netflix_shape', 'roofline', 'launch_us'),
         'B8192_frac': leg('throughput_mode', 'roofline', 'frac'), 'B65536_frac': leg('throughput_mode_B65536', 'roofline', 'frac'),
         'topk_ms': leg('topk', 'ms_per_pass'), 'topk_frac': leg('topk', 'roofline', 'frac'),
         'topk_nf_ms': leg('topk_netflix_shape', 'ms_per_pass'), 'topk_nf_frac': leg('topk_netflix_shape', 'roofline', 'frac'),
         'topk_k512_ms': leg('topk_wide', 'k512', 'ms_per_pass'), 'topk_k768_ms': leg('topk_wide', 'k768', 'ms_per_pass'),
         'vbpr_ms': leg('vbpr', 'ms_per_step'), 'vbpr_frac': leg('vbpr', 'roofline', 'frac'),
         'vbpr_dc128_ms': leg('vbpr', 'dense_dc128', 'ms_per_step'),
         'cpu_Mtps': (leg('cpu_baseline', 'value') or 0) / 1e6 or None}
    for S in (2, 4, 8):
        if leg('shards_on_one_gpu', 'S%d' % S, 'value'):
            s['shards%d_Mtps' % S] = leg('shards_on_one_gpu', 'S%d' % S, 'value') / 1e6
            s['shards%d_owners' % S] = leg('shards_on_one_gpu', 'S%d' % S, 'owners_per_shard')
            s['shards%d_exchange_us' % S] = leg('shards_on_one_gpu', 'S%d' % S, 'exchange_us', 'total')
    if out.get('n_gpus', 1) > 1:
        s['exchange_us'] = leg('epoch_mode', 'exchange_us')
        s['epoch_Mtps'] = (leg('epoch_mode', 'value') or 0) / 1e6 or None
        s['owners'] = leg('roofline', 'owners')           # K2o owners of this rank (0: K2f / K2) ...
        s['ranks_on_device'] = out.get('ranks_on_device')  # ... and how many ranks share its GPU: a silently degraded rank shows in the tail
    return {k: v for k, v in s.items() if v is not None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3906 * 2)          # two reference epochs (10^6 // 256 batches each)
    ap.add_argument('--warmup', type=int, default=1536)             # three chunks: both plan buffers and the side stream have been through a full cycle
    ap.add_argument('--batch-size', type=int, default=256)          # train.py / single/bpr.py:103 default
    ap.add_argument('--k', type=int, default=128)
    ap.add_argument('--shape', default='ml10m', choices=['ml10m', 'netflix'])
    ap.add_argument('--epoch-sample-limit', type=int, default=10 ** 6)   # train.py:6
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true')
    ap.add_argument('--no-live-traffic', action='store_true')       # skip the two rocprofv3 counter passes behind roofline.traffic (~25 s)
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    assert world == args.gpus, 'launch with torch.distributed.run --nproc-per-node %d' % args.gpus
    if os.environ.get('TKR_BENCH_SINGLE_DEVICE') == '1':        # test hook: several ranks share GPU 0 (with gloo)
        local = 0
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    if world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        backend = os.environ.get('TKR_BENCH_BACKEND', 'nccl')       # 'nccl' = RCCL over xGMI
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)
        else:
            dist.init_process_group(backend)

    B, k = args.batch_size, args.k
    r, csr, eng, nnz = build_problem(args.shape, k, rank, world, device)
    sync_every = max(1, (args.epoch_sample_limit // B) // world)       # batches per rank and epoch, as BPR.train deals them (dist.batches_per_rank)
    # the headline: per-launch events only where the launches are long (a run of at least one full chunk)
    wall, step_ms, loop = timed_run(eng, csr, B, args.steps, args.warmup, sync_every, world, per_launch_events=args.steps >= 512)
    loop_timed_exchanges = loop.timed_exchanges
    wall = max_over_ranks(wall, device, world)
    raw_wall = wall
    launch_us = step_ms * 1e3 / args.steps
    achieved = B * algorithmic_bytes_per_triplet(k) / (launch_us * 1e-6) / 1e9
    # whole epochs with K1 of every batch (and, at N > 1, every exchange) inside the timed region: at N = 1 the steady state of the headline
    em = epoch_mode(eng, csr, B, k, world, device, loop)
    # SURVEY §8d defines the metric with the all-reduce inside the wall.  A short timed window at N > 1 holds fewer exchanges than its
    # share (`--steps 20` of a 488-batch epoch: none), so the headline charges the share that is missing at the exchange wall
    # measured over whole epochs in this same process (epoch_mode.exchange_us: pack + collective + unpack + what the boundary exposes)
    exch_ms = 0.0
    share = 0.0
    if world > 1 and 'exchange_us' in em:
        x = em['exchange_us']
        exch_ms = max_over_ranks((x['pack'] + x['collective'] + x['unpack'] + x.get('exposed_after_exchange', 0.0)) * 1e-3, device, world)
        share = max(0.0, args.steps / float(sync_every) - loop_timed_exchanges)
        wall = raw_wall + share * exch_ms * 1e-3
    value = world * args.steps * B / wall
    out = {
        'metric': 'BPR training triplets/sec (sampler + plan + step), %s shape' % ('MovieLens-10M' if args.shape == 'ml10m' else 'Netflix'),
        'value': value, 'unit': 'triplets/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': wall * 1e3 / args.steps, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': 'BPR %s k=%d B=%d' % ('ML-10M' if args.shape == 'ml10m' else 'Netflix', k, B),
                   'detail': '%dx%d, %d positives, k=%d, B=%d, RMSProp 1e-4, ref. defaults' % (r['n_users'], r['n_in'] + r['n_out'], nnz, k, B),
                   'batch_size': B, 'k': k, 'sharding': 'users sharded over %d GPU(s), item tables replicated, '
                                                        'all-reduce every %d steps' % (world, sync_every) if world > 1 else 'single GPU'},
        'timed_region': {'per_batch': ['K1 tkr_sample_plan: (u,i,j) draw + plan of the timed batches (planned inside the region: nothing is left over '
                                       'from the warm-up)', step_kernel(eng, B)[1] + (' with the loss of every batch (the obj of sess.run([solver, obj]))' if WANT_LOSS else ' WITHOUT the per-batch loss')],
                         'exchanges_inside': loop_timed_exchanges,
                         'exchange_share_charged': share, 'exchange_ms_charged_each': exch_ms,
                         'raw': {'wall_ms': raw_wall * 1e3, 'value': world * args.steps * B / raw_wall,
                                 'note': 'the --steps batches alone, barrier to barrier, before the missing share of the exchange is charged'},
                         'note': 'exactly --steps batches; at N > 1 the exchange keeps its per-epoch cadence (every %d batches, counted from the first '
                                 'warm-up batch); value = triplets / (timed wall + (steps / %d - exchanges inside) x the exchange wall measured over '
                                 'whole epochs in epoch_mode): the all-reduce is inside the metric as SURVEY 8d defines it' % (sync_every, sync_every)},
        'roofline': {'kernel': step_kernel(eng, B)[0] + (' (one persistent launch per chunk of batches)' if eng.layout == 'flow' else ''),
                     'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': achieved / HBM_PEAK_GBS, 'traffic': None,
                     'traffic_from_profile': pmc_traffic(step_kernel(eng, B)[2]) if (k == 128 and args.shape == 'ml10m') else None,
                     'owners': eng._plan_owners(B),    # K2o: workgroups that own item rows (the CUs, split between the ranks that share a GPU); 0: K2f / K2
                     'launch_us': launch_us,          # per BATCH: the persistent kernel's launch covers many batches, duration / batches
                     'events': 'around every step launch' if args.steps >= 512 else 'ONE pair around the timed region (a short run is one launch that plans and steps; '
                               'the pair also spans the host\'s way to that launch: an upper bound of the launch)',
                     'algorithmic_bytes_per_launch': B * algorithmic_bytes_per_triplet(k)},
    }
    out['epoch_mode'] = em
    out['ranks_on_device'] = int(getattr(eng, 'ranks_on_device', 1))
    if world == 1:
        out['steady_state'] = em
    if rank == 0 and world == 1 and not args.no_extras and not args.no_live_traffic and eng.layout == 'flow':
        t_bytes, t_note = live_traffic(k, B, args.shape)
        out['roofline']['traffic'] = t_bytes
        out['roofline']['traffic_how'] = t_note
        if t_bytes:
            out['roofline']['traffic_over_algorithmic'] = t_bytes / (B * algorithmic_bytes_per_triplet(k))
    if rank == 0 and world == 1 and not args.no_extras:
        # throughput mode: same kernels, batch_size 8192 (fresh tables)
        from single import _engine
        B2 = 8192
        eng2 = _engine.BprEngine(eng.n_users, eng.n_items, k, eng.hp, device, seed=99)
        T2 = 1024     # 8 plan chunks of 128: the first chunk's planning is the only one the steps do not hide (with 2 chunks it is half)
        w2, s2, _ = timed_run(eng2, csr, B2, T2, T2, 10 ** 9, 1)       # warm-up as long as the run: same chunking, side stream warm
        a2 = B2 * algorithmic_bytes_per_triplet(k) / (s2 * 1e-3 / T2) / 1e9
        out['throughput_mode'] = {'batch_size': B2, 'steps': T2, 'value': T2 * B2 / w2, 'unit': 'triplets/s',
                                  'ms_per_step': w2 * 1e3 / T2,
                                  'roofline': {'bound': 'hbm', 'achieved': a2, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                               'frac': a2 / HBM_PEAK_GBS, 'launch_us': s2 * 1e3 / T2,
                                               'traffic': None,
                                               'traffic_from_profile': pmc_traffic('bpr_step_B8192') if (k == 128 and args.shape == 'ml10m') else None,
                                               'hbm_utilisation_from_profile': (pmc_traffic('bpr_step_B8192') / (s2 * 1e-3 / T2) / 1e9 / HBM_PEAK_GBS)
                                               if (k == 128 and args.shape == 'ml10m' and pmc_traffic('bpr_step_B8192')) else None}}
        del eng2
        # batch_size 65,536 and 1,048,576 (planned grid-wide: csrc/planner_big.hip)
        for Bb, Tb in ((65536, 128), (1048576, 8)):
            engb = _engine.BprEngine(eng.n_users, eng.n_items, k, eng.hp, device, seed=98)
            wb, sb, _ = timed_run(engb, csr, Bb, Tb, Tb, 10 ** 9, 1)
            ab = Bb * algorithmic_bytes_per_triplet(k) / (sb * 1e-3 / Tb) / 1e9
            out['throughput_mode_B%d' % Bb] = {'batch_size': Bb, 'steps': Tb, 'value': Tb * Bb / wb, 'unit': 'triplets/s', 'ms_per_step': wb * 1e3 / Tb,
                                               'roofline': {'bound': 'hbm', 'achieved': ab, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': ab / HBM_PEAK_GBS,
                                                            'launch_us': sb * 1e3 / Tb, 'traffic': None,
                                                            'traffic_from_profile': pmc_traffic('bpr_step_B%d' % Bb) if (k == 128 and args.shape == 'ml10m') else None,
                                                            # the REAL utilisation: counter bytes of the committed profile over this run's launch time
                                                            # (duplicate draws of a large batch hit in cache: the algorithmic fraction overstates it)
                                                            'hbm_utilisation_from_profile': (pmc_traffic('bpr_step_B%d' % Bb) / (sb * 1e-3 / Tb) / 1e9 / HBM_PEAK_GBS)
                                                            if (k == 128 and args.shape == 'ml10m' and pmc_traffic('bpr_step_B%d' % Bb)) else None,
                                                            'note': 'algorithmic bytes give no credit for in-batch duplicates: %d item draws over %d items'
                                                                    % (2 * Bb, eng.n_items)}}
            del engb
            torch.cuda.empty_cache()
        # legacy plain-SGD optimiser (old/methods/bpr.py:57-61): same path, no RMSProp slot traffic
        for Bs, key, steps_s in ((B, 'sgd_mode', 2048), (B2, 'sgd_throughput_mode', T2)):
            eng3 = _engine.BprEngine(eng.n_users, eng.n_items, k, dict(eng.hp, opt='sgd'), device, seed=77)
            w3, s3, _ = timed_run(eng3, csr, Bs, steps_s, steps_s, 10 ** 9, 1)
            a3 = Bs * (24 * k + 40) / (s3 * 1e-3 / steps_s) / 1e9        # 3 rows x (read+write) + biases + ids
            out[key] = {'batch_size': Bs, 'steps': steps_s, 'value': steps_s * Bs / w3, 'unit': 'triplets/s',
                        'ms_per_step': w3 * 1e3 / steps_s,
                        'roofline': {'bound': 'hbm', 'achieved': a3, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': a3 / HBM_PEAK_GBS,
                                     'launch_us': s3 * 1e3 / steps_s, 'algorithmic_bytes_per_triplet': 24 * k + 40, 'traffic': None}}
            del eng3
    if not args.no_extras:
        topk = topk_bench(r, k, device, rank, world)
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            topk['cpu_baseline'] = topk_cpu_baseline(r, k)
        out['topk'] = topk
        if rank == 0 and world == 1:
            # the widths above the matrix-pipe arithmetics: k = 512 (what the trainer goes to) and 768 (K4's limit) through the slab
            # kernels on fp32 MFMA (csrc/topk.hip score_topk_slab_kernel<2 / 3>; the 3-slab form spills 168 registers)
            out['topk_wide'] = {}
            for kw in (512, 768):
                tw = topk_bench(r, kw, device, rank, world, reps=2)
                out['topk_wide']['k%d' % kw] = {'value': tw['value'], 'unit': 'users/s', 'ms_per_pass': tw['ms_per_pass'],
                                                'roofline': {q: tw['roofline'][q] for q in ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac')}}
        if rank == 0 and world == 1:
            out['topk_netflix_shape'] = topk_bench_netflix(k, device)
            out['vbpr'] = vbpr_bench(r, csr, k, device)
            out['bpr_netflix_shape'] = netflix_train_bench(k, device)
            if B <= 512:
                out['shards_on_one_gpu'] = shards_on_one_gpu(r, k, device, B, em['value'], args.epoch_sample_limit)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(r, k, B)
    if rank == 0:
        out['summary'] = summary(out)         # LAST key: the driver keeps the tail of the line
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
