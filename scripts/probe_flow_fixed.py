"""Fixed cost of one K2f call: wall time of run_batches(n) + synchronize for small n at the ML-10M-like bench shape.
python scripts/probe_flow_fixed.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'top-k-rec_amd')]
import numpy as np, torch
import synth
from single import _engine
dev = torch.device('cuda', 0)
r = synth.make_ratings(seed=42, **dict(synth.ML10M))
row_ptr, pos, _, tr_users = synth.positives_csr(r)
n_users, n_items = r['n_users'], r['n_in'] + r['n_out']
csr = _engine.TrainingCSR.from_arrays(row_ptr, pos, np.asarray(tr_users, dtype=np.int32), dev)
hp = dict(lu=2.5e-3, li=2.5e-3, lj=2.5e-4, lb=0.0, lr=1e-4, mode='l2')
eng = _engine.BprEngine(n_users, n_items, 128, hp, dev, seed=5)
eng.run_batches(csr, 64, 256, want_loss=False); torch.cuda.synchronize()
for n in (1, 2, 5, 10, 20, 50, 100, 200):
    best, best_ev = 1e9, 1e9
    for rep in range(12):
        torch.cuda.synchronize()
        eng.step_events = []
        t0 = time.perf_counter()
        eng.run_batches(csr, n, 256, want_loss=False)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        ev = sum(a.elapsed_time(b) for a, b, _ in eng.step_events) * 1e3
        eng.step_events = None
        if t2 - t0 < best:
            best, enq, best_ev = t2 - t0, t1 - t0, ev
    eng.check()
    print('n %4d: wall %7.1f us (enqueue %6.1f), events %7.1f us, %.2f us/batch, %.1f M triplets/s' %
          (n, best * 1e6, enq * 1e6, best_ev, best * 1e6 / n, n * 256 / best / 1e6), flush=True)

print('one-shot, as bench.py --steps 20 --warmup 5 times it:')
for trial in range(4):
    eng = _engine.BprEngine(n_users, n_items, 128, hp, dev, seed=5 + trial)
    eng.run_batches(csr, 5, 256, want_loss=False)
    torch.cuda.synchronize(); torch.cuda.synchronize()
    eng.step_events = [] if trial % 2 == 0 else None
    t0 = time.perf_counter()
    eng.run_batches(csr, 20, 256, want_loss=False)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print('  trial %d (events %s): enqueue %6.1f us, first sync %6.1f, second sync %5.1f -> %.1f M triplets/s' %
          (trial, eng.step_events is not None, (t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6, 20 * 256 / (t3 - t0) / 1e6), flush=True)
    eng.step_events = None
    eng.check()
