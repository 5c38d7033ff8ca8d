"""dev probe: K1/K2 timings at ML-10M shape for several batch sizes (not part of the product)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'top-k-rec_amd')]
import numpy as np, torch
import synth, tkr_hip
from single import _engine

k = int(os.environ.get('K', 128))
t0 = time.time()
r = synth.make_ratings(**synth.ML10M, seed=42)
row_ptr, pos, srt, tr_users = synth.positives_csr(r)
print('synth %.1fs nnz=%d tr_users=%d' % (time.time() - t0, len(pos), len(tr_users)), flush=True)
dev = torch.device('cuda', 0)
n_users, n_items = r['n_users'], r['n_in'] + r['n_out']
csr = _engine.TrainingCSR.from_arrays(row_ptr, pos, tr_users, dev)
hp = dict(lu=2.5e-3, li=2.5e-3, lj=2.5e-4, lb=0.0, lr=1e-4, mode='l2')
for B in [int(x) for x in os.environ.get('BS', '256,2048,8192').split(',')]:
    eng = _engine.BprEngine(n_users, n_items, k, hp, dev, seed=1)
    nb = max(8, min(3906, (1 << 21) // B))
    eng.run_batches(csr, nb, B, want_loss=False)            # warm-up
    torch.cuda.synchronize()
    eng.step_events = []
    t0 = time.perf_counter()
    eng.run_batches(csr, nb, B, want_loss=False)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ts = sum(a.elapsed_time(b) for a, b, _ in eng.step_events)
    eng.step_events = None
    hd = eng.plan.hdr.view(-1, 4)[:8].float().mean(0).tolist()
    print('B=%5d nb=%4d wall %.2f us/batch | step launches %.2f us/batch -> %.1f M triplets/s, %.1f GB/s algorithmic | hdr mean %s'
          % (B, nb, wall / nb * 1e6, ts / nb * 1e3, nb * B / wall / 1e6, nb * B * (48 * k + 56) / (ts * 1e-3) / 1e9, hd), flush=True)
