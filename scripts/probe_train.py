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
    nb = max(8, min(3906, (1 << 20) // B))
    plan = eng._ensure_plan(nb, B)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    nb = min(nb, plan.cap)
    st = eng.state()
    for rep in range(3):
        ev[0].record()
        tkr_hip.sample_plan(csr, n_users, n_items, 1, rep * nb * B, nb, B, eng.cnt, plan)
        ev[1].record()
        t1 = time.time()
        tkr_hip.bpr_run(st, plan, B, nb, None)
        t2 = time.time()
        ev[2].record()
        torch.cuda.synchronize()
        tp, ts = ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])
    ntask = int((plan.task.view(nb, 3 * B, 4)[:, :, 0] != -1).sum()) / nb
    maxocc = int(plan.task.view(nb, 3 * B, 4)[:, :, 2].max())
    hd = plan.hdr.view(-1, 4)[:nb].float().mean(0).tolist()
    print('hdr mean (blocks, light, heavy, tasks):', hd)
    print('B=%5d nb=%4d plan %.3f ms (%.2f us/batch) | step %.3f ms = %.2f us/batch host-issue %.2f us/batch -> %.1f M triplets/s, '
          '%.1f GB/s algorithmic | tasks/batch %.0f max_occ %d' % (B, nb, tp, tp / nb * 1e3, ts, ts / nb * 1e3, (t2 - t1) / nb * 1e6,
          nb * B / ((tp + ts) * 1e-3) / 1e6, nb * B * (48 * k + 56) / (ts * 1e-3) / 1e9, ntask, maxocc), flush=True)
