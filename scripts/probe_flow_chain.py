"""K2f hop latency: batch_size 1, every user's only positive is item 0 -> the task on item 0 of batch t+1 waits for batch t's:
time per batch = one store-to-load hand-off + one light task.  python scripts/probe_flow_chain.py [waves,...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'top-k-rec_amd')]
import numpy as np, torch
import tkr_hip
from single import _engine
waves = [int(x, 0) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 else [4]
n_users = n_items = 100000
dev = torch.device('cuda', 0)
row_ptr = np.arange(n_users + 1, dtype=np.int64)
pos = np.zeros(n_users, dtype=np.int32)
csr = _engine.TrainingCSR.from_arrays(row_ptr, pos, np.arange(n_users, dtype=np.int32), dev)
hp = dict(lu=2.5e-3, li=2.5e-3, lj=2.5e-4, lb=0.0, lr=1e-4, mode='l2')
eng = _engine.BprEngine(n_users, n_items, 128, hp, dev, seed=5)
for B in (1, 4, 8, 12, 16):
    for w in waves:
        eng.cfg.flow_waves_per_cu = w
        eng.run_batches(csr, 512, B, want_loss=False); torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.run_batches(csr, 2048, B, want_loss=False); torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        eng.check()
        print('B=%d (item 0 occurs %d x per batch), waves/CU 0x%04x: %.2f us per batch' % (B, B, w, wall / 2048 * 1e6), flush=True)
