"""persistent dataflow step (K2f) vs one launch per batch (K2) by batch size: python scripts/probe_flow_vs_launch.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'top-k-rec_amd')]
import numpy as np, torch
import bench
from single import _engine
dev = torch.device('cuda', 0)
r, csr, eng, nnz = bench.build_problem('ml10m', 128, 0, 1, dev)
for B in (64, 128, 256, 512, 1024):
    T = max(256, 262144 // B)
    for layout in ('flow', 'bulk'):
        e = _engine.BprEngine(eng.n_users, eng.n_items, 128, eng.hp, dev, seed=9)
        e.wants_flow = (lambda B_, v=(layout == 'flow'): v)
        e.run_batches(csr, T, B, want_loss=False); torch.cuda.synchronize()
        t0 = time.perf_counter()
        e.run_batches(csr, T, B, want_loss=False); torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        e.check()
        print('B %5d %s: %.2f us/batch, %.1f M triplets/s' % (B, 'K2f persistent' if layout == 'flow' else 'K2 per launch ', wall / T * 1e6, T * B / wall / 1e6), flush=True)
        del e
