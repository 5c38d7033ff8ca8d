"""timing of the large-batch path: python scripts/probe_bigB.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'top-k-rec_amd')]
import numpy as np, torch
import bench
from single import _engine
dev = torch.device('cuda', 0)
r, csr, eng, nnz = bench.build_problem('ml10m', 128, 0, 1, dev)
for B, T in ((8192, 256), (65536, 64), (1048576, 6)):
    e = _engine.BprEngine(eng.n_users, eng.n_items, 128, eng.hp, dev, seed=9)
    e.run_batches(csr, T, B, want_loss=False); torch.cuda.synchronize()
    e.step_events = []
    t0 = time.perf_counter()
    e.run_batches(csr, T, B, want_loss=False); torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    step_ms = sum(a.elapsed_time(b) for a, b, _ in e.step_events)
    e.step_events = None
    print('B %8d: %.1f M triplets/s wall, step %.1f us/batch -> %.0f GB/s algorithmic (%.1f %% of 8 TB/s); wall/batch %.1f us'
          % (B, T * B / wall / 1e6, step_ms * 1e3 / T, B * 6200 / (step_ms * 1e-3 / T) / 1e9, B * 6200 / (step_ms * 1e-3 / T) / 8e10, wall / T * 1e6), flush=True)
    del e; torch.cuda.empty_cache()
