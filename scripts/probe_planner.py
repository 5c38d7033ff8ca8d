"""K1 time per batch, per-batch planner vs grid-wide planner (TKR_PLAN_BIG_FROM): python scripts/probe_planner.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'top-k-rec_amd')]
import numpy as np, torch
import bench, tkr_hip
from single import _engine
dev = torch.device('cuda', 0)
r, csr, eng, nnz = bench.build_problem('ml10m', 128, 0, 1, dev)
for B, nb in ((256, 512), (1024, 512), (2048, 256), (4096, 128), (8192, 128)):
    plan = _engine.PlanBuffers(nb, B, dev)
    cnt = _engine.UpdateCounters(eng.n_users, eng.n_items, dev)
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        tkr_hip.sample_plan(csr, eng.n_users, eng.n_items, 5, rep * nb * B, nb, B, cnt, plan)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('B %5d x %3d batches (%s planner): %.2f us per batch' % (B, nb, 'grid-wide' if plan.ws is not None else 'per-batch', dt / nb * 1e6), flush=True)
