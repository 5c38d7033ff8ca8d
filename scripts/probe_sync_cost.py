"""Cost of one per-epoch exchange of the replicated item tables WITHOUT the collective (ItemSync.begin + end with the all-reduce
stubbed out), dataflow layout (batch 256) and plain layout (batch 8192).  python scripts/probe_sync_cost.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'top-k-rec_amd')]
import numpy as np, torch
import bench
import dist as tdist
dev = torch.device('cuda', 0)
r, csr, eng, nnz = bench.build_problem('ml10m', 128, 0, 1, dev)
tdist.world = lambda: (0, 8)                                   # pretend: 8 ranks
tdist.dist.all_reduce = lambda t, op=None, group=None: None    # the collective itself is not what is measured
for B in (256, 8192):
    eng.run_batches(csr, 64, B, want_loss=False)
    isync = tdist.ItemSync(eng)
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        isync.begin(); torch.cuda.synchronize(); t1 = time.perf_counter()
        eng.run_batches(csr, 16, B, want_loss=False); torch.cuda.synchronize(); t2 = time.perf_counter()
        isync.end(); torch.cuda.synchronize(); t3 = time.perf_counter()
    isync.timing = []
    for rep in range(4):
        isync.begin(); eng.run_batches(csr, 16, B, want_loss=False); isync.end()
    torch.cuda.synchronize()
    ev = [(a.elapsed_time(b) * 1e3, b.elapsed_time(c) * 1e3, c.elapsed_time(d) * 1e3) for a, b, c, d in isync.timing]
    print('batch %5d (%s layout): begin %.0f us, end %.0f us host wall (no collective); device: pack %.1f us, unpack %.1f us'
          % (B, eng.layout, (t1 - t0) * 1e6, (t3 - t2) * 1e6, min(e[0] for e in ev), min(e[2] for e in ev)), flush=True)
