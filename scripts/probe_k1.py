"""dev tool: device time of K1 (owned dataflow plan) for a full chunk and for 20 batches, alone on the GPU"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'top-k-rec_amd')]
import torch
import bench, tkr_hip
from single import _engine
dev = torch.device('cuda', 0)
r, csr, eng, nnz = bench.build_problem('ml10m', 128, 0, 1, dev)
eng.prepare(256)
for owners in (0, tkr_hip.bpr_own_owners(eng.n_items, eng.k)):
    plan = _engine.PlanBuffers(512, 256, dev, flow=True, owners=owners)
    for nb in (512, 20):
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            tkr_hip.sample_plan(csr, eng.n_users, eng.n_items, 7, rep * 512 * 256, nb, 256, eng._cnt, plan)
            e1.record()
            torch.cuda.synchronize()
        print('owners %d nb %d: K1 %.1f us' % (owners, nb, e0.elapsed_time(e1) * 1e3), flush=True)
