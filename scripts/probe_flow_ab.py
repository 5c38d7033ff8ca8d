"""timing probe of K2f at the headline shape with tune bits: python scripts/probe_flow_ab.py [steps] [0xTTWW,...] [B] [shape]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'top-k-rec_amd')]
os.environ['TKR_OWN'] = '0'
import numpy as np, torch
import bench, tkr_hip
from single import _engine
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
waves = [int(x, 0) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else [8]
B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
shape = sys.argv[4] if len(sys.argv) > 4 else 'ml10m'
dev = torch.device('cuda', 0)
r, csr, eng, nnz = bench.build_problem(shape, 128, 0, 1, dev)
for w in waves:
    eng.cfg.flow_waves_per_cu = w
    eng.run_batches(csr, 1024, B, want_loss=False)
    torch.cuda.synchronize()
    eng.check()
    eng.ctl[tkr_hip.FLOW_CTL_SPINS] = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    eng.run_batches(csr, steps, B, want_loss=False)
    e1.record()
    torch.cuda.synchronize()
    eng.check()
    print('%s B %d K2f 0x%04x bufs %d: %.3f us/batch, %.1f M triplets/s, %.2f spin passes per task' %
          (shape, B, w, eng.V.bufs, e0.elapsed_time(e1) * 1e3 / steps, steps * B / (e0.elapsed_time(e1) * 1e-3) / 1e6,
           int(eng.ctl[tkr_hip.FLOW_CTL_SPINS]) / (steps * 3.0 * B)), flush=True)
