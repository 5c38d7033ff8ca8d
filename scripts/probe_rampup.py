"""per-call time of consecutive 512-batch calls from a cold process: is the slow start of a run a matter of time (clocks), of calls, or of data?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'top-k-rec_amd')]
import torch
import bench

dev = torch.device('cuda', 0)
r, csr, eng, nnz = bench.build_problem('ml10m', 128, 0, 1, dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
idle = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
eng.run_batches(csr, 512, 256, want_loss=False)
torch.cuda.synchronize()
if idle:
    time.sleep(idle)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
ev[0].record()
for q in range(n):
    eng.run_batches(csr, 512, 256, want_loss=False)
    ev[q + 1].record()
torch.cuda.synchronize()
print('us per batch, call by call:', ' '.join('%.2f' % (ev[q].elapsed_time(ev[q + 1]) * 1e3 / 512) for q in range(n)))
