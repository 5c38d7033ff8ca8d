"""what does the per-batch loss (the `obj` of sess.run([solver, obj]), single/bpr.py:141) cost the step?  python scripts/probe_loss.py [B] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'top-k-rec_amd')]
import torch, bench
dev = torch.device('cuda', 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 7812
r, csr, eng, nnz = bench.build_problem('ml10m', 128, 0, 1, dev)
eng.run_batches(csr, max(steps // 5, 4), B, want_loss=True); torch.cuda.synchronize()
for wl in (False, True, False, True):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.run_batches(csr, steps, B, want_loss=wl)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('want_loss', wl, '%.3f us/batch  %.1f M triplets/s' % (dt / steps * 1e6, steps * B / dt / 1e6), flush=True)
eng.check()
