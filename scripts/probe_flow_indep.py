"""K2f without dependencies between batches (uniform draws over huge tables): the task-throughput ceiling of the kernel"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'top-k-rec_amd')]
import numpy as np, torch
import tkr_hip
from single import _engine
n_users = n_items = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
waves = [int(x, 0) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else [4, 8]
B, k, steps, deg = 256, 128, 2048, 8
dev = torch.device('cuda', 0)
rng = np.random.Generator(np.random.PCG64(1))
pos = rng.integers(0, n_items, n_users * deg).astype(np.int32)
row_ptr = np.arange(0, (n_users + 1) * deg, deg, dtype=np.int64)
csr = _engine.TrainingCSR.from_arrays(row_ptr, pos, np.arange(n_users, dtype=np.int32), dev)
hp = dict(lu=2.5e-3, li=2.5e-3, lj=2.5e-4, lb=0.0, lr=1e-4, mode='l2')
eng = _engine.BprEngine(n_users, n_items, k, hp, dev, seed=5)
for w in waves:
    eng.cfg.flow_waves_per_cu = w
    eng.run_batches(csr, 512, B, want_loss=False)
    torch.cuda.synchronize()
    eng.ctl[tkr_hip.FLOW_CTL_SPINS] = 0
    eng.ctl[tkr_hip.FLOW_CTL_PROF:tkr_hip.FLOW_CTL_PROF + 16] = 0
    t0 = time.perf_counter()
    eng.run_batches(csr, steps, B, want_loss=False)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    eng.check()
    spins = int(eng.ctl[tkr_hip.FLOW_CTL_SPINS])
    print('independent rows, %d x %d: waves/CU 0x%04x: %.2f us/batch, %.1f M triplets/s, %.2f spin passes per task' %
          (n_users, n_items, w, wall / steps * 1e6, steps * B / wall / 1e6, spins / (steps * 3.0 * B)), flush=True)
    if os.environ.get('TKR_FLOW_PROFILE') == '1':
        pr = eng.ctl[tkr_hip.FLOW_CTL_PROF:tkr_hip.FLOW_CTL_PROF + 16].cpu().numpy().view(np.uint64)
        tasks = max(int(pr[5]), 1)
        print('   cycles per task: grab %.0f  record %.0f  rows+math %.0f  war %.0f  finish %.0f' % tuple(pr[q] / tasks for q in range(5)), flush=True)
