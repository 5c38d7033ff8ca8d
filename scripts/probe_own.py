"""timing probe of K2o (owned item rows) against K2f at the headline shape:
python scripts/probe_own.py [steps] [variant,variant,...] [B] [shape]; variant = f (K2f) | o<owner waves> (K2o, 0 = default)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'top-k-rec_amd')]
import numpy as np, torch
import bench, tkr_hip
from single import _engine

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
variants = sys.argv[2].split(',') if len(sys.argv) > 2 else ['f', 'o0']
B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
shape = sys.argv[4] if len(sys.argv) > 4 else 'ml10m'
dev = torch.device('cuda', 0)
r, csr, eng, nnz = bench.build_problem(shape, 128, 0, 1, dev)
for v in variants:
    eng.cfg.own = '0' if v == 'f' else '1'
    eng.cfg.own_waves = int(v[1:], 0) if v != 'f' else 0
    eng.run_batches(csr, 512, B, want_loss=False)
    torch.cuda.synchronize()
    eng.check()
    eng.ctl[tkr_hip.FLOW_CTL_SPINS] = 0
    eng.ctl[tkr_hip.FLOW_CTL_PROF:tkr_hip.FLOW_CTL_PROF + 32] = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    eng.run_batches(csr, steps, B, want_loss=False)
    e1.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    eng.check()
    spins = int(eng.ctl[tkr_hip.FLOW_CTL_SPINS])
    print('%s B %d %-3s owners %d: %.3f us/batch (events %.3f), %.1f M triplets/s, %.2f spin passes per task' %
          (shape, B, v, eng._plan_owners(B), wall / steps * 1e6, e0.elapsed_time(e1) * 1e3 / steps, steps * B / wall / 1e6,
           spins / (steps * 3.0 * B)), flush=True)
    ld = eng.ctl[tkr_hip.FLOW_CTL_PROF + 64:tkr_hip.FLOW_CTL_PROF + 68].cpu().numpy().astype(np.float64)
    if ld[0] > 0:
        print('   loader form: %d item tasks through the ring; per task: %.2f spins waiting for its slot, %.2f loader spins blocked; %.1f %% of the tasks waited for a row (own chain or a stale / unfinished partner)'
              % (ld[0], ld[1] / ld[0], ld[2] / ld[0], 100 * ld[3] / ld[0]), flush=True)
    eng.ctl[tkr_hip.FLOW_CTL_PROF + 64:tkr_hip.FLOW_CTL_PROF + 68] = 0
    if os.environ.get('TKR_OWN_PROF') == '1' and v != 'f':          # a library built with -DTKR_OWN_PROF (TKR_HIP_LIB)
        pr = eng.ctl[tkr_hip.FLOW_CTL_PROF:tkr_hip.FLOW_CTL_PROF + 32].cpu().numpy().view(np.uint64).astype(np.float64)
        n, m = max(pr[6], 1), max(pr[13], 1)
        print('   item task (cycles; 2400 = 1 us): partners %.0f  own row %.0f  math %.0f  lds+acks %.0f  stores %.0f  next %.0f   (%d tasks, %.0f %% from LDS)'
              % (pr[0] / n, pr[1] / n, pr[2] / n, pr[3] / n, pr[4] / n, pr[5] / n, n, 100 * pr[7] / n), flush=True)
        print('   user task: between %.0f  rows %.0f  own %.0f  acks %.0f  stores %.0f   (%d tasks)'
              % (pr[8] / m, pr[9] / m, pr[10] / m, pr[11] / m, pr[12] / m, m), flush=True)
