"""timing probe of K2f at the headline shape: python scripts/probe_flow_bench.py [steps] [waves,waves,...] [B]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'top-k-rec_amd')]
import numpy as np, torch
import bench, tkr_hip
from single import _engine

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
waves = [int(x, 0) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else [8]     # 0xTTWW: tune bits TT, waves per CU WW
B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
dev = torch.device('cuda', 0)
r, csr, eng, nnz = bench.build_problem('ml10m', 128, 0, 1, dev)
for w in waves:
    eng.cfg.flow_waves_per_cu = w
    eng.run_batches(csr, 512, B, want_loss=False)
    torch.cuda.synchronize()
    eng.ctl[tkr_hip.FLOW_CTL_SPINS] = 0
    eng.ctl[tkr_hip.FLOW_CTL_PROF:tkr_hip.FLOW_CTL_PROF + 16] = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    eng.run_batches(csr, steps, B, want_loss=False)
    e1.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    eng.check()
    spins = int(eng.ctl[tkr_hip.FLOW_CTL_SPINS])
    print('B %d waves/CU 0x%04x: %.2f us/batch (events %.2f), %.1f M triplets/s, %.1f spin passes per task' %
          (B, w, wall / steps * 1e6, e0.elapsed_time(e1) * 1e3 / steps, steps * B / wall / 1e6, spins / (steps * 3.0 * B)), flush=True)
    if os.environ.get('TKR_FLOW_PROFILE') == '1':
        pr = eng.ctl[tkr_hip.FLOW_CTL_PROF:tkr_hip.FLOW_CTL_PROF + 16].cpu().numpy().view(np.uint64)
        tasks = max(int(pr[5]), 1)
        print('   cycles per task: grab %.0f  record %.0f  rows+math %.0f  war %.0f  finish %.0f   (tasks %d, idle slots %d; 2.4 GHz: 2400 cycles = 1 us)'
              % (pr[0] / tasks, pr[1] / tasks, pr[2] / tasks, pr[3] / tasks, pr[4] / tasks, tasks, int(pr[6])), flush=True)
        print('   acknowledge wait: %.0f %% of it in item tasks' % (100.0 * pr[7] / max(int(pr[3]), 1)), flush=True)
