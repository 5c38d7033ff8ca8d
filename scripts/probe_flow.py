"""debug / timing probe of the persistent dataflow step (K2f): python scripts/probe_flow.py [nb] [B] [k]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'top-k-rec_amd'), os.path.join(ROOT, 'tests')]
import numpy as np, torch
import tkr_hip
from oracle import plan_np as P, ref_np as R
import test_gpu_flow as T

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
k = int(sys.argv[3]) if len(sys.argv) > 3 else 16
n_users, n_items = 400, 120
tr, tr_users = T._toy(n_users, n_items, seed=k + B)
rng = np.random.Generator(np.random.PCG64(k))
ref = R.init_bpr_state(n_users, n_items, k, rng)
hp = dict(lu=2.5e-3, li=2.5e-3, lj=2.5e-4, lb=1e-3, lr=0.05, mode='l2')
F = T._Flow(tkr_hip, ref, n_users, n_items, k, hp)
plan, exp, _, _ = T._plan(tkr_hip, tr, tr_users, n_users, n_items, 42, 0, nb, B)
t0 = time.time()
F.run(plan, B, nb, None)
st, ctl = F.status()
print('nb', nb, 'B', B, 'k', k, 'status', st, 'time %.3f s' % (time.time() - t0), 'spins', ctl[tkr_hip.FLOW_CTL_SPINS])
print('debug', ctl[tkr_hip.FLOW_CTL_DEBUG:tkr_hip.FLOW_CTL_DEBUG + 16].tolist())
print('heads', ctl[0:1024:32].tolist(), 'arrive/exit', ctl[1024:1026].tolist())
if st == 0:
    ucnt, icnt, uocc, iocc, _ = T._oracle(ref, exp, n_users, n_items, nb, B, hp)
    T._check(F, ref, ucnt, icnt, uocc, iocc)
    print('parity ok')
