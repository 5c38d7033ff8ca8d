"""dev tool: one parity case of tests/test_gpu_flow.py through K2o, printing the post-mortem words when a spin runs out"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'top-k-rec_amd'), os.path.join(ROOT, 'tests')]
import numpy as np, torch
import tkr_hip
import test_gpu_flow as TF
from oracle import ref_np as R
k, B, nb, owners = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
n_users, n_items = 400, 120
tr, tr_users = TF._toy(n_users, n_items, seed=k + B)
rng = np.random.Generator(np.random.PCG64(k))
ref = R.init_bpr_state(n_users, n_items, k, rng)
hp = dict(lu=2.5e-3, li=2.5e-3, lj=2.5e-4, lb=1e-3, lr=0.05, mode='l2')
F = TF._Flow(tkr_hip, ref, n_users, n_items, k, hp)
plan, exp, _, _ = TF._plan(tkr_hip, tr, tr_users, n_users, n_items, 42, 0, nb, B, owners=owners)
F.run(plan, B, nb, None, waves_per_cu=int(sys.argv[5]) if len(sys.argv) > 5 else 0)
status, ctl = F.status()
print('k', k, 'B', B, 'nb', nb, 'status', status, 'debug', ctl[tkr_hip.FLOW_CTL_DEBUG:tkr_hip.FLOW_CTL_DEBUG + 16].tolist(), 'spins', ctl[tkr_hip.FLOW_CTL_SPINS], 'bufs', F.bufs)
