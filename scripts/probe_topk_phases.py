"""Cycle budget of the K4 tile loop by phase, from a TKR_ABL=256 build (scripts/ablate_topk.sh 256):
TKR_HIP_LIB=$PWD/_ab_libs/libtkr_abl256.so python scripts/probe_topk_phases.py [netflix|ml10m] [mode]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'top-k-rec_amd')]
import numpy as np, torch
import tkr_hip
name = sys.argv[1] if len(sys.argv) > 1 else 'netflix'
mode = sys.argv[2] if len(sys.argv) > 2 else 'refine'
n_users, n_items, deg = dict(ml10m=(69878, 10380, 130), netflix=(480189, 17770, 150))[name]
dev = torch.device('cuda', 0)
k, K = 128, 30
g = torch.Generator(device=dev); g.manual_seed(11)
U = (torch.randn((n_users, k), device=dev, generator=g) * 0.01 * 1e6).round() / 1e6
V = (torch.randn((n_items, k), device=dev, generator=g) * 0.01 * 1e6).round() / 1e6
ptr = torch.arange(0, (n_users + 1) * deg, deg, dtype=torch.int64, device=dev)
cols = torch.randint(0, n_items, (n_users * deg,), device=dev, generator=g, dtype=torch.int32)
mask, pitch = tkr_hip.build_rated_mask(ptr, cols, n_users, n_items)
tkr_hip.set_topk_math(mode)
tkr_hip.score_topk(U, V, K, mask=mask, mask_pitch=pitch); torch.cuda.synchronize()
out = (C.c_ulonglong * 8)()
tkr_hip.lib().tkr_k4_prof_read(out)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); tkr_hip.score_topk(U, V, K, mask=mask, mask_pitch=pitch); e1.record(); torch.cuda.synchronize()
tkr_hip.lib().tkr_k4_prof_read(out)
p = np.array(list(out), dtype=np.float64)
tiles = max(p[7], 1)
names = ['mfma+bias', 'staging', 'barrier', 'sched trims', 'filter', 'prologue', 'final stage']
print('%s %s: %.2f ms; %d wave-tiles; 100 MHz ticks per wave-tile (x24 = 2.4 GHz cycles):' % (name, mode, e0.elapsed_time(e1), tiles))
tot = p[:7].sum()
for i, n in enumerate(names):
    print('  %-12s %8.2f ticks per wave-tile  %5.1f %%' % (n, p[i] / tiles, 100 * p[i] / tot))
