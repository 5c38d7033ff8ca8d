"""K4 under its three score arithmetics at the two benchmark shapes: time per pass, agreement of 'refine' with 'fp32'.
python scripts/probe_topk_modes.py [ml10m|netflix|both] [mode,mode,...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'top-k-rec_amd')]
import numpy as np, torch
import tkr_hip
which = sys.argv[1] if len(sys.argv) > 1 else 'both'
modes = sys.argv[2].split(',') if len(sys.argv) > 2 else ['bf16x3', 'fp32', 'refine']
dev = torch.device('cuda', 0)
shapes = [('ml10m', 69878, 10380, 130), ('netflix', 480189, 17770, 150)]
for name, n_users, n_items, deg in shapes:
    if which not in ('both', name):
        continue
    k, K = 128, 30
    g = torch.Generator(device=dev); g.manual_seed(11)
    U = (torch.randn((n_users, k), device=dev, generator=g) * 0.01 * 1e6).round() / 1e6
    V = (torch.randn((n_items, k), device=dev, generator=g) * 0.01 * 1e6).round() / 1e6
    ptr = torch.arange(0, (n_users + 1) * deg, deg, dtype=torch.int64, device=dev)
    cols = torch.randint(0, n_items, (n_users * deg,), device=dev, generator=g, dtype=torch.int32)
    mask, pitch = tkr_hip.build_rated_mask(ptr, cols, n_users, n_items)
    out = {}
    for mode in modes:
        tkr_hip.set_topk_math(mode)
        out[mode] = tkr_hip.score_topk(U, V, K, mask=mask, mask_pitch=pitch, want_scores=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            tkr_hip.score_topk(U, V, K, mask=mask, mask_pitch=pitch)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        print('%-8s %-7s %7.2f ms per pass = %5.1f M users/s' % (name, mode, ms, n_users / ms / 1e3), flush=True)
    if len(modes) < 3:
        continue
    same_ids = torch.equal(out['refine'][0], out['fp32'][0])
    same_sc = torch.equal(out['refine'][1].view(torch.int32), out['fp32'][1].view(torch.int32))
    print('%-8s refine == fp32: ids %s, score bits %s;  bf16x3 ids equal fp32 in %.2f %% of rows' %
          (name, same_ids, same_sc, 100.0 * (out['bf16x3'][0] == out['fp32'][0]).all(1).float().mean().item()), flush=True)
    tkr_hip.set_topk_math('bf16x3')
