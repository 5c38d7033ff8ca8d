"""dev probe: K4 alone at a given shape (for rocprofv3 PMC passes)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'top-k-rec_amd')]
import numpy as np, torch, tkr_hip
n_users, n_items, k, K = (int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else (69878, 10380, 128, 30)))
reps = int(os.environ.get('REPS', 3))
g = torch.Generator(device='cuda'); g.manual_seed(1)
U = (torch.randn((n_users, k), device='cuda', generator=g) * 0.01)
V = (torch.randn((n_items, k), device='cuda', generator=g) * 0.01)
deg = 100
ptr = torch.arange(0, (n_users + 1) * deg, deg, dtype=torch.int64, device='cuda')
cols = torch.randint(0, n_items, (n_users * deg,), device='cuda', generator=g, dtype=torch.int32)
mask, pitch = tkr_hip.build_rated_mask(ptr, cols, n_users, n_items)
if os.environ.get('TKR_TOPK_MATH'):          # 'fp32' / 'bf16x3' / 'refine': the other score arithmetics (PMC passes of the fp32-MFMA kernel)
    tkr_hip.set_topk_math(os.environ['TKR_TOPK_MATH'])
tkr_hip.score_topk(U, V, K, mask=mask, mask_pitch=pitch)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    tkr_hip.score_topk(U, V, K, mask=mask, mask_pitch=pitch)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print('%d x %d k=%d K=%d: %.3f ms  %.1f TFLOP/s  %.2f M users/s' % (n_users, n_items, k, K, ms, 2.0 * k * n_items * n_users / ms / 1e9, n_users / ms / 1e3))
