"""Which fp32 dot product is K4's fp32-MFMA arithmetic (v_mfma_f32_32x32x2_f32 over the two k-halves)?  Compares the scores
the kernel returns with candidate summation orders emulated on the host (products exact in float64, one rounding per step).
python scripts/probe_mfma_order.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'top-k-rec_amd')]
import numpy as np, torch
import tkr_hip
rng = np.random.Generator(np.random.PCG64(3))
f32 = np.float32


def fma(a, b, c):           # fl32(a*b + c): a*b is exact in float64; the sum is rounded twice (53 then 24 bits) -- rarely matters
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)


for k in (128, 64, 100, 50):
    n_rows, n_cols, K = 64, 320, 30
    U = (rng.standard_normal((n_rows, k)) * 0.01).astype(f32)
    V = (rng.standard_normal((n_cols, k)) * 0.01).astype(f32)
    tkr_hip.set_topk_math('fp32')
    ids, sc = tkr_hip.score_topk(torch.from_numpy(U).cuda(), torch.from_numpy(V).cuda(), K, want_scores=True)
    ids, sc = ids.cpu().numpy(), sc.cpu().numpy()
    KH = (k + 1) // 2
    uu, vv = U[:, None, :].repeat(K, 1), V[ids]                       # [rows, K, k]
    res = {}
    acc = np.zeros((n_rows, K), f32)
    for kk in range(KH):                                             # (A) fma chain, half 0 then half 1 per step
        acc = fma(uu[..., kk], vv[..., kk], acc)
        if KH + kk < k:
            acc = fma(uu[..., KH + kk], vv[..., KH + kk], acc)
    res['chain h0,h1'] = acc
    acc = np.zeros((n_rows, K), f32)
    for kk in range(KH):                                             # (A') half 1 first
        if KH + kk < k:
            acc = fma(uu[..., KH + kk], vv[..., KH + kk], acc)
        acc = fma(uu[..., kk], vv[..., kk], acc)
    res['chain h1,h0'] = acc
    acc = np.zeros((n_rows, K), f32)
    for kk in range(KH):                                             # (B) both products exact, one rounding per MFMA
        t = uu[..., kk].astype(np.float64) * vv[..., kk] + acc.astype(np.float64)
        if KH + kk < k:
            t = t + uu[..., KH + kk].astype(np.float64) * vv[..., KH + kk]
        acc = t.astype(f32)
    res['one rounding per mfma'] = acc
    acc = np.zeros((n_rows, K), f32)
    for kk in range(k):
        acc = fma(uu[..., kk], vv[..., kk], acc)
    res['chain k ascending'] = acc
    for name, a in res.items():
        print('k %3d  %-24s bitwise equal: %6.2f %%   max |diff| %.3g' % (k, name, 100.0 * np.mean(a.view(np.int32) == sc.view(np.int32)), np.abs(a - sc).max()), flush=True)
