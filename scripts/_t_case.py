import os, sys
sys.path[:0] = ['/root/repo', '/root/repo/top-k-rec_amd', '/root/repo/tests']
import numpy as np, torch, tkr_hip
import test_gpu_topk as T
tkr_hip.set_topk_math(os.environ.get('MODE', 'bf16x3'))
n_rows, n_cols, k, K = 257, 95, 200, 30
rng = np.random.Generator(np.random.PCG64(n_rows * 7 + n_cols))
lim = max(1, int(np.sqrt((1 << 23) / k)) // 2)
U, V = T._exact(rng, n_rows, k, min(lim, 512)), T._exact(rng, n_cols, k, min(lim, 512))
b = (rng.integers(-64, 65, n_cols).astype(np.float32) / 64.0) if n_cols % 2 else None
rated = [rng.choice(n_cols, int(rng.integers(0, min(n_cols, 60))), replace=False).tolist() for _ in range(n_rows)]
rated[0] = list(range(n_cols)); rated[1] = list(range(max(0, n_cols - 3))); rated[2] = []
exp, s = T._oracle_lists(U, V, b, rated, K)
ids, scores = T._gpu_lists(tkr_hip, U, V, b, rated, K, want_scores=True)
ids = ids.cpu().numpy()
bad = [r for r in range(n_rows) if [int(c) for c in ids[r] if c >= 0] != exp[r]]
print('bad rows', bad[:20], len(bad))
for r in bad[:2]:
    print(r, [int(c) for c in ids[r]], exp[r], len(rated[r]))
sc = scores.cpu().numpy()
for r in bad[:3]:
    g = [int(c) for c in ids[r] if c >= 0]
    d = [(c, float(sc[r, q]), float(s[r, c])) for q, c in enumerate(g) if sc[r, q] != s[r, c]]
    print('row', r, 'wrong scores (col, gpu, true):', d, 'missing', sorted(set(exp[r]) - set(g)), 'extra', sorted(set(g) - set(exp[r])))
