"""dev probe: K3 (VBPR) per-kernel timing at ML-10M shape, dense d=20000."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'top-k-rec_amd')]
import numpy as np, torch, synth, tkr_hip
from single import _engine
k, d, B, nb = 128, int(os.environ.get('D', 20000)), int(os.environ.get('B', 256)), int(os.environ.get('NB', 256))
r = synth.make_ratings(**synth.ML10M, seed=42)
row_ptr, pos, srt, tr_users = synth.positives_csr(r)
dev = torch.device('cuda', 0)
n_users, n_items = r['n_users'], r['n_in'] + r['n_out']
csr = _engine.TrainingCSR.from_arrays(row_ptr, pos, tr_users, dev)
g = torch.Generator(device=dev); g.manual_seed(7)
feat = torch.zeros((n_items, d), device=dev)
if os.environ.get('DENSE') == '1':          # the literal "d = 128" reading of BASELINE.json configs[2]: a narrow DENSE feat
    feat = torch.rand((n_items, d), device=dev, generator=g) + 0.1
else:
    cols = torch.randint(0, d, (n_items, 100), device=dev, generator=g)
    feat.scatter_(1, cols, torch.rand((n_items, 100), device=dev, generator=g) + 0.1)
feat /= feat.norm(dim=1, keepdim=True)
hp = dict(lu=2.5e-3, li=2.5e-3, lj=2.5e-4, lb=0.0, le=0.0, lr=1e-4, mode='l2')
eng = _engine.VbprEngine(n_users, n_items, k, d, feat, hp, dev, seed=3, sparse=False if os.environ.get('VIEW') == 'dense' else None)   # VIEW=dense: the fp32-MFMA kernels
eng.run_batches(csr, nb, B, want_loss=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
eng.run_batches(csr, nb, B, want_loss=False)
torch.cuda.synchronize()
print('B=%d d=%d cols=%s cpb=%s: %.1f us/batch' % (B, d, eng.wants_cols(B), eng.cols_per_block(B) if eng.sparse is not None else None,
                                                   (time.perf_counter() - t0) / nb * 1e6))
if os.environ.get('LOSS') == '1':           # what does the per-batch loss (vbpr.py:114 returns obj every batch) cost?
    for wl in (True, False, True, False):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.run_batches(csr, nb, B, want_loss=wl)
        torch.cuda.synchronize()
        print('  want_loss %s: %.1f us/batch' % (wl, (time.perf_counter() - t0) / nb * 1e6), flush=True)
if os.environ.get('GRAPH') == '1':          # the same step launches replayed from a captured graph: is the loop host-bound?
    plan = eng.plan
    step = eng.step_fn(B)
    gnb = min(nb, eng._cap(B))
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        step(plan, 0, gnb, None)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            step(plan, 0, gnb, None)
        graph.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            graph.replay()
        torch.cuda.synchronize()
        print('graph replay of %d batches: %.1f us/batch' % (gnb, (time.perf_counter() - t0) / (4 * gnb) * 1e6))
