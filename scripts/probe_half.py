"""K2o on a share of the CUs, ALONE: one engine with ranks_on_device = S runs whole epochs; us per batch.  usage: probe_half.py [S ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'top-k-rec_amd'))
import numpy as np, torch
import synth
from single import _engine
r = synth.make_ratings(seed=42, **dict(synth.ML10M))
row_ptr, pos, _, tr_users = synth.positives_csr(r)
n_users, n_items = r['n_users'], r['n_in'] + r['n_out']
dev = torch.device('cuda')
hp = dict(lu=2.5e-3, li=2.5e-3, lj=2.5e-4, lb=0.0, lr=1e-4, mode='l2')
csr = _engine.TrainingCSR.from_arrays(row_ptr, pos, np.asarray(tr_users, dtype=np.int32), dev)
for S in [int(x) for x in sys.argv[1:]] or [1, 2, 4]:
    e = _engine.BprEngine(n_users, n_items, 128, hp, dev, seed=1234)
    e.ranks_on_device = S
    e.prepare(256)
    nb = 3906
    e.run_batches(csr, nb, 256, want_loss=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        e.run_batches(csr, nb, 256, want_loss=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('share 1/%d: owners %d, layout %s, %.3f us per batch, %.1f M triplets/s' % (S, e._plan_owners(256), e.layout, dt / (2 * nb) * 1e6, 2 * nb * 256 / dt / 1e6))
    del e
