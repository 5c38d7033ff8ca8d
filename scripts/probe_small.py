"""dev probe: K2 launch time vs table footprint at B=256 (is the gather level footprint/TLB-bound?)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'top-k-rec_amd')]
import numpy as np, torch, tkr_hip
from single import _engine
dev = torch.device('cuda', 0)
hp = dict(lu=2.5e-3, li=2.5e-3, lj=2.5e-4, lb=0.0, lr=1e-4, mode='l2')
B, k, nb = 256, 128, 2048
for n_users, n_items in ((2000, 1500), (20000, 5000), (69878, 10380), (480189, 17770), (2000000, 100000)):
    rng = np.random.Generator(np.random.PCG64(0))
    deg = 30
    row_ptr = np.arange(0, (n_users + 1) * deg, deg, dtype=np.int64)
    pos = rng.integers(0, n_items, n_users * deg).astype(np.int32)
    csr = _engine.TrainingCSR.from_arrays(row_ptr, pos, np.arange(n_users, dtype=np.int32), dev)
    eng = _engine.BprEngine(n_users, n_items, k, hp, dev, seed=1)
    eng.run_batches(csr, nb, B, want_loss=False)
    torch.cuda.synchronize()
    eng.step_events = []
    eng.run_batches(csr, nb, B, want_loss=False)
    torch.cuda.synchronize()
    ts = sum(a.elapsed_time(b) for a, b, _ in eng.step_events)
    hd = eng.plan.hdr.view(-1, 4)[:64].float().mean(0).tolist()
    mb = (n_users + n_items) * k * 4 * 4 / 1e6
    print('%8d users %7d items tables %8.1f MB: %.2f us/launch  hdr %s' % (n_users, n_items, mb, ts / nb * 1e3, [round(x, 1) for x in hd]), flush=True)
    del eng, csr
