"""K1 alone at a mid-size batch (no steps): for rocprofv3 --kernel-trace --stats.  usage: probe_mid.py [B] [n_batches] [calls]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'top-k-rec_amd'))
import numpy as np, torch
import synth, tkr_hip
from single import _engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 64
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 10
r = synth.make_ratings(seed=42, **dict(synth.ML10M))
row_ptr, pos, _, tr_users = synth.positives_csr(r)
n_users, n_items = r['n_users'], r['n_in'] + r['n_out']
dev = torch.device('cuda')
csr = _engine.TrainingCSR.from_arrays(row_ptr, pos, np.asarray(tr_users, dtype=np.int32), dev)
cnt = _engine.UpdateCounters(n_users, n_items, dev)
plan = _engine.PlanBuffers(nb, B, dev)
for c in range(calls):
    tkr_hip.sample_plan(csr, n_users, n_items, 7, c * nb * B, nb, B, cnt, plan)
torch.cuda.synchronize()
print('ok', int(cnt.ucnt.sum()), int(cnt.icnt.sum()))
if hasattr(tkr_hip.lib(), 'tkr_debug_mid_prof'):
    import ctypes as C
    out = (C.c_ulonglong * 32)()
    tkr_hip.lib().tkr_debug_mid_prof(out)
    names = ['sums', 'count', 'scan+tasks', 'fill', 'thread sort', 'wave sort', 'huge', 'occ out']
    for kind, off in (('user', 0), ('item', 16)):
        tot = sum(out[off:off + 8]) or 1
        print(kind, ' '.join('%s %.1f%%' % (n, 100.0 * out[off + i] / tot) for i, n in enumerate(names)), 'cycles per wg-call', tot)
