"""Critical path of a K2f chain, from a -DTKR_FLOW_TRACE build (see csrc/bpr_flow.hip): batch size B, every user's only positive
is item 0, so item 0's task of batch t+1 waits for batch t's.  Prints where the per-batch period goes.
TKR_HIP_LIB=$PWD/_ab_libs/libtkr_trace.so python scripts/probe_flow_timeline.py [B]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'top-k-rec_amd')]
import numpy as np, torch
import tkr_hip
from single import _engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n_users = n_items = 100000
dev = torch.device('cuda', 0)
row_ptr = np.arange(n_users + 1, dtype=np.int64)
pos = np.zeros(n_users, dtype=np.int32)
csr = _engine.TrainingCSR.from_arrays(row_ptr, pos, np.arange(n_users, dtype=np.int32), dev)
hp = dict(lu=2.5e-3, li=2.5e-3, lj=2.5e-4, lb=0.0, lr=1e-4, mode='l2')
eng = _engine.BprEngine(n_users, n_items, 128, hp, dev, seed=5)
nb = 512
eng.run_batches(csr, nb, B, want_loss=False); torch.cuda.synchronize()          # one whole plan chunk: the traced call starts a new one
buf = torch.zeros(nb * 8, dtype=torch.int64, device=dev)
tkr_hip.lib().tkr_flow_trace_buffer(C.c_void_p(buf.data_ptr()))
torch.cuda.synchronize()
eng.run_batches(csr, nb, B, want_loss=False); torch.cuda.synchronize()
eng.check()
tkr_hip.lib().tkr_flow_trace_buffer(C.c_void_p(0))
raw = buf.cpu().numpy().reshape(nb, 8)
t = raw.astype(np.float64) * 0.01          # 100 MHz -> us
t = t[16:-4]
period = np.diff(t[:, 3])
print('B=%d: period %.2f us per batch (median %.2f)' % (B, period.mean(), np.median(period)))
print('  stores of batch t issued -> batch t+1 own row validated:           %.2f us' % np.median(t[1:, 4] - t[:-1, 3]))
print('  own row validated -> gradients done:                             %.2f us   (partner rows validated %.2f us before the own row)' % (np.median(t[:, 1] - t[:, 4]), np.median(t[:, 4] - t[:, 5])))
print('  gradients done -> readers acknowledged:                          %.2f us' % np.median(t[:, 2] - t[:, 1]))
print('  acknowledged -> stores issued:                                   %.2f us' % np.median(t[:, 3] - t[:, 2]))
print('  record in hand -> rows valid (how long the task was held):       %.2f us' % np.median(t[:, 1] - t[:, 0]))
