import os, sys
ROOT='/root/repo'
sys.path[:0] = [ROOT, os.path.join(ROOT, 'top-k-rec_amd')]
import numpy as np, torch
import bench, tkr_hip
from single import _engine
dev = torch.device('cuda', 0)
r, csr, eng, nnz = bench.build_problem('ml10m', 128, 0, 1, dev)
eng.run_batches(csr, 512, 256, want_loss=False); torch.cuda.synchronize()
plan = eng.plan
prec = plan.prec.cpu().numpy().reshape(-1, 32)        # 512*768 records of 32 ints
rowk, ver, nocc = prec[:, 0], prec[:, 1], prec[:, 2]
used = rowk != -1
item = used & (rowk < 0)
print('records', len(rowk), 'used', used.sum(), 'item tasks', item.sum())
b = prec[:, 4]
nb = 512
mx = np.zeros(nb, int)
for t in range(nb):
    sl = slice(t*768, (t+1)*768)
    m = item[sl]
    mx[t] = nocc[sl][m].max()
print('max occurrences of an item task per batch: mean %.1f, >4: %.2f, >8: %.2f, >16: %.2f' % (mx.mean(), (mx>4).mean(), (mx>8).mean(), (mx>16).mean()))
cnt = np.bincount(nocc[item].clip(0, 40))
print('item tasks by occurrences:', {i:int(c) for i,c in enumerate(cnt) if c})
print('tasks with >8 per batch: %.2f' % ((nocc[item] > 8).sum() / nb))
