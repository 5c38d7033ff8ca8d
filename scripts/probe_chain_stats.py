"""dev probe: how many tasks of a 256-batch have their row updated again in the very next batch (ML-10M shape)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'top-k-rec_amd')]
import torch, bench
dev = torch.device('cuda', 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
r, csr, eng, nnz = bench.build_problem('ml10m', 128, 0, 1, dev)
eng.prepare(B)
ch = eng._next_chunk(csr, B, 512)
buf = eng.pipe.acquire(ch.idx)
torch.cuda.synchronize()
R = buf.prec[: ch.nb * 3 * B * 32].view(ch.nb, 3 * B, 32)
used = (R[:, :, 0] != -1)
key = R[:, :, 0].long() & 0xffffffff                     # users ascending, then items (bit 31), unused slots last
nxt, cur = key[1:].contiguous(), key[:-1].contiguous()
pos = torch.searchsorted(nxt, cur).clamp_(max=3 * B - 1)
go = (nxt.gather(1, pos) == cur) & (cur != 0xffffffff)
item = R[:-1, :, 0] < 0
print('tasks per batch %.1f, with a task of the same row in the next batch %.1f (items %.1f, users %.1f)' % (
    used.float().sum(1).mean(), go.float().sum(1).mean(), (go & item).float().sum(1).mean(), (go & ~item).float().sum(1).mean()))
# chain lengths: rows present in EVERY batch
keys = R[:, :, 0]
first = set(keys[0][used[0]].tolist())
for b in range(1, ch.nb):
    first &= set(keys[b][used[b]].tolist())
print('rows in all %d batches: %d' % (ch.nb, len(first)))
