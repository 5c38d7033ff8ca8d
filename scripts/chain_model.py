"""How many cross-CU hand-offs per batch does the dependency structure of the BPR step FORCE?  (CPU only; the numbers behind
DESIGN.md §4 K2o "what bounds it".)

For a stream of batches drawn by the oracle's sampler at a benchmark shape, every task (one per touched row and batch) gets a
finish time = max over its producers (finish + hop) + work, with hop = 1 for a hand-off through memory and 0 for a hand-off
inside a workgroup's LDS, work = `work` hops.  The longest path divided by the number of batches is the number of memory hand-offs
per batch no amount of parallelism can remove:
  K2f            own row: memory; partner rows (user, other item): memory
  K2o row-read   own ITEM row: LDS (owned); everything else: memory
  K2o scalar     own item row: LDS; user rows: memory; the other item only through its scalar of the SAME batch (memory), which its
                 task can publish once it has ITS own row and user row
python scripts/chain_model.py [shape] [batches] [B] [work]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'top-k-rec_amd')]
import numpy as np
import synth
from oracle import plan_np as P

shape = sys.argv[1] if len(sys.argv) > 1 else 'ml10m'
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 400
B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
work = float(sys.argv[4]) if len(sys.argv) > 4 else 0.15
r = synth.make_ratings(seed=42, **dict(synth.ML10M if shape == 'ml10m' else synth.NETFLIX))
row_ptr, pos, srt, tr_users = synth.positives_csr(r)
n_users, n_items = r['n_users'], r['n_in'] + r['n_out']
u, i, j = P.sample_triplets(np.asarray(tr_users), row_ptr, pos, srt, n_items, 1234, 0, nb * B)


def run(form):
    fu, fi = np.zeros(n_users), np.zeros(n_items)          # finish time of the last update of every row
    end = []
    for b in range(nb):
        sl = slice(b * B, (b + 1) * B)
        ub, ib, jb = u[sl], i[sl], j[sl]
        own_hop_item = 1.0 if form == 'k2f' else 0.0
        if form != 'scalar':
            # a user task: own row + both item rows through memory
            tu = np.maximum(fu[ub] + 1.0, np.maximum(fi[ib], fi[jb]) + 1.0)
            # an item task: own row (memory or LDS), the user row and the OTHER item's row through memory
            ti = np.maximum(np.maximum(fi[ib] + own_hop_item, fu[ub] + 1.0), fi[jb] + 1.0)
            tj = np.maximum(np.maximum(fi[jb] + own_hop_item, fu[ub] + 1.0), fi[ib] + 1.0)
        else:
            tu = np.maximum(fu[ub] + 1.0, np.maximum(fi[ib], fi[jb]) + 1.0)
            # the scalar of a role is out once its task has its own row (LDS) and the user row (memory), + the dots
            ready_i = np.maximum(fi[ib], fu[ub] + 1.0)
            ready_j = np.maximum(fi[jb], fu[ub] + 1.0)
            # a row's task computes ALL its scalars when the row and the LAST of its user rows are there
            ri, rj = np.zeros(n_items), np.zeros(n_items)
            np.maximum.at(ri, ib, ready_i); np.maximum.at(ri, jb, ready_j)
            dots = ri + work * 0.5
            ti = np.maximum(dots[ib], dots[jb] + 1.0)          # own scalar, the partner's through memory
            tj = np.maximum(dots[jb], dots[ib] + 1.0)
        nfu, nfi = fu.copy(), fi.copy()
        # a row's task finishes when the last of its occurrences could be served, + its work
        tmp = np.zeros(n_users); np.maximum.at(tmp, ub, tu); m = tmp > 0; nfu[m] = tmp[m] + work
        tmp = np.zeros(n_items); np.maximum.at(tmp, ib, ti); np.maximum.at(tmp, jb, tj); m = tmp > 0; nfi[m] = tmp[m] + work
        fu, fi = nfu, nfi
        end.append(max(fu.max(), fi.max()))
    end = np.array(end)
    half = nb // 2
    return (end[-1] - end[half]) / (nb - 1 - half)


print('%s shape, batch %d, %d batches, work per task %.2f hops' % (shape, B, nb, work))
for form, name in (('k2f', 'K2f (own rows through memory)'), ('rowread', 'K2o, item tasks read rows'), ('scalar', 'K2o, scalar exchange')):
    per = run(form)
    print('  %-32s critical path %.2f per batch = %.2f memory hand-offs + work' % (name, per, per))
