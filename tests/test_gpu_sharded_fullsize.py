"""-m gpu: the SHARDED forms of BASELINE.json's configs 4 and 5 at full shape (nothing sharded had run above 3,000
users), two ranks on the one visible GPU through the gloo hook (the 8-GPU run uses RCCL on the same code), plus a 2-rank VBPR
run against an oracle simulation of both shards.

  * BPR.train, one reference epoch at the Netflix shape (480,189 users x 17,770 items, k = 128, batch 256), users sharded
    (single/bpr.py:136-147 is the loop each rank runs on its shard): sampler invariants on every triplet of the rank, the
    device stream equal to the oracle's, the owned-row gather covers every training user exactly once, the replicas are
    bit-identical after the exchange, each rank's user rows against the oracle replay of its own stream, the item tables
    against V0 + the sum of both ranks' oracle deltas.
  * the sharded ranking of evaluate.py (:84-112 sharded by test line) at 480,189 x 17,770: each rank's block of id lists
    identical to the same rows of the single-process ranking, the all-reduced accuracy identical to the single-process one.
"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'top-k-rec_amd')

_HEAD = r'''
import os, sys
sys.path[:0] = [%(root)r, %(pkg)r]
import numpy as np, torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group('gloo')
rank, world = dist.get_rank(), dist.get_world_size()
'''

_TRAIN_NF = _HEAD + r'''
import synth
import dist as tdist
from single import BPR
from oracle import plan_np as P, ref_np as R
n_users, n_items, k, B, limit, seed, lr = 480189, 17770, 128, 256, 10 ** 6, 11, 1e-3
row_ptr, pos, srt, tr_users = synth.train_csr_shape(n_users, n_items, mean_pos=90.0, seed=43)
m = BPR(k=k, lr=lr)
m.n_users, m.n_items = n_users, n_items
m.tr_data = {int(u): pos[row_ptr[u]:row_ptr[u + 1]] for u in tr_users}        # what load_training_data builds from the text (bpr.py:63-65)
m.tr_users = [int(u) for u in tr_users]
m._csr_arrays = (row_ptr.astype(np.int64), pos, tr_users)
rng = np.random.Generator(np.random.PCG64(0))
init = [(rng.standard_normal((n_users, k), dtype=np.float32) * 0.01), (rng.standard_normal((n_items, k), dtype=np.float32) * 0.01),
        np.zeros((n_items, 1), np.float32)]
m.fue, m.fie, m.fib = (a.copy() for a in init)
m.train(epochs=1, batch_size=B, epoch_sample_limit=limit, seed=seed, verbose=False)
eng = m._eng
assert eng.layout == 'flow' and eng.n_users == len(m._owned) == len(tdist.shard_users(m.tr_users, rank, world))
want = 0 if os.environ.get('TKR_OWN') == '0' else torch.cuda.get_device_properties(0).multi_processor_count // world
assert eng.ranks_on_device == world and eng.plan.owners == want, (eng.plan.owners, want)      # the step that ran: K2o on half the CUs, or K2f
# (1) replicas bit-identical after the exchange
tdist.assert_replicated(eng)
# (2) the owned-row gather covers every training user exactly once
parts = tdist.gather_owned_rows(m._owned, *eng.get('U'))
ids = np.concatenate([p[0] for p in parts])
assert len(ids) == len(tr_users) and np.array_equal(np.sort(ids), np.sort(tr_users.astype(np.int64)))
# (3) this rank's stream: the oracle sampler on the shard's users, invariants on every triplet
nb = (limit // B) // world
users = tdist.shard_users(m.tr_users, rank, world)
u, i, j = P.sample_triplets(users, row_ptr, pos, srt, n_items, seed, rank * 1 * nb * B, nb * B)
key_pos = np.repeat(np.arange(n_users, dtype=np.int64), np.diff(row_ptr)) * n_items + srt
def member(uu, cc):
    q = uu.astype(np.int64) * n_items + cc
    at = np.minimum(np.searchsorted(key_pos, q), len(key_pos) - 1)
    return key_pos[at] == q
assert member(u, i).all() and not member(u, j).any() and np.isin(u, np.asarray(users)).all()
last = nb %% 512 or 512                                           # the plan of the last chunk is still on the device
dev_u = m._owned[eng.plan.u.cpu().numpy()[: last * B]]
assert np.array_equal(dev_u, u[(nb - last) * B:]) and np.array_equal(eng.plan.i.cpu().numpy()[: last * B], i[(nb - last) * B:])
assert np.array_equal(eng.plan.j.cpu().numpy()[: last * B], j[(nb - last) * B:])
# (4) the oracle replays this rank's epoch from the common start; the exchange is V0 + sum of deltas, slots averaged
hp = dict(lu=m.lu, li=m.li, lj=m.lj, lb=m.lb, lr=lr, mode='l2')
st = dict(U=init[0].copy(), V=init[1].copy(), b=init[2].ravel().copy(), msU=np.ones_like(init[0]), msV=np.ones_like(init[1]),
          msb=np.ones(n_items, np.float32))
for s in range(nb):
    R.bpr_step(st, u[s * B:(s + 1) * B], i[s * B:(s + 1) * B], j[s * B:(s + 1) * B], hp)
mine = np.asarray(users)
np.testing.assert_allclose(m.fue[mine], st['U'][mine], rtol=2e-4, atol=1e-6)
others = np.setdiff1d(np.arange(n_users), tr_users)
np.testing.assert_array_equal(m.fue[others], init[0][others])     # users nobody trains keep their start
dV, db = torch.from_numpy(st['V'] - init[1]), torch.from_numpy(st['b'] - init[2].ravel())
dist.all_reduce(dV); dist.all_reduce(db)
np.testing.assert_allclose(m.fie, init[1] + dV.numpy(), rtol=2e-4, atol=1e-6)
np.testing.assert_allclose(m.fib.ravel(), init[2].ravel() + db.numpy(), rtol=2e-4, atol=1e-6)
assert np.abs(m.fie - init[1]).max() > 1e-4                       # and it did train
# user rows of the OTHER rank arrive through the gather: their checksum must equal what that rank computed
chk = torch.tensor([float(np.abs(m.fue).sum(dtype=np.float64))], dtype=torch.float64)
lo, hi = chk.clone(), chk.clone()
dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
assert torch.equal(lo, hi)
dist.barrier(); dist.destroy_process_group()
print('ok', rank)
'''

_SCORE_NF = _HEAD + r'''
import evaluate as E
import tkr_hip
n_users, n_items, k, total, step, deg, nlike = 480189, 17770, 128, 30, 5, 60, 3
dev = torch.device('cuda', 0)
rng = np.random.Generator(np.random.PCG64(5))
U = np.round(rng.standard_normal((n_users, k), dtype=np.float32) * 0.01, 6)
V = np.round(rng.standard_normal((n_items, k), dtype=np.float32) * 0.01, 6)
bias = np.round(rng.standard_normal(n_items).astype(np.float32) * 0.001, 6)
def rows_csr(per_row):
    key = np.unique(np.repeat(np.arange(n_users, dtype=np.int64), per_row) * n_items + rng.integers(0, n_items, n_users * per_row))
    ptr = np.zeros(n_users + 1, dtype=np.int64)
    np.cumsum(np.bincount(key // n_items, minlength=n_users), out=ptr[1:])
    return ptr, (key %% n_items).astype(np.int32)
rated_ptr, rated_cols = rows_csr(deg)
like_ptr, like_cols = rows_csr(nlike)
ids_tab = {str(c): c for c in range(n_items)}
full = E.Scenario(ids_tab, np.arange(n_users, dtype=np.int64), like_ptr, like_cols, rated_ptr, rated_cols)
Ud = torch.from_numpy(U).to(dev)
# the single-process ranking of every test line ...
ids_full = E.rank_scenario(Ud, V, bias, ids_tab, full, total, dev)
interval = total // step
hits_full = tkr_hip.count_hits(ids_full, torch.from_numpy(like_ptr).to(dev), torch.from_numpy(like_cols).to(dev), step, interval).cpu().numpy()
single = [float(h) / full.tcount for h in hits_full]
# ... against this rank's block of the sharded run (evaluate.py under a launcher: shard_scenario + one all-reduce of interval + 1 integers)
mine = E.shard_scenario(full, rank, world)
lo, hi = rank * n_users // world, (rank + 1) * n_users // world
assert len(mine.users) == hi - lo and mine.tcount == int(like_ptr[hi] - like_ptr[lo])
ids_mine = E.rank_scenario(Ud, V, bias, ids_tab, mine, total, dev)
assert torch.equal(ids_mine, ids_full[lo:hi])                       # id lists identical to the single-process run, all 30 per row
assert bool((ids_mine >= 0).all())
sharded = E.evaluate_loaded(Ud, V, bias, ids_tab, full, step, total, dev)
assert sharded == single, (sharded, single)                         # integers reduced, one division: bit-identical
assert single[-1] > single[0] > 0.0
dist.barrier(); dist.destroy_process_group()
print('ok', rank)
'''

_VBPR = _HEAD + r'''
from single import VBPR
import dist as tdist
from oracle import plan_np as P, ref_np as R
data = %(data)r
k, d, B, epochs, limit, lr = 16, 40, 32, 2, 32 * 12, 0.02
kh = k // 2
m = VBPR(k=k, d=d, lr=lr, lambda_b=1e-3, lambda_e=1e-3)
m.load_training_data(data + '/uid', data + '/vid', data + '/f0tr.txt')
rng = np.random.Generator(np.random.PCG64(0))
feat = np.abs(rng.standard_normal((m.n_items, d))).astype(np.float32) * (rng.random((m.n_items, d)) < 0.3)
feat = (feat / np.maximum(np.linalg.norm(feat, axis=1, keepdims=True), 1e-6)).astype(np.float32)
m.feat = feat
init = [(rng.standard_normal((m.n_users, k)) * 0.1).astype(np.float32), (rng.standard_normal((m.n_items, k)) * 0.1).astype(np.float32),
        (rng.standard_normal((m.n_items, 1)) * 0.01).astype(np.float32)]
m.fue, m.fie, m.fib = (a.copy() for a in init)
m.train(epochs=epochs, batch_size=B, epoch_sample_limit=limit, seed=11, verbose=False)
# ---- oracle simulation of BOTH ranks (single/vbpr.py:50-73 per shard + the per-epoch exchange of I, irb, cem, icb)
hp = dict(lu=m.lu, li=m.li, lj=m.lj, lb=m.lb, le=m.le, lr=lr, mode='l2')
nb = (limit // B) // world
row_ptr, pos, srt = P.build_csr(m.tr_data, m.n_users)
def fresh():
    s = dict(ure=init[0][:, :kh].copy(), uce=init[0][:, kh:2 * kh].copy(), ire=init[1][:, :kh].copy(), irb=init[2].ravel().copy(),
             cem=np.full((d, kh), 2.0 / (d * k), np.float32), icb=np.zeros(d, np.float32))
    for n in list(s):
        s['ms_' + n] = np.ones_like(s[n])
    return s
st = [fresh() for _ in range(world)]
drawn = [r * epochs * nb * B for r in range(world)]
shared = ('ire', 'irb', 'cem', 'icb')
for e in range(epochs):
    start = {n: st[0][n].copy() for n in shared}
    for r in range(world):
        users = tdist.shard_users(m.tr_users, r, world)
        u, i, j = P.sample_triplets(users, row_ptr, pos, srt, m.n_items, 11, drawn[r], nb * B)
        drawn[r] += nb * B
        for s in range(nb):
            R.vbpr_step(st[r], feat, u[s*B:(s+1)*B], i[s*B:(s+1)*B], j[s*B:(s+1)*B], hp)
    for n in shared:
        new = start[n] + sum(x[n] - start[n] for x in st)
        ms = sum(x['ms_' + n] for x in st) / world
        for x in st:
            x[n], x['ms_' + n] = new.copy(), ms.copy()
U0 = np.concatenate([init[0][:, :kh], init[0][:, kh:2 * kh]], 1)
U = U0 + sum(np.concatenate([x['ure'], x['uce']], 1) - U0 for x in st)
tol = dict(rtol=3e-4, atol=2e-5)
np.testing.assert_allclose(m.fue, U, **tol)
np.testing.assert_allclose(m.fie, np.concatenate([st[0]['ire'], feat @ st[0]['cem']], 1), **tol)          # vbpr.py:124-126
np.testing.assert_allclose(m.fib.ravel(), st[0]['irb'] + feat @ st[0]['icb'], **tol)
tdist.assert_replicated(m._eng)
dist.barrier(); dist.destroy_process_group()
print('ok', rank)
'''


def _launch(script_text, tmp_path, port, world=2, timeout=900, env=None, **fmt):
    script = tmp_path / 'worker.py'
    script.write_text(script_text % dict(root=ROOT, pkg=PKG, **fmt))
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=%d' % world,
                          '--master-addr', '127.0.0.1', '--master-port', str(port), str(script)],
                         capture_output=True, text=True, timeout=timeout, env=dict(os.environ, **(env or {})))
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-4000:]
    assert out.stdout.count('ok') == world


@pytest.mark.parametrize('own', ['1', '0'])
def test_netflix_shape_sharded_epoch_two_ranks(tmp_path, own):
    """BASELINE.json configs[3], sharded: BPR.train over two user shards, one reference epoch (10^6 // 256 = 3906 batches, 1953 per rank);
    through K2o with the CUs split between the two ranks of this one GPU (128 owners each, 139 rows of 1 KB + slots per owner), and through K2f"""
    _launch(_TRAIN_NF, tmp_path, 29681, env=dict(TKR_OWN=own))


def test_netflix_shape_sharded_scoring_two_ranks(tmp_path):
    """BASELINE.json configs[4], sharded: top-30 of all 17,770 items for 480,189 test lines, block-sharded over two ranks"""
    _launch(_SCORE_NF, tmp_path, 29682)


def test_two_rank_vbpr_matches_oracle(tmp_path):
    """VBPR.train on two user shards: the dense content tables cem / icb travel in the per-epoch exchange beside I and irb"""
    sys.path.insert(0, PKG)
    import synth
    r = synth.make_ratings(120, 60, 0, seed=13, mu=2.6, sigma=0.4, min_r=4, max_r=25)
    data = str(tmp_path / 'data')
    synth.write_dataset(data, r)
    _launch(_VBPR, tmp_path, 29683, data=data)
