"""Synthetic interaction data in the reference's formats.

The reference's rating files (data/f0tr.txt, f0te.*.txt, meta.pkl) are not distributed
with it, so every config runs on synthetic data of the same *shape*, written in the
reference's text formats (id lists: one token per line; ratings:
``uid,iid:like[,iid:like...]``, utils.py:58-70 / evaluate.py:30-45,84-93) and, for the
large benchmark shapes, produced directly as CSR arrays (no text round trip).

Generator: planted low-rank taste (16 factors), item popularity ~ rank^-alpha, per-user
rating counts ~ clipped lognormal, like ~ Bernoulli(sigmoid(gain*<p,q>/4 + 0.2)), an
80/20 train/test split per user; out-of-matrix items appear only in the ``om`` test file.
"""
from __future__ import annotations

import os

import numpy as np

ML10M = dict(n_users=69878, n_in=8305, n_out=2075, mu=4.2, sigma=1.0)        # data/uid, vid, f0t*.idl sizes
NETFLIX = dict(n_users=480189, n_in=14216, n_out=3554, mu=4.84, sigma=1.0)   # 17,770 items, 80/20 in/out
# ML-10M shape with selection bias and a flatter popularity curve: the preset on which a trained BPR model beats
# the popularity-only ranking (acc@30 0.066 vs 0.057, scripts/acceptance_ml10m.py) -- use it for accuracy studies
ML10M_SIGNAL = dict(ML10M, alpha=0.6, gain=2.0, select=4.0)


def make_ratings(n_users, n_in, n_out=0, seed=42, mu=4.2, sigma=1.0, alpha=0.9, gain=1.5,
                 min_r=5, max_r=3000, om_per_user=8, rank=16, select=0.0):
    """-> dict of flat arrays.  ``tr_*``/``im_*``: (user, item, like) triples of the train
    file and the in-matrix test file (items 0..n_in-1); ``om_*``: the out-of-matrix test
    file (items n_in..n_in+n_out-1).  Item numbers here are *positions in the vid list*."""
    rng = np.random.Generator(np.random.PCG64(seed))
    P = rng.standard_normal((n_users, rank)).astype(np.float32)
    Q = rng.standard_normal((n_in + n_out, rank)).astype(np.float32)
    pop = np.arange(1, n_in + 1, dtype=np.float64) ** (-alpha)
    cdf = np.cumsum(pop / pop.sum())
    perm = rng.permutation(n_in)                       # popularity rank -> item position
    cnt = np.clip(rng.lognormal(mu, sigma, n_users), min_r, min(max_r, n_in // 2)).astype(np.int64)
    over = 4 if select > 0 else 1
    users = np.repeat(np.arange(n_users, dtype=np.int64), cnt * over)
    items = perm[np.minimum(np.searchsorted(cdf, rng.random(len(users))), n_in - 1)]
    if select > 0:
        aff = np.einsum('ij,ij->i', P[users], Q[items]) * (select / 4.0)
        keep = rng.random(len(users)) < 1.0 / (1.0 + np.exp(-aff))
        users, items = users[keep], items[keep]
    key = np.unique(users * n_in + items)              # drop repeated (user,item) draws
    users, items = key // n_in, key % n_in
    shuffle = rng.permutation(len(users))              # file order inside a user's line is arbitrary
    order = shuffle[np.argsort(users[shuffle], kind='stable')]
    users, items = users[order], items[order]

    def likes(u, it):
        z = np.einsum('ij,ij->i', P[u], Q[it]) * (gain / 4.0) + 0.2
        return (rng.random(len(u)) < 1.0 / (1.0 + np.exp(-z))).astype(np.int8)

    like = likes(users, items)
    is_test = rng.random(len(users)) < 0.2
    out = dict(n_users=n_users, n_in=n_in, n_out=n_out,
               tr_u=users[~is_test], tr_i=items[~is_test], tr_l=like[~is_test],
               im_u=users[is_test], im_i=items[is_test], im_l=like[is_test])
    if n_out > 0:
        ou = np.repeat(np.arange(n_users, dtype=np.int64), om_per_user)
        oi = n_in + rng.integers(0, n_out, len(ou))
        k2 = np.unique(ou * (n_in + n_out) + oi)
        ou, oi = k2 // (n_in + n_out), k2 % (n_in + n_out)
        out.update(om_u=ou, om_i=oi, om_l=likes(ou, oi))
    else:
        out.update(om_u=np.zeros(0, np.int64), om_i=np.zeros(0, np.int64), om_l=np.zeros(0, np.int8))
    return out


def _write_ratings(path, u, it, like, uid_names, vid_names):
    with open(path, 'w') as fh:
        if len(u) == 0:
            return
        cuts = np.flatnonzero(np.r_[True, u[1:] != u[:-1], True])
        for a, b in zip(cuts[:-1], cuts[1:]):
            fh.write(uid_names[u[a]] + ''.join(',%s:%d' % (vid_names[it[p]], like[p]) for p in range(a, b)) + '\n')


def write_dataset(data_dir, r, uid_names=None, vid_names=None, fold=0, vid_order=None):
    """Write uid, vid, f{fold}tr.txt, f{fold}te.{im,om}.idl/.txt for ratings ``r``.
    ``vid_order`` (a permutation of item positions) lets the vid file list the items in
    an order different from the in-matrix/out-of-matrix id lists, as the real data does."""
    os.makedirs(data_dir, exist_ok=True)
    n_items = r['n_in'] + r['n_out']
    uid_names = uid_names or [str(x + 1) for x in range(r['n_users'])]
    vid_names = vid_names or [str(1000 + 3 * x) for x in range(n_items)]
    vid_order = np.arange(n_items) if vid_order is None else np.asarray(vid_order)
    with open(os.path.join(data_dir, 'uid'), 'w') as fh:
        fh.write(''.join(n + '\n' for n in uid_names))
    with open(os.path.join(data_dir, 'vid'), 'w') as fh:
        fh.write(''.join(vid_names[p] + '\n' for p in vid_order))
    with open(os.path.join(data_dir, 'f%dtr.idl' % fold), 'w') as fh:
        fh.write(''.join(vid_names[p] + '\n' for p in range(r['n_in'])))
    with open(os.path.join(data_dir, 'f%dte.im.idl' % fold), 'w') as fh:
        fh.write(''.join(vid_names[p] + '\n' for p in range(r['n_in'])))
    with open(os.path.join(data_dir, 'f%dte.om.idl' % fold), 'w') as fh:
        fh.write(''.join(vid_names[p] + '\n' for p in range(r['n_in'], n_items)))
    _write_ratings(os.path.join(data_dir, 'f%dtr.txt' % fold), r['tr_u'], r['tr_i'], r['tr_l'], uid_names, vid_names)
    _write_ratings(os.path.join(data_dir, 'f%dte.im.txt' % fold), r['im_u'], r['im_i'], r['im_l'], uid_names, vid_names)
    _write_ratings(os.path.join(data_dir, 'f%dte.om.txt' % fold), r['om_u'], r['om_i'], r['om_l'], uid_names, vid_names)
    return uid_names, vid_names


def positives_csr(r, n_users=None):
    """Train positives (like == 1) of ``r`` as CSR over users: row_ptr int32[n_users+1],
    pos_cols int32 (file order), cols_sorted int32 (per-row ascending), tr_users int32
    (users with >= 1 positive, ascending = first-appearance order of the sorted file)."""
    n_users = n_users or r['n_users']
    keep = r['tr_l'] == 1
    u, it = r['tr_u'][keep], r['tr_i'][keep]
    deg = np.bincount(u, minlength=n_users)
    row_ptr = np.zeros(n_users + 1, dtype=np.int64)
    np.cumsum(deg, out=row_ptr[1:])
    pos = it.astype(np.int32)                      # already grouped by user (stable order)
    srt = (np.sort(u * (r['n_in'] + r['n_out'] + 1) + it) % (r['n_in'] + r['n_out'] + 1)).astype(np.int32)
    return row_ptr.astype(np.int32), pos, srt, np.flatnonzero(deg > 0).astype(np.int32)


def train_csr_shape(n_users, n_items, mean_pos=90.0, sigma=1.0, alpha=0.9, seed=42):
    """Training positives of a given SHAPE straight as CSR (no ratings, likes or test split): per-user counts ~ clipped
    lognormal with the given mean, items by popularity rank^-alpha, repeated draws of a user dropped.  For the Netflix-shape
    training benchmarks (480,189 x 17,770: ~4 x 10^7 positives), where make_ratings' 10^8 (user, item, like) triples are
    only needed by the scoring side.  -> row_ptr int32[n_users+1], pos_cols, cols_sorted, tr_users (as positives_csr)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    mu = np.log(mean_pos) - 0.5 * sigma * sigma
    cnt = np.clip(rng.lognormal(mu, sigma, n_users), 2, n_items // 4).astype(np.int64)
    pop = np.arange(1, n_items + 1, dtype=np.float64) ** (-alpha)
    cdf = np.cumsum(pop / pop.sum())
    perm = rng.permutation(n_items)
    users = np.repeat(np.arange(n_users, dtype=np.int64), cnt)
    items = perm[np.minimum(np.searchsorted(cdf, rng.random(len(users), dtype=np.float32)), n_items - 1)]
    key = np.unique(users * n_items + items)                      # sorted by (user, item): also the per-row ascending copy
    users, srt = key // n_items, (key % n_items).astype(np.int32)
    del key, items
    row_ptr = np.zeros(n_users + 1, dtype=np.int64)
    np.cumsum(np.bincount(users, minlength=n_users), out=row_ptr[1:])
    shuffle = rng.permutation(len(srt))                           # "file order" inside a user's row is arbitrary
    pos = srt[shuffle[np.argsort(users[shuffle], kind='stable')]]
    assert row_ptr[-1] < 2 ** 31
    return row_ptr.astype(np.int32), pos, srt, np.flatnonzero(np.diff(row_ptr) > 0).astype(np.int32)


def rated_csr(r, n_users=None):
    """Every train-rated item (like 0 or 1) per user, ascending: the evaluate.py rated set."""
    n_users = n_users or r['n_users']
    n_items = r['n_in'] + r['n_out']
    key = np.sort(r['tr_u'] * n_items + r['tr_i'])
    u, it = key // n_items, key % n_items
    row_ptr = np.zeros(n_users + 1, dtype=np.int64)
    np.cumsum(np.bincount(u, minlength=n_users), out=row_ptr[1:])
    return row_ptr, it.astype(np.int32)


def make_content(n_items, d, nnz_per_row=100, seed=7):
    """tf-idf-like sparse positive features, row-L2-normalised
    -> scipy.sparse.csr_matrix [n_items, d] fp32."""
    import scipy.sparse as ss
    rng = np.random.Generator(np.random.PCG64(seed))
    nnz = min(nnz_per_row, d)
    cols = np.stack([rng.choice(d, nnz, replace=False) for _ in range(n_items)]) if d < 4096 else \
        rng.integers(0, d, (n_items, nnz))
    vals = rng.gamma(2.0, 1.0, (n_items, nnz)).astype(np.float32)
    rows = np.repeat(np.arange(n_items), nnz)
    m = ss.csr_matrix((vals.ravel(), (rows, cols.ravel())), shape=(n_items, d), dtype=np.float32)
    m.sum_duplicates()
    norm = np.sqrt(np.asarray(m.multiply(m).sum(axis=1)).ravel())
    return ss.diags(1.0 / np.maximum(norm, 1e-12)).dot(m).astype(np.float32).tocsr()
