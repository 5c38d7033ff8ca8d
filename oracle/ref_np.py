"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

A NumPy (fp32) restatement of the reference's BPR/VBPR hot path, written from the
behaviour of domainxz/top-k-rec (file:line citations are relative to the reference
root).  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module.

Pinning status
--------------
* loader, legacy sampler, text I/O, evaluate pipeline: PINNED against golden vectors
  captured from the reference's own importable code (``tests/golden/make_golden.py``,
  fixtures G1-G7 under ``tests/golden/``).
* BPR / VBPR train step (loss, gradients, sparse + dense RMSProp): **PARITY UNPINNED**.
  The arithmetic lives in tensorflow-gpu==1.15.* (reference ``requirements.txt:3``),
  which is neither vendored in the reference nor installable here.  The restatement
  follows ``single/bpr.py:81-100`` / ``single/vbpr.py:50-73`` and the published
  TF-1.15 ``RMSPropOptimizer`` semantics (decay 0.9, momentum 0, epsilon 1e-10 inside
  the root, ``rms`` slot initialised to ones, IndexedSlices de-duplicated by
  unique + sequential segment-sum before one sparse update per touched row).  It is
  checked analytically (torch-CPU autograd on the literal loss expression, and a
  hand-worked RMSProp known answer) in ``tests/test_oracle_step.py``.
"""
from __future__ import annotations

import os
from collections import defaultdict

import numpy as np

F32 = np.float32
RHO = F32(0.9)        # TF RMSPropOptimizer default decay
EPS = F32(1e-10)      # TF RMSPropOptimizer default epsilon (inside the root)


# --------------------------------------------------------------------------------------
# a1-a3  loaders (utils.py:10-16, utils.py:58-70, single/bpr.py:51-69,167-171)
# --------------------------------------------------------------------------------------
def read_id_list(path):
    """utils.py:10-16 (twin: evaluate.py:5-10).  id string -> ``len(dict)`` at the time the
    line is read: a duplicated id is re-pointed at the current size and the size does not
    grow (so a later new id shares that index).  Missing file -> empty dict."""
    ids = {}
    if os.path.isfile(path):
        with open(path) as fh:
            for line in fh:
                ids[line.strip()] = len(ids)
    return ids


def read_inverse_id_list(path):
    """evaluate.py:12-17: line index -> id string (index counts dict size, like above)."""
    inv = {}
    with open(path) as fh:
        for line in fh:
            inv[len(inv)] = line.strip()
    return inv


def read_positive_pairs(path, uids, iids):
    """utils.py:58-70: keep (uid, iid) string pairs whose like field is exactly '1' and
    whose ids are known; file order, duplicates kept; lines with no items skipped."""
    pairs = []
    if os.path.isfile(path):
        with open(path) as fh:
            for line in fh:
                fields = line.strip().split(',')
                if fields[0] in uids and len(fields) > 1:
                    for tok in fields[1:]:
                        iid, like = tok.split(':')[0], tok.split(':')[1]
                        if iid in iids and like == '1':
                            pairs.append((fields[0], iid))
    return pairs


def build_training(pairs, uids, iids):
    """single/bpr.py:63-65,167-171: uidx -> [iidx...] in file order (duplicates preserved)
    and tr_users in first-appearance order."""
    tr = defaultdict(list)
    for u, i in pairs:
        tr[uids[u]].append(iids[i])
    return tr, list(tr.keys())


def load_training(uid_file, iid_file, tr_file):
    """single/bpr.py:51-69 as one call.  Returns a dict of the public attributes."""
    uids = read_id_list(uid_file)
    iids = read_id_list(iid_file)
    pairs = read_positive_pairs(tr_file, uids, iids)
    tr_data, tr_users = build_training(pairs, uids, iids)
    return dict(uids=uids, iids=iids, data=pairs, epoch_sample_limit=len(pairs),
                n_users=len(uids), n_items=len(iids), tr_data=tr_data, tr_users=tr_users)


def load_content(content_file, iid_file, iids, n_items, d):
    """single/rec.py:23-33: unpickle (latin1) a dense or scipy-sparse [n_feat_items, d]
    matrix aligned to ``iid_file`` and place its rows by the model's item index; items
    absent from the feature id list stay zero."""
    import pickle
    import scipy.sparse as ss
    fiids = read_id_list(iid_file)
    feat = np.zeros((n_items, d), dtype=F32)
    with open(content_file, 'rb') as fh:
        raw = pickle.load(fh, encoding='latin1')
    if ss.issparse(raw):
        raw = raw.toarray()
    for iid, idx in iids.items():
        if iid in fiids:
            feat[idx, :] = raw[fiids[iid], :]
    return feat


# --------------------------------------------------------------------------------------
# a6  the reference's sampler (single/bpr.py:155-165), legacy global numpy RNG
# --------------------------------------------------------------------------------------
def legacy_uniform_user_sampler(tr_users, tr_data, n_items, batch_size):
    """Same legacy ``np.random`` calls in the same order as single/bpr.py:159-164, so with
    the same ``np.random.seed`` it reproduces the reference stream bit for bit (G2).
    Output buffers for i/j are reused across yields, as in the reference."""
    ib = np.zeros(batch_size, dtype=np.int32)
    jb = np.zeros(batch_size, dtype=np.int32)
    while True:
        ub = np.random.choice(tr_users, batch_size)
        for t in range(batch_size):
            pos = tr_data[ub[t]]
            ib[t] = np.random.choice(pos)
            jb[t] = np.random.choice(n_items)
            while jb[t] in pos:
                jb[t] = np.random.choice(n_items)
        yield ub, ib, jb


def batches_per_epoch(epoch_sample_limit, batch_size):
    """single/bpr.py:113,138-147: batch_limit = limit//B + 1, bno starts at 1 and the loop
    breaks when bno == batch_limit *after* the increment -> limit//B batches run.
    (limit < B gives batch_limit == 1, which the loop never hits: see ``BPR.train``.)"""
    return int(epoch_sample_limit) // int(batch_size)


# --------------------------------------------------------------------------------------
# a5/a7  BPR step (single/bpr.py:81-100)  -- PARITY UNPINNED (see module docstring)
# --------------------------------------------------------------------------------------
def _softplus_neg(x):
    """log(1+exp(-x)) (bpr.py:93), evaluated stably; identical to the naive form in fp32
    wherever the naive form does not overflow."""
    x = x.astype(F32)
    return (np.maximum(-x, F32(0)) + np.log1p(np.exp(-np.abs(x)))).astype(F32)


def _sigmoid_neg(x):
    """sigma(-x) = 1/(1+exp(x)) = -d/dx log(1+exp(-x))."""
    x = x.astype(F32)
    e = np.exp(-np.abs(x)).astype(F32)
    return np.where(x >= 0, e / (F32(1) + e), F32(1) / (F32(1) + e)).astype(F32)


def _segment_sum(index, values):
    """TF de-duplication of IndexedSlices: unique rows (ascending id here; the order of
    rows is irrelevant) with a *sequential* sum of duplicates in slice order."""
    uniq, inv = np.unique(index, return_inverse=True)
    out = np.zeros((len(uniq),) + values.shape[1:], dtype=F32)
    np.add.at(out, inv, values.astype(F32))      # unbuffered, in order
    return uniq, out


def _rmsprop_rows(P, ms, rows, g, lr):
    """TF-1.15 SparseApplyRMSProp with momentum 0:  ms = rho*ms + (1-rho)*g^2 ;
    P -= lr * g * rsqrt(ms + eps).  Only the touched rows change."""
    g = g.astype(F32)
    new_ms = (RHO * ms[rows] + (F32(1) - RHO) * g * g).astype(F32)
    ms[rows] = new_ms
    P[rows] = (P[rows] - F32(lr) * g / np.sqrt(new_ms + EPS)).astype(F32)


def bpr_step(state, ub, ib, jb, hp):
    """One mini-batch of single/bpr.py:81-100.  ``state`` = dict(U,V,b,msU,msV,msb) fp32,
    updated in place; ``hp`` = dict(lu,li,lj,lb,lr,mode).  Returns the summed objective
    (bpr.py:93-99) evaluated at the pre-step parameters, as fp32.

    Slice order for the item variables: the i-slices (batch order) followed by the
    j-slices (batch order) -- gradient aggregation of the two gathers on the same variable
    is a concat; ASSUMED (TF absent), and immaterial beyond fp32 rounding."""
    U, V, b = state['U'], state['V'], state['b']
    lu, li, lj, lb = (F32(hp[k]) for k in ('lu', 'li', 'lj', 'lb'))
    ub = np.asarray(ub, dtype=np.int64); ib = np.asarray(ib, dtype=np.int64); jb = np.asarray(jb, dtype=np.int64)
    ue, ie, je = U[ub], V[ib], V[jb]
    bi, bj = b[ib], b[jb]
    x_ui = np.sum(ue * ie, axis=1, dtype=F32)
    x_uj = np.sum(ue * je, axis=1, dtype=F32)
    x = (bi - bj + x_ui - x_uj).astype(F32)
    s = _sigmoid_neg(x)[:, None]
    if hp.get('mode', 'l2') == 'l2':
        loss = (np.sum(_softplus_neg(x), dtype=F32)
                + F32(0.5) * np.sum(ue * ue * lu + ie * ie * li + je * je * lj, dtype=F32)
                + F32(0.5) * np.sum(bi * bi + bj * bj, dtype=F32) * lb)
        ru, ri, rj, rbi, rbj = lu * ue, li * ie, lj * je, lb * bi, lb * bj
    else:   # bpr.py:96-99: L1, no 1/2; subgradient sign(.), sign(0)=0
        loss = (np.sum(_softplus_neg(x), dtype=F32)
                + np.sum(np.abs(ue) * lu + np.abs(ie) * li + np.abs(je) * lj, dtype=F32)
                + np.sum(np.abs(bi) + np.abs(bj), dtype=F32) * lb)
        ru, ri, rj = lu * np.sign(ue), li * np.sign(ie), lj * np.sign(je)
        rbi, rbj = lb * np.sign(bi), lb * np.sign(bj)
    gU = (-s * (ie - je) + ru).astype(F32)
    gVi = (-s * ue + ri).astype(F32)
    gVj = (s * ue + rj).astype(F32)
    gbi = (-s[:, 0] + rbi).astype(F32)
    gbj = (s[:, 0] + rbj).astype(F32)
    rows_u, sum_u = _segment_sum(ub, gU)
    items = np.concatenate([ib, jb])
    rows_v, sum_v = _segment_sum(items, np.concatenate([gVi, gVj]))
    rows_b, sum_b = _segment_sum(items, np.concatenate([gbi, gbj]))
    if hp.get('opt', 'rmsprop') == 'sgd':
        # legacy optimiser, old/methods/bpr.py:57-61: P <- P - lr * dcost/dP.  Theano's dense
        # gradient of the gathered rows is the same per-row sum (zero on untouched rows); L2 objective only
        # (old/methods/bpr.py:43-51 == single/bpr.py:92-95).  PARITY UNPINNED (Theano absent), checked
        # against torch autograd in tests/test_oracle_step.py.
        lr = F32(hp['lr'])
        U[rows_u] = (U[rows_u] - lr * sum_u).astype(F32)
        V[rows_v] = (V[rows_v] - lr * sum_v).astype(F32)
        b[rows_b] = (b[rows_b] - lr * sum_b).astype(F32)
        return F32(loss)
    _rmsprop_rows(U, state['msU'], rows_u, sum_u, hp['lr'])
    _rmsprop_rows(V, state['msV'], rows_v, sum_v, hp['lr'])
    _rmsprop_rows(b, state['msb'], rows_b, sum_b, hp['lr'])
    return F32(loss)


def legacy_pregenerated_sampler(train_dict, n_items, n_samples):
    """old/methods/bpr.py:88-99 (_uniform_user_sampling): ALL users first with one vectorised
    ``np.random.randint(len(train_dict), size=n)`` over ``list(train_dict.keys())``, then per
    sample one positive (``randint(len(pos))``) and rejection-sampled negative
    (``randint(n_items)`` until not in the user's positives).  Uses the legacy global
    ``np.random`` stream like the reference (pinned by golden G8).  Returns three int64 arrays."""
    keys = np.array(list(train_dict.keys()))
    users = keys[np.random.randint(len(train_dict), size=n_samples)]
    pos, neg = [], []
    for u in users:
        items = train_dict[u]
        pos.append(items[np.random.randint(len(items))])
        j = np.random.randint(n_items)
        while j in items:
            j = np.random.randint(n_items)
        neg.append(j)
    return users.astype(np.int64), np.asarray(pos, dtype=np.int64), np.asarray(neg, dtype=np.int64)


def legacy_batches(n_samples, batch_size):
    """old/methods/bpr.py:72-77: ``while (z+1)*batch_size < n_sgd_samples`` -- strict, so a final batch
    that would end exactly at n_samples is dropped as well."""
    return max(0, (n_samples - 1) // batch_size) if n_samples > 0 else 0


def init_bpr_state(n_users, n_items, k, rng):
    """single/bpr.py:77-79: U,V ~ N(0, 0.01^2), b = 0; RMSProp ``rms`` slots = 1."""
    return dict(U=(rng.standard_normal((n_users, k)) * 0.01).astype(F32),
                V=(rng.standard_normal((n_items, k)) * 0.01).astype(F32),
                b=np.zeros(n_items, dtype=F32),
                msU=np.ones((n_users, k), dtype=F32), msV=np.ones((n_items, k), dtype=F32),
                msb=np.ones(n_items, dtype=F32))


# --------------------------------------------------------------------------------------
# a8/a9  VBPR step (single/vbpr.py:50-73)  -- PARITY UNPINNED
# --------------------------------------------------------------------------------------
def init_vbpr_state(n_users, n_items, k, d, rng):
    """single/vbpr.py:37-48.  kh = k//2; cem = const 2/(d*k); irb, icb = 0."""
    kh = k // 2
    st = dict(ure=(rng.standard_normal((n_users, kh)) * 0.01).astype(F32),
              uce=(rng.standard_normal((n_users, kh)) * 0.01).astype(F32),
              ire=(rng.standard_normal((n_items, kh)) * 0.01).astype(F32),
              irb=np.zeros(n_items, dtype=F32),
              cem=np.full((d, kh), 2.0 / (d * k), dtype=F32),
              icb=np.zeros(d, dtype=F32))
    for name in list(st):
        st['ms_' + name] = np.ones_like(st[name])
    return st


def vbpr_step(state, feat, ub, ib, jb, hp):
    """One mini-batch of single/vbpr.py:50-73 (+ the host gather of :114).  Sparse RMSProp
    on ure/uce/ire/irb (touched rows), *dense* RMSProp on cem/icb (every element, every
    batch; the whole cem / icb is regularised each batch, vbpr.py:65,67).

    The data term is a sum over ALL PAIRS of the batch.  item_rating_bias has shape [n_items, 1] (vbpr.py:43), so
    ``irbb - jrbb`` is [B, 1], ``x_ui - x_uj`` is [B] and ``matmul(ic - jc, icb)`` is [B, 1] (:59-61): their sum
    broadcasts to x_uij[a, b] = alpha[a] + beta[b] with alpha = irb_i - irb_j + (f_i - f_j).icb (row index) and
    beta = x_ui - x_uj (column index), and ``reduce_sum(log(1 + exp(-x_uij)))`` (:64) runs over the whole [B, B] matrix.
    (tests/test_oracle_step.py checks this function against autograd on the literal expressions WITH those shapes.)"""
    ure, uce, ire, irb, cem, icb = (state[n] for n in ('ure', 'uce', 'ire', 'irb', 'cem', 'icb'))
    lu, li, lj, lb, le = (F32(hp[k]) for k in ('lu', 'li', 'lj', 'lb', 'le'))
    ub = np.asarray(ub, dtype=np.int64); ib = np.asarray(ib, dtype=np.int64); jb = np.asarray(jb, dtype=np.int64)
    ur, uc, ir, jr = ure[ub], uce[ub], ire[ib], ire[jb]
    bi, bj = irb[ib], irb[jb]
    ic, jc = feat[ib], feat[jb]                       # vbpr.py:114 host gather
    ice = (ic @ cem).astype(F32)                      # vbpr.py:56-57
    jce = (jc @ cem).astype(F32)
    x_ui = np.sum(ur * ir + uc * ice, axis=1, dtype=F32)
    x_uj = np.sum(ur * jr + uc * jce, axis=1, dtype=F32)
    dfeat = (ic - jc).astype(F32)
    alpha = (bi - bj + dfeat @ icb).astype(F32)       # the [B, 1] terms of vbpr.py:61
    beta = (x_ui - x_uj).astype(F32)                  # the [B] terms
    x = (alpha[:, None] + beta[None, :]).astype(F32)  # x_uij [B, B]
    sg = _sigmoid_neg(x)
    sa = np.sum(sg, axis=1, dtype=F32)[:, None]       # d obj / d alpha_a = -sa[a]
    s = np.sum(sg, axis=0, dtype=F32)[:, None]        # d obj / d beta_b  = -s[b]
    if hp.get('mode', 'l2') == 'l2':
        loss = (np.sum(_softplus_neg(x), dtype=F32)
                + F32(0.5) * np.sum(cem * cem, dtype=F32) * le
                + F32(0.5) * np.sum((ur * ur + uc * uc) * lu + ir * ir * li + jr * jr * lj, dtype=F32)
                + F32(0.5) * (np.sum(bi * bi + bj * bj, dtype=F32) + np.sum(icb * icb, dtype=F32)) * lb)
        r_ur, r_uc, r_ir, r_jr = lu * ur, lu * uc, li * ir, lj * jr
        r_bi, r_bj, r_cem, r_icb = lb * bi, lb * bj, le * cem, lb * icb
    else:
        loss = (np.sum(_softplus_neg(x), dtype=F32)
                + np.sum(np.abs(cem), dtype=F32) * le
                + np.sum((np.abs(ur) + np.abs(uc)) * lu + np.abs(ir) * li + np.abs(jr) * lj, dtype=F32)
                + (np.sum(np.abs(bi) + np.abs(bj), dtype=F32) + np.sum(np.abs(icb), dtype=F32)) * lb)
        r_ur, r_uc, r_ir, r_jr = lu * np.sign(ur), lu * np.sign(uc), li * np.sign(ir), lj * np.sign(jr)
        r_bi, r_bj, r_cem, r_icb = lb * np.sign(bi), lb * np.sign(bj), le * np.sign(cem), lb * np.sign(icb)
    g_ur = (-s * (ir - jr) + r_ur).astype(F32)
    g_uc = (-s * (ice - jce) + r_uc).astype(F32)
    g_ir = (-s * ur + r_ir).astype(F32)
    g_jr = (s * ur + r_jr).astype(F32)
    g_bi = (-sa[:, 0] + r_bi).astype(F32)
    g_bj = (sa[:, 0] + r_bj).astype(F32)
    d_ice = (-s * uc).astype(F32)                     # dL/d(iceb); dL/d(jceb) = -d_ice
    g_cem = (ic.T @ d_ice + jc.T @ (-d_ice) + r_cem).astype(F32)
    g_icb = (dfeat.T @ (-sa[:, 0]) + r_icb).astype(F32)
    lr = hp['lr']
    rows_u, sum_ur = _segment_sum(ub, g_ur)
    _, sum_uc = _segment_sum(ub, g_uc)
    items = np.concatenate([ib, jb])
    rows_i, sum_ir = _segment_sum(items, np.concatenate([g_ir, g_jr]))
    _, sum_b = _segment_sum(items, np.concatenate([g_bi, g_bj]))
    _rmsprop_rows(ure, state['ms_ure'], rows_u, sum_ur, lr)
    _rmsprop_rows(uce, state['ms_uce'], rows_u, sum_uc, lr)
    _rmsprop_rows(ire, state['ms_ire'], rows_i, sum_ir, lr)
    _rmsprop_rows(irb, state['ms_irb'], rows_i, sum_b, lr)
    for name, g in (('cem', g_cem), ('icb', g_icb)):      # dense ApplyRMSProp
        ms = state['ms_' + name]
        ms[...] = (ms + (g * g - ms) * (F32(1) - RHO)).astype(F32)
        state[name][...] = (state[name] - F32(lr) * g / np.sqrt(ms + EPS)).astype(F32)
    return F32(loss)


def vbpr_fold(state, feat):
    """single/vbpr.py:124-126: fue=[ure|uce], fie=[ire|feat.cem], fib=irb+feat.icb."""
    fue = np.concatenate([state['ure'], state['uce']], axis=1)
    fie = np.concatenate([state['ire'], (feat @ state['cem']).astype(F32)], axis=1)
    fib = (state['irb'][:, None] + (feat @ state['icb'])[:, None]).astype(F32)
    return fue, fie, fib


# --------------------------------------------------------------------------------------
# a10  text matrices (utils.py:28-55)
# --------------------------------------------------------------------------------------
def write_embed_text(path, mat):
    """utils.py:47-55: every element as '%f ' (trailing space), one row per line; the
    parent directory is created non-recursively when missing."""
    parent = os.path.dirname(path)
    if not os.path.isdir(parent):
        os.mkdir(parent)
    with open(path, 'w') as fh:
        for row in np.asarray(mat):
            fh.write(''.join('%f ' % v for v in row) + '\n')


def read_embed_text(path, ids=None):
    """utils.py:28-44: rows addressed by the id dict's index (or all lines) -> fp32."""
    if not os.path.isfile(path):
        return None
    with open(path) as fh:
        lines = fh.readlines()
    order = range(len(lines)) if ids is None else ids.values()
    n = len(lines) if ids is None else len(ids)
    out = None
    for idx in order:
        vals = np.float32(lines[idx].strip().split(' '))
        if out is None:
            out = np.zeros((n, len(vals)), dtype=F32)
        out[idx, :] = vals
    return out


# --------------------------------------------------------------------------------------
# a11-a14  evaluate.py pipeline
# --------------------------------------------------------------------------------------
def read_history(path):
    """evaluate.py:30-45: uid -> set of every vid string on the user's train line
    (like 0 or 1).  (The popularity counter the reference also builds is unused.)"""
    rated = {}
    with open(path) as fh:
        for line in fh:
            fields = line.strip().split(',')
            rated[fields[0]] = {tok.split(':')[0] for tok in fields[1:]}
    return rated


def read_test_likes(path, teids):
    """evaluate.py:84-93: per test line, (uid, set of test-column ids with like == 1)."""
    out = []
    with open(path) as fh:
        for line in fh:
            fields = line.strip().split(',')
            likes = set()
            for tok in fields[1:]:
                vid, like = tok.split(':')[0], int(tok.split(':')[1])
                if like == 1:
                    likes.add(teids[vid])
            out.append((fields[0], likes))
    return out


def scenario_scores(umat, vmat, bmat, vids, teids):
    """evaluate.py:75-80 with the F5 fix: bias gathered per test id (the reference adds the
    whole vid-ordered bias row, which only broadcasts when the .idl equals vid)."""
    temat = np.zeros((len(teids), vmat.shape[1]), dtype=F32)
    tebias = np.zeros(len(teids), dtype=F32)
    for vid, col in teids.items():
        temat[col, :] = vmat[vids[vid], :]
        if bmat is not None:
            tebias[col] = bmat.reshape(-1)[vids[vid]]
    scores = np.dot(umat, temat.T)
    if bmat is not None:
        scores += tebias.reshape((1, -1))
    return scores


def mfma_chain_scores(umat, temat, tebias=None):
    """The fp32 dot product of evaluate.py:78-80 (``np.dot(umat, temat.T)`` + bias: BLAS picks the summation order there) in
    the ONE order the build's fp32 arithmetic uses (include/tkr.h tkr_topk_set_math modes 1 and 2): the two halves of the factor
    vector interleaved -- ``acc = fma(v[j], u[j], acc); acc = fma(v[KH+j], u[KH+j], acc)`` for j = 0..KH-1, KH = ceil(k/2) --
    one rounding per fused multiply-add, then ``fl(acc + bias)``.  Products are exact in float64 (24 x 24 bits); the sum is rounded
    to 53 and then to 24 bits, which differs from a single rounding only when the float64 sum lands exactly on a float32 tie --
    the comparing tests allow for a 1e-4 fraction of such scores.  Any order is a valid reading of the reference line; this one is
    what the kernels are held to, bit for bit."""
    u64, v64 = np.asarray(umat, dtype=np.float64), np.asarray(temat, dtype=np.float64)
    k = u64.shape[1]
    kh = (k + 1) // 2
    acc = np.zeros((u64.shape[0], v64.shape[0]), dtype=F32)
    for j in range(kh):
        acc = (np.outer(u64[:, j], v64[:, j]) + acc).astype(F32)
        if kh + j < k:
            acc = (np.outer(u64[:, kh + j], v64[:, kh + j]) + acc).astype(F32)
    if tebias is not None:
        acc = (acc + np.asarray(tebias, dtype=F32).reshape(1, -1)).astype(F32)
    return acc + F32(0.0)


def filtered_topk(scores_row, rated_cols, total, canonical=True):
    """evaluate.py:96-105 ranking part for one user: walk the ascending argsort from the
    end, skip rated columns, stop after ``total`` kept.  ``canonical=True`` uses the
    build's stated tie rule (descending score, ties -> higher column first = stable
    ascending argsort read backwards); ``False`` uses numpy's default (unspecified) kind,
    exactly as the reference does."""
    order = np.argsort(scores_row, kind='stable') if canonical else np.argsort(scores_row)
    kept = []
    for c in order[::-1]:
        if c not in rated_cols:
            kept.append(int(c))
            if len(kept) == total:
                break
    return kept


def bucket_hits(kept, likes, step, interval):
    """evaluate.py:99-103: a liked item at filtered position p adds one to buckets
    p//step .. interval-1 (range is empty when p//step >= interval)."""
    hits = [0] * interval
    for p, c in enumerate(kept):
        if c in likes:
            for q in range(p // step, interval):
                hits[q] += 1
    return hits


def evaluate_scenario(umat, vmat, bmat, uids, vids, rated, teids, teivt, test_lines,
                      step=5, total=30, canonical=True, return_lists=False):
    """evaluate.py:72-112 for one scenario -> list of interval accuracies
    (sum of hits / sum of |likes|).  ZeroDivisionError when no test like exists."""
    interval = total // step
    scores = scenario_scores(umat, vmat, bmat, vids, teids)
    tres = [0.0] * interval
    tcount = 0
    lists = {}
    for uid, likes in test_lines:
        if len(likes) == 0:
            continue
        rated_cols = {teids[v] for v in rated[uid] if v in teids}
        kept = filtered_topk(scores[uids[uid]], rated_cols, total, canonical)
        lists[uid] = kept
        hits = bucket_hits(kept, likes, step, interval)
        for q in range(interval):
            tres[q] += hits[q]
        tcount += len(likes)
    acc = [tres[q] / tcount for q in range(interval)]
    return (acc, lists) if return_lists else acc


# --------------------------------------------------------------------------------------
# n3  utils.py:18-24, 73-127: get_iv_dict / get_history_from_file / get_score / evaluate  (pinned by golden G9)
# --------------------------------------------------------------------------------------
def read_iv_list(path):
    """utils.py:18-24 get_iv_dict_from_file: line number -> stripped token ({} when the file is missing)."""
    ivt = {}
    if os.path.isfile(path):
        with open(path) as fh:
            for line in fh:
                ivt[len(ivt)] = line.strip()
    return ivt


def read_history_counts(path):
    """utils.py:73-89 get_history_from_file: (uid -> set of every iid on its LAST line, iid -> number of
    lines liking it with the literal string '1').  Missing file -> two empty dicts."""
    browsed, counter = {}, {}
    if os.path.isfile(path):
        with open(path) as fh:
            for line in fh:
                fields = line.strip().split(',')
                browsed[fields[0]] = set()
                for tok in fields[1:]:
                    iid, like = tok.split(':')[0], tok.split(':')[1]
                    browsed[fields[0]].add(iid)
                    if like == '1':
                        counter[iid] = counter.get(iid, 0) + 1
    return browsed, counter


def utils_get_score(U, V, iids, sub_iids):
    """utils.py:92-98: rows of V re-ordered to the sub id list (ids missing from iids stay zero), then np.dot."""
    subV = np.zeros((len(sub_iids), V.shape[1]), dtype=F32)
    for iid in iids:
        if iid in sub_iids:
            subV[sub_iids[iid], :] = V[iids[iid], :]
    return np.dot(U, subV.T)


def utils_evaluate(score, rated, likes, uids, te_iids, te_ivt, step, total, interval, canonical=True):
    """utils.py:101-127: walk every user's ranking from the top; an unrated liked item at RAW rank t (rated items
    counted) adds 1 and 1/(t+1) to buckets t//step .. interval-1; stop after ``total`` unrated items.
    -> (hits, trrs, count).  ``canonical``: stable argsort (ties -> higher column first) instead of numpy's
    unspecified default kind."""
    count = 0
    hits, trrs = [0.0] * interval, [0.0] * interval
    ranks = np.argsort(score, axis=1, kind='stable') if canonical else np.argsort(score, axis=1)
    n = len(te_iids)
    for uid in likes:
        like = likes[uid]
        if len(like) == 0:
            continue
        idx = 0
        hit, rrs = [0.0] * interval, [0.0] * interval            # per user first, then into the totals (:110-124)
        for t in range(n):
            riid = te_ivt[ranks[uids[uid], n - 1 - t]]
            if riid not in rated[uid]:
                if riid in like:
                    for q in range(t // step, interval):
                        hit[q] += 1
                        rrs[q] += 1.0 / (t + 1)
                idx += 1
            if idx == total:
                break
        for q in range(interval):
            hits[q] += hit[q]
            trrs[q] += rrs[q]
        count += len(like)
    return hits, trrs, count


def evaluate_cli(data_dir, model_dir, fold=0, step=5, total=30, scenarios=('im', 'om'),
                 canonical=True):
    """The whole evaluate.py CLI as a function -> list of stdout lines 'S,%.6f,...'."""
    uids = read_id_list(os.path.join(data_dir, 'uid'))
    vids = read_id_list(os.path.join(data_dir, 'vid'))
    rated = read_history(os.path.join(data_dir, 'f%dtr.txt' % fold))
    umat = read_embed_text(os.path.join(model_dir, 'final-U.dat'), uids)
    vmat = read_embed_text(os.path.join(model_dir, 'final-V.dat'), vids)
    bpath = os.path.join(model_dir, 'final-B.dat')
    bmat = read_embed_text(bpath, vids) if os.path.exists(bpath) else None
    lines = []
    for sc in scenarios:
        idl = os.path.join(data_dir, 'f%dte.%s.idl' % (fold, sc))
        teids, teivt = read_id_list(idl), read_inverse_id_list(idl)
        tests = read_test_likes(os.path.join(data_dir, 'f%dte.%s.txt' % (fold, sc)), teids)
        acc = evaluate_scenario(umat, vmat, bmat, uids, vids, rated, teids, teivt, tests,
                                step, total, canonical)
        lines.append(sc + ''.join(',%.6f' % a for a in acc))
    return lines
