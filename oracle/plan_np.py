"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

Bit-exact NumPy restatement of the build's *device* sampler + batch planner (kernel K1,
``top-k-rec_amd/csrc/sampler.hip``).  Distribution = the reference's
``BPR._uniform_user_sampling`` (single/bpr.py:155-165): u ~ U(tr_users) with replacement,
i ~ U(tr_data[u]) (file order, duplicates kept), j ~ U{0..n_items-1} rejected while
j in tr_data[u].  The random STREAM is the build's own (Philox4x32-10, counter-based) --
the reference is unseeded (global legacy numpy RNG), so no stream exists to reproduce;
the verbatim legacy sampler lives in ``ref_np.legacy_uniform_user_sampler`` (pinned by G2).

Stream definition (everything below is integer arithmetic; the HIP kernel must match
bit for bit):
  key      = (seed & 0xffffffff, seed >> 32)
  counter  = (g & 0xffffffff, g >> 32, round, 0)       g = first_triplet + b*B + t
  round 0  : w = philox(counter);  u = tr_users[mulhi64(w0|w1<<32, n_tr)]
                                   i = pos_cols[row_ptr[u] + mulhi64(w2|w3<<32, deg(u))]
  round r>0: two negative candidates  mulhi64(w0|w1<<32, n_items), mulhi64(w2|w3<<32, n_items)
             taken in that order; first one not in tr_data[u] wins; r = 1, 2, ... MAX_ROUNDS
             After MAX_ROUNDS (only reachable when a user has rated almost every item) the
             last candidate is advanced cyclically (j+1 mod n_items) until it is not in
             tr_data[u], at most n_items steps (a user who rated EVERY item makes the
             reference spin forever; here the candidate comes back unchanged).
  mulhi64(x, n) = floor(x * n / 2**64)

Plan definition (per batch of B triplets, consumed by the step kernels):
  user occurrences   t = 0..B-1 sorted by (u[t], t)           -> occ[p]    = (i[t], j[t])
  item occurrences   o = 0..2B-1 (o<B: item i[o], role 0; else item j[o-B], role 1)
                     sorted by (item[o], o)                    -> occ[B+p] = (u[t], other | role<<31)
  tasks              one per unique user (ascending id) then one per unique item
                     (ascending id): (row | kind<<31, occ_start, occ_count, parity); unused
                     slots up to 3B hold (-1, 0, 0, 0).
  parity             which of the two table buffers holds a row's value at the START of the
                     batch = (number of earlier batches that updated the row) & 1, from the
                     running per-row update counters ucnt / icnt.  Stored in task[.,3] for the
                     task's own row and in bit 30 of every occ id for the partner rows.
Launch plan (what the step kernel actually reads; derived from the above):
  a task with <= light_max(B) occurrences (4 for B <= 4096, else 16) is LIGHT (one wave), otherwise HEAVY (a team of TEAM
  waves = one workgroup, wave w takes occurrences w, w+TEAM, ...; TEAM = team_for(B): 4 up to
  B = 1024, 8 up to 16,384, 16 above).  Workgroups hold TEAM wave
  records; light tasks fill workgroups 0..nlb-1 in task order, heavy task h is workgroup nlb+h.
  wave record = 16 int32: [0] row|kind<<31 (-1 = idle)  [1] parity | team<<8 | rank<<16
                          [2] occurrences of this wave   [3] index of its first occurrence
                          [4..11] its first <=4 occurrences (a,b)   [12] task occ_count
                          [13] t0 | t1<<16  [14] t2 | t3<<16  (triplet index of those occurrences)  [15] 0
  occt[3B]           triplet index t of every sorted occurrence (VBPR reads per-triplet results)
  header per batch = (workgroups used, light workgroups, heavy tasks, tasks); light workgroups hold
  light_per_block(B) tasks each, remaining wave slots are idle (-1).
"""
from __future__ import annotations

import numpy as np

U32 = np.uint32
U64 = np.uint64
MAX_ROUNDS = 64
LIGHT_MAX = 4          # occurrences a single wave handles for B <= 4096 (csrc/sampler.hip)
LIGHT_MAX_BIG = 16     # ... and for larger batches (fewer, fuller 16-wave teams: dispatching them dominates)
TEAM = 16              # waves per workgroup / per heavy task for batches > TEAM_MID_MAX_B
TEAM_MID = 8           # ... for TEAM_SMALL_MAX_B < batch <= TEAM_MID_MAX_B (three 8-wave workgroups share a CU where one of 16 waves sits alone)
TEAM_MID_MAX_B = 16384
TEAM_SMALL = 4         # ... and for small batches: 4-wave workgroups spread a 256-batch over ~180 CUs
TEAM_SMALL_MAX_B = 1024


def light_per_block(B):
    """light tasks packed into one workgroup (= every wave slot; half-filled groups were measured slower)"""
    return team_for(B)


LIGHT_BIG_MIN_B = 4096  # batches above this use LIGHT_MAX_BIG


def light_max(B):
    return LIGHT_MAX if B <= LIGHT_BIG_MIN_B else LIGHT_MAX_BIG


def team_for(B):
    """tkr_plan_team: waves per workgroup (= per heavy-row team) of the step kernels"""
    return TEAM_SMALL if B <= TEAM_SMALL_MAX_B else TEAM_MID if B <= TEAM_MID_MAX_B else TEAM
PAR_BIT = 1 << 30

_M0, _M1 = U64(0xD2511F53), U64(0xCD9E8D57)
_W0, _W1 = 0x9E3779B9, 0xBB67AE85
_MASK = U64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox4x32-10 (Salmon et al., Random123) on arrays of uint32 counters/keys."""
    c0, c1, c2, c3 = (np.asarray(x, dtype=U64) & _MASK for x in (c0, c1, c2, c3))
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    for rnd in range(10):
        p0 = _M0 * c0
        p1 = _M1 * c2
        hi0, lo0 = p0 >> U64(32), p0 & _MASK
        hi1, lo1 = p1 >> U64(32), p1 & _MASK
        c0, c1, c2, c3 = (hi1 ^ c1 ^ U64(k0)), lo1, (hi0 ^ c3 ^ U64(k1)), lo0
        k0 = (k0 + _W0) & 0xFFFFFFFF
        k1 = (k1 + _W1) & 0xFFFFFFFF
    return c0.astype(U32), c1.astype(U32), c2.astype(U32), c3.astype(U32)


def mulhi64(lo, hi, n):
    """floor((hi<<32 | lo) * n / 2**64) for uint32 arrays lo, hi and 0 < n < 2**32."""
    lo = np.asarray(lo, dtype=U64)
    hi = np.asarray(hi, dtype=U64)
    n = np.asarray(n, dtype=U64)
    return ((hi * n + ((lo * n) >> U64(32))) >> U64(32)).astype(np.int64)


def build_csr(tr_data, n_users):
    """tr_data (uidx -> [iidx...]) -> row_ptr[n_users+1], pos_cols (file order, duplicates
    kept), cols_sorted (per-row ascending copy used for the membership test)."""
    row_ptr = np.zeros(n_users + 1, dtype=np.int64)
    for u, items in tr_data.items():
        row_ptr[u + 1] = len(items)
    row_ptr = np.cumsum(row_ptr)
    pos = np.zeros(int(row_ptr[-1]), dtype=np.int32)
    srt = np.zeros(int(row_ptr[-1]), dtype=np.int32)
    for u, items in tr_data.items():
        a = np.asarray(items, dtype=np.int32)
        pos[row_ptr[u]:row_ptr[u + 1]] = a
        srt[row_ptr[u]:row_ptr[u + 1]] = np.sort(a)
    return row_ptr.astype(np.int32), pos, srt


def _member(u, j, row_ptr, cols_sorted, n_items):
    """j in tr_data[u] via one global searchsorted over (u*n_items + col) keys."""
    deg = np.diff(row_ptr.astype(np.int64))
    owner = np.repeat(np.arange(len(deg), dtype=np.int64), deg)
    keys = owner * int(n_items) + cols_sorted.astype(np.int64)       # ascending overall
    q = u.astype(np.int64) * int(n_items) + j.astype(np.int64)
    p = np.searchsorted(keys, q)
    p = np.minimum(p, max(len(keys) - 1, 0))
    return (keys[p] == q) if len(keys) else np.zeros(len(q), dtype=bool)


def sample_triplets(tr_users, row_ptr, pos_cols, cols_sorted, n_items, seed, first_triplet, count):
    """`count` consecutive triplets of the device stream starting at global index
    `first_triplet` -> (u, i, j) int32 arrays."""
    tr_users = np.asarray(tr_users, dtype=np.int32)
    g = np.arange(count, dtype=np.uint64) + np.uint64(first_triplet)
    c0, c1 = (g & _MASK).astype(U32), (g >> U64(32)).astype(U32)
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    zeros = np.zeros(count, dtype=U32)
    w0, w1, w2, w3 = philox4x32_10(c0, c1, zeros, zeros, k0, k1)
    u = tr_users[mulhi64(w0, w1, len(tr_users))]
    start = row_ptr[u].astype(np.int64)
    deg = row_ptr[u + 1].astype(np.int64) - start
    i = pos_cols[start + mulhi64(w2, w3, deg)]
    j = np.zeros(count, dtype=np.int32)
    pending = np.arange(count)
    for rnd in range(1, MAX_ROUNDS + 1):
        if len(pending) == 0:
            break
        w0, w1, w2, w3 = philox4x32_10(c0[pending], c1[pending], np.full(len(pending), rnd, dtype=U32),
                                       zeros[pending], k0, k1)
        ca = mulhi64(w0, w1, n_items).astype(np.int32)
        cb = mulhi64(w2, w3, n_items).astype(np.int32)
        ra = _member(u[pending], ca, row_ptr, cols_sorted, n_items)
        rb = _member(u[pending], cb, row_ptr, cols_sorted, n_items)
        j[pending] = np.where(~ra, ca, cb)          # if both rejected, cb stays as "last candidate"
        pending = pending[ra & rb]
    for p in pending:                                # cyclic-scan fallback, see module docstring
        lo, hi = int(row_ptr[u[p]]), int(row_ptr[u[p] + 1])
        rated = set(cols_sorted[lo:hi].tolist())
        cand = int(j[p])
        for _ in range(int(n_items)):
            if cand not in rated:
                break
            cand = (cand + 1) % int(n_items)
        j[p] = cand
    return u.astype(np.int32), i.astype(np.int32), j


def plan_batch(u, i, j, return_t=False):
    """Plan of ONE batch (see module docstring) -> (task int32[3B,4], occ int32[3B,2][, occt int32[3B]])."""
    B = len(u)
    task = np.zeros((3 * B, 4), dtype=np.int32)
    task[:, 0] = -1
    occ = np.zeros((3 * B, 2), dtype=np.int32)
    # users
    order = np.argsort(u.astype(np.int64), kind='stable')
    su = u[order]
    occ[:B, 0] = i[order]
    occ[:B, 1] = j[order]
    occt = np.zeros(3 * B, dtype=np.int32)
    occt[:B] = order
    heads = np.flatnonzero(np.r_[True, su[1:] != su[:-1]])
    ends = np.r_[heads[1:], B]
    nu = len(heads)
    task[:nu, 0] = su[heads]
    task[:nu, 1] = heads
    task[:nu, 2] = ends - heads
    # items
    item = np.concatenate([i, j]).astype(np.int64)
    order = np.argsort(item, kind='stable')
    si = item[order]
    t = order % B
    role = (order >= B)
    other = np.where(role, i[t], j[t]).astype(np.int64)
    occ[B:, 0] = u[t]
    occt[B:] = t
    occ[B:, 1] = (other | (role.astype(np.int64) << 31)).astype(np.uint32).view(np.int32)
    heads = np.flatnonzero(np.r_[True, si[1:] != si[:-1]])
    ends = np.r_[heads[1:], 2 * B]
    ni = len(heads)
    task[nu:nu + ni, 0] = (si[heads] | (1 << 31)).astype(np.uint32).view(np.int32)
    task[nu:nu + ni, 1] = B + heads
    task[nu:nu + ni, 2] = ends - heads
    return (task, occ, occt) if return_t else (task, occ)


def max_blocks(B):
    """workgroups a batch can need: light tasks 16 per group + heavy tasks (>= 5 occurrences each)"""
    lpb = light_per_block(B)
    return (3 * B + lpb - 1) // lpb + (3 * B) // (light_max(B) + 1)


def resolve_parity(task, occ, B, ucnt, icnt):
    """Patch parities into one batch's plan (in place) and bump the update counters."""
    live = np.flatnonzero(task[:, 0] != -1)
    rows = task[live, 0] & 0x7FFFFFFF
    is_item = task[live, 0] < 0
    par = np.zeros(len(live), dtype=np.int32)
    par[is_item] = icnt[rows[is_item]] & 1
    par[~is_item] = ucnt[rows[~is_item]] & 1
    task[live, 3] = par
    uo = occ[:B]                                     # user occurrences: (i, j) both items
    uo[:, 0] |= ((icnt[uo[:, 0]] & 1) << 30).astype(np.int32)
    uo[:, 1] |= ((icnt[uo[:, 1]] & 1) << 30).astype(np.int32)
    io = occ[B:]                                     # item occurrences: (u, other|role<<31)
    other = io[:, 1] & 0x3FFFFFFF
    io[:, 1] |= ((icnt[other] & 1) << 30).astype(np.int32)
    io[:, 0] |= ((ucnt[io[:, 0]] & 1) << 30).astype(np.int32)
    ucnt[rows[~is_item]] += 1
    icnt[rows[is_item]] += 1


def flow_records(task, occ, B, ucnt, icnt, batch, last_u=None, last_i=None, n_owner=0, occt=None):
    """Dataflow form of one batch's plan (tkr_sample_plan with prec / pocc; consumed by the persistent step kernels,
    csrc/bpr_flow.hip and csrc/bpr_own.hip).  `task`, `occ` are plan_batch's output BEFORE resolve_parity; ucnt / icnt the
    update counters before this batch = the VERSION of every row this batch reads; last_u / last_i (optional, updated in
    place) the last batch of this call that updated each row, -1 = none.
      pocc[3B,4]   per sorted occurrence (a, version of a, b | role<<31, version of b)
      prec[3B,32]  per task slot: [0] row|kind<<31 (-1 unused) [1] version [2] occurrences [3] batch*3B + first occurrence
                   [4] batch [5] last batch < [4] of this call that updated the row (-1: none)
                   [8+4q..11+4q] pocc of occurrence q < min(4, occurrences); everything else 0
    n_owner > 0 (tkr_sample_plan_owned, K2o): the records of the batch's ITEM tasks sit in (row % n_owner, row) order instead of
    row order (same slots) and ohdr[n_owner] = first slot | tasks << 16 of every owner's run is returned as a third value.
    occt (the triplet index of every sorted occurrence): prec[24+q] = triplet of occurrence q < min(4, occurrences) -- K2o's item
    tasks exchange the scalars <u, v> + b of a triplet through a slot named by it."""
    pocc = np.zeros((3 * B, 4), dtype=np.int32)
    pocc[:B, 0] = occ[:B, 0]
    pocc[:B, 1] = icnt[occ[:B, 0]]
    pocc[:B, 2] = occ[:B, 1]
    pocc[:B, 3] = icnt[occ[:B, 1]]
    pocc[B:, 0] = occ[B:, 0]
    pocc[B:, 1] = ucnt[occ[B:, 0]]
    pocc[B:, 2] = occ[B:, 1]
    pocc[B:, 3] = icnt[occ[B:, 1] & 0x3FFFFFFF]
    prec = np.zeros((3 * B, 32), dtype=np.int32)
    prec[:, 0] = -1
    live = np.flatnonzero(task[:, 0] != -1)
    dst = {int(s): int(s) for s in live}
    ohdr = None
    if n_owner > 0:
        items = [int(s) for s in live if task[s, 0] < 0]                 # row order
        first = items[0]
        owner = [(int(task[s, 0]) & 0x7FFFFFFF) % n_owner for s in items]
        order = sorted(range(len(items)), key=lambda q: (owner[q], q))     # stable: row order inside an owner
        for rank, q in enumerate(order):
            dst[items[q]] = first + rank
        cnt = np.bincount(owner, minlength=n_owner)
        start = first + np.concatenate([[0], np.cumsum(cnt)[:-1]])
        ohdr = (start | (cnt << 16)).astype(np.int32)
    for s in live:
        rowk, start_, cnt_, _ = task[s]
        row = int(rowk) & 0x7FFFFFFF
        r = prec[dst[int(s)]]
        r[0] = rowk
        r[1] = icnt[row] if rowk < 0 else ucnt[row]
        r[2] = cnt_
        r[3] = batch * 3 * B + start_
        r[4] = batch
        last = last_i if rowk < 0 else last_u
        r[5] = -1 if last is None else last[row]
        if last is not None:
            last[row] = batch
        inl = min(int(cnt_), 4)
        r[8:8 + 4 * inl] = pocc[start_:start_ + inl].reshape(-1)
        if occt is not None:
            r[24:24 + inl] = occt[start_:start_ + inl]
    return (pocc, prec) if n_owner <= 0 else (pocc, prec, ohdr)


def triplet_parity(u, i, j, ucnt, icnt):
    """per triplet: parity of u | parity of i << 1 | parity of j << 2 BEFORE this batch's updates (tkr_sample_plan `tpar`)"""
    return ((ucnt[u] & 1) | ((icnt[i] & 1) << 1) | ((icnt[j] & 1) << 2)).astype(np.int32)


def _pack_t(ts):
    """two 16-bit triplet indices per word (what K3 reads; batches above 65,536 keep the low 16 bits, K3 stops at 8,192)"""
    ts = [int(x) & 0xFFFF for x in ts] + [0] * (4 - len(ts))
    words = np.array([ts[0] | (ts[1] << 16), ts[2] | (ts[3] << 16)], dtype=np.uint32).view(np.int32)
    return int(words[0]), int(words[1])


def launch_plan(task, occ, B, occt=None):
    """wave records + header of one batch (see module docstring)."""
    occt = np.zeros(3 * B, dtype=np.int32) if occt is None else occt
    TEAM = team_for(B)
    nblk = max_blocks(B)
    rec = np.zeros((nblk * TEAM, 16), dtype=np.int32)
    live = np.flatnonzero(task[:, 0] != -1)
    light = [t for t in live if task[t, 2] <= light_max(B)]
    heavy = [t for t in live if task[t, 2] > light_max(B)]
    LPB = light_per_block(B)
    nlb = (len(light) + LPB - 1) // LPB
    rec[: nlb * TEAM, 0] = -1
    for li, t in enumerate(light):
        slot = (li // LPB) * TEAM + li % LPB
        rowk, start, cnt, par = task[t]
        rec[slot, 0:4] = (rowk, par | (1 << 8), cnt, start)
        inl = min(cnt, 4)
        rec[slot, 4:4 + 2 * inl] = occ[start:start + inl].reshape(-1)
        rec[slot, 12] = cnt
        rec[slot, 13:15] = _pack_t(occt[start:start + inl])
    for h, t in enumerate(heavy):
        rowk, start, cnt, par = task[t]
        for w in range(TEAM):
            mine = np.arange(start + w, start + cnt, TEAM)
            r = rec[(nlb + h) * TEAM + w]
            r[0:4] = (rowk, par | (TEAM << 8) | (w << 16), len(mine), start + w)
            first = occ[mine[:4]].reshape(-1)
            r[4:4 + len(first)] = first
            r[12] = cnt
            r[13:15] = _pack_t(occt[mine[:4]])
    hdr = np.array([nlb + len(heavy), nlb, len(heavy), len(live)], dtype=np.int32)
    return rec, hdr


def sample_and_plan(tr_users, row_ptr, pos_cols, cols_sorted, n_items, seed, first_triplet,
                    n_batches, B, ucnt=None, icnt=None, n_users=None, n_owner=0):
    """What ``tkr_sample_plan`` produces for n_batches batches: (u,i,j)[n_batches*B],
    task[n_batches,3B,4], occ[n_batches,3B,2], rec[n_batches, max_blocks*TEAM, 16],
    hdr[n_batches,4], occt[n_batches,3B]; ucnt/icnt (int32 update counters) are advanced in place."""
    n_users = n_users if n_users is not None else len(row_ptr) - 1
    ucnt = np.zeros(n_users, dtype=np.int32) if ucnt is None else ucnt
    icnt = np.zeros(n_items, dtype=np.int32) if icnt is None else icnt
    u, i, j = sample_triplets(tr_users, row_ptr, pos_cols, cols_sorted, n_items, seed,
                              first_triplet, n_batches * B)
    tasks = np.zeros((n_batches, 3 * B, 4), dtype=np.int32)
    occs = np.zeros((n_batches, 3 * B, 2), dtype=np.int32)
    recs = np.zeros((n_batches, max_blocks(B) * team_for(B), 16), dtype=np.int32)
    hdrs = np.zeros((n_batches, 4), dtype=np.int32)
    occts = np.zeros((n_batches, 3 * B), dtype=np.int32)
    tpars = np.zeros((n_batches, B), dtype=np.int32)
    poccs = np.zeros((n_batches, 3 * B, 4), dtype=np.int32)
    precs = np.zeros((n_batches, 3 * B, 32), dtype=np.int32)
    raw_tasks = np.zeros((n_batches, 3 * B, 4), dtype=np.int32)
    raw_occs = np.zeros((n_batches, 3 * B, 2), dtype=np.int32)
    last_u, last_i = np.full(n_users, -1, dtype=np.int32), np.full(n_items, -1, dtype=np.int32)
    ohdrs = np.zeros((max(n_owner, 0), n_batches), dtype=np.int32)
    for b in range(n_batches):
        sl = slice(b * B, (b + 1) * B)
        tasks[b], occs[b], occts[b] = plan_batch(u[sl], i[sl], j[sl], return_t=True)
        tpars[b] = triplet_parity(u[sl], i[sl], j[sl], ucnt, icnt)
        raw_tasks[b], raw_occs[b] = tasks[b], occs[b]
        if n_owner > 0:
            poccs[b], precs[b], ohdrs[:, b] = flow_records(tasks[b], occs[b], B, ucnt, icnt, b, last_u, last_i, n_owner, occts[b])
        else:
            poccs[b], precs[b] = flow_records(tasks[b], occs[b], B, ucnt, icnt, b, last_u, last_i, occt=occts[b])
        resolve_parity(tasks[b], occs[b], B, ucnt, icnt)
        recs[b], hdrs[b] = launch_plan(tasks[b], occs[b], B, occts[b])
    sample_and_plan.last_tpars = tpars            # per-triplet parities of the last call (kept off the return tuple)
    sample_and_plan.last_flow = dict(pocc=poccs, prec=precs, task=raw_tasks, occ=raw_occs, ohdr=ohdrs)      # dataflow form of the same plan
    return u, i, j, tasks, occs, recs, hdrs, occts


def vbpr_colplan(f_ptr, f_col, f_val, d, ti, tj, row_cap):
    """The column plan of ONE VBPR batch (include/tkr.h tkr_vbpr_colplan; built beside K1 by csrc/vbpr_cols.hip): which (triplet,
    feature column) pairs the batch touches.  From the CSR of feat (f_ptr / f_col ascending inside a row / f_val) and the batch's
    positive / negative items ti, tj:
      tcnt [B], tent [B][2*row_cap][2]   per triplet (column, value) of f_i's nonzeros (+value) followed by f_j's (-value)
      cent [E][2]                        the same entries as (t, +-value) grouped by column, every run in (t, side) order
      colh [d][8]                        (entries, first entry in cent, the first three entries (t, value bits) inline, zeros beyond)
    Values are returned as float32 with their sign applied; the device stores their bit patterns in int32 words.
    This is what single/vbpr.py:114 feeds per batch -- feat[ib], feat[jb] -- regrouped for the kernels, nothing more."""
    B, tcap = len(ti), 2 * row_cap
    tcnt = np.zeros(B, np.int32)
    tent_c = np.zeros((B, tcap), np.int32)
    tent_v = np.zeros((B, tcap), np.float32)
    runs = [[] for _ in range(d)]
    for t in range(B):
        n = 0
        for side, item in ((0, int(ti[t])), (1, int(tj[t]))):
            a, b = int(f_ptr[item]), int(f_ptr[item + 1])
            b = min(b, a + row_cap)
            for e in range(a, b):
                v = np.float32(-f_val[e]) if side else np.float32(f_val[e])
                tent_c[t, n], tent_v[t, n] = f_col[e], v
                runs[int(f_col[e])].append((t, v))
                n += 1
        tcnt[t] = n
    cent_t = np.array([t for r in runs for t, _ in r], np.int32)
    cent_v = np.array([v for r in runs for _, v in r], np.float32)
    colh_n = np.array([len(r) for r in runs], np.int32)
    colh_beg = np.concatenate([[0], np.cumsum(colh_n)[:-1]]).astype(np.int32)
    return dict(tcnt=tcnt, tent_c=tent_c, tent_v=tent_v, cent_t=cent_t, cent_v=cent_v, colh_n=colh_n, colh_beg=colh_beg)
