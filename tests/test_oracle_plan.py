"""The device-stream oracle (oracle/plan_np.py): Philox known answers, sampler invariants
and distribution (the reference's sampler semantics, single/bpr.py:155-165), plan layout."""
import numpy as np

from oracle import plan_np as P


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32-10
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, exp in kat:
        out = P.philox4x32_10(*[np.array([c], dtype=np.uint32) for c in ctr], *key)
        assert tuple(int(o[0]) for o in out) == exp


def test_mulhi64():
    rng = np.random.Generator(np.random.PCG64(0))
    lo = rng.integers(0, 2**32, 1000, dtype=np.uint64).astype(np.uint32)
    hi = rng.integers(0, 2**32, 1000, dtype=np.uint64).astype(np.uint32)
    for n in (1, 2, 7, 10380, 69878, 2**32 - 1):
        got = P.mulhi64(lo, hi, n)
        exp = [((int(h) << 32 | int(l)) * n) >> 64 for l, h in zip(lo, hi)]
        assert got.tolist() == exp


def _toy(n_users=60, n_items=40, seed=3):
    rng = np.random.Generator(np.random.PCG64(seed))
    tr = {}
    for u in rng.permutation(n_users)[: n_users - 7]:        # some users have no positives
        deg = int(rng.integers(1, 12))
        tr[int(u)] = [int(x) for x in rng.integers(0, n_items, deg)]      # duplicates allowed
    tr[5] = list(range(n_items - 1))                                       # rated all but one item
    return tr, list(tr.keys()), n_items


def test_sampler_invariants_and_distribution():
    tr, tr_users, n_items = _toy()
    row_ptr, pos, srt = P.build_csr(tr, 60)
    u, i, j = P.sample_triplets(tr_users, row_ptr, pos, srt, n_items, seed=99, first_triplet=0, count=60000)
    assert set(u.tolist()) <= set(tr_users)
    for uu, ii, jj in zip(u[:5000], i[:5000], j[:5000]):
        assert ii in tr[uu] and jj not in tr[uu]
    assert np.all(j[u == 5] == n_items - 1)                  # the only legal negative
    # chi-square: users uniform over tr_users
    cnt = np.array([np.sum(u == x) for x in tr_users])
    e = len(u) / len(tr_users)
    assert np.sum((cnt - e) ** 2 / e) < len(tr_users) + 6 * np.sqrt(2 * len(tr_users))
    # positives: uniform over the (duplicate-preserving) list of a user
    uu = max(tr_users, key=lambda x: len(tr[x]) if x != 5 else 0)
    sel = i[u == uu]
    vals, mult = np.unique(tr[uu], return_counts=True)
    obs = np.array([np.sum(sel == v) for v in vals])
    e = len(sel) * mult / mult.sum()
    assert np.sum((obs - e) ** 2 / e) < len(vals) + 6 * np.sqrt(2 * len(vals))
    # negatives: uniform over the complement
    comp = sorted(set(range(n_items)) - set(tr[uu]))
    selj = j[u == uu]
    obs = np.array([np.sum(selj == v) for v in comp])
    e = len(selj) / len(comp)
    assert np.sum((obs - e) ** 2 / e) < len(comp) + 6 * np.sqrt(2 * len(comp))


def test_stream_is_counter_based():
    tr, tr_users, n_items = _toy()
    csr = P.build_csr(tr, 60)
    a = P.sample_triplets(tr_users, *csr, n_items, 7, 0, 1000)
    b = P.sample_triplets(tr_users, *csr, n_items, 7, 400, 100)
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x[400:500], y)
    c = P.sample_triplets(tr_users, *csr, n_items, 8, 0, 1000)
    assert not np.array_equal(a[0], c[0])


def test_plan_layout():
    tr, tr_users, n_items = _toy()
    csr = P.build_csr(tr, 60)
    B = 32
    ucnt, icnt = np.zeros(60, np.int32), np.zeros(n_items, np.int32)
    u, i, j, tasks, occs, recs, hdrs, occts = P.sample_and_plan(tr_users, *csr, n_items, 11, 0, 3, B, ucnt, icnt)
    seen_u, seen_i = np.zeros(60, np.int32), np.zeros(n_items, np.int32)
    for b in range(3):
        ub, ib, jb = (x[b * B:(b + 1) * B] for x in (u, i, j))
        task, occ = tasks[b], occs[b]
        live = task[task[:, 0] != -1]
        rows = live[:, 0] & 0x7fffffff
        kind = (live[:, 0].view(np.uint32) >> 31).astype(int)
        nu = int(np.sum(kind == 0))
        assert np.all(kind[:nu] == 0) and np.all(kind[nu:] == 1)
        assert rows[:nu].tolist() == sorted(set(ub.tolist()))
        assert rows[nu:].tolist() == sorted(set(ib.tolist()) | set(jb.tolist()))
        assert live[:nu, 2].sum() == B and live[nu:, 2].sum() == 2 * B
        assert np.all(task[len(live):, 0] == -1)
        M = 0x3fffffff
        for row, start, cnt, par in live[:nu]:
            ts = np.flatnonzero(ub == row)
            assert par == seen_u[row] & 1
            assert (occ[start:start + cnt, 0] & M).tolist() == ib[ts].tolist()
            assert (occ[start:start + cnt, 1] & M).tolist() == jb[ts].tolist()
            assert ((occ[start:start + cnt, 0] >> 30) & 1).tolist() == (seen_i[ib[ts]] & 1).tolist()
        for rk, start, cnt, par in live[nu:]:
            row = rk & 0x7fffffff
            assert par == seen_i[row] & 1
            exp = [(int(ub[t]), int(jb[t]), 0) for t in np.flatnonzero(ib == row)] + \
                  [(int(ub[t]), int(ib[t]), 1) for t in np.flatnonzero(jb == row)]
            got = [(int(a & M), int(o & M), int((o >> 31) & 1)) for a, o in occ[start:start + cnt]]
            assert got == exp
            assert [int((a >> 30) & 1) for a, _ in occ[start:start + cnt]] == [int(seen_u[e[0]] & 1) for e in exp]
        assert occts[b][:B].tolist() == np.argsort(ub, kind='stable').tolist()
        assert (occts[b][B:] == np.argsort(np.concatenate([ib, jb]), kind='stable') % B).all()
        # launch plan: every task appears once; heavy tasks are split over a team of 16 waves
        rec, hdr = recs[b], hdrs[b]
        TEAM = P.team_for(B)
        n_light = sum(1 for t in live if t[2] <= P.light_max(B))
        nlb = hdr[1]
        assert hdr[3] == len(live) and n_light + hdr[2] == hdr[3] and hdr[0] == nlb + hdr[2]
        assert nlb == (n_light + P.light_per_block(B) - 1) // P.light_per_block(B)
        used = rec[: hdr[0] * TEAM]
        lightrec = used[: nlb * TEAM]
        assert [r for r in lightrec[:, 0] if r != -1] == [t[0] for t in live if t[2] <= P.light_max(B)]
        for h, t in enumerate([t for t in live if t[2] > P.light_max(B)]):
            team = used[(nlb + h) * TEAM:(nlb + h + 1) * TEAM]
            assert np.all(team[:, 0] == t[0]) and team[:, 2].sum() == t[2] and np.all(team[:, 12] == t[2])
            assert ((team[:, 1] >> 16) & 0xff).tolist() == list(range(TEAM))
        seen_u[np.unique(ub)] += 1
        seen_i[np.unique(np.concatenate([ib, jb]))] += 1
    np.testing.assert_array_equal(seen_u, ucnt)
    np.testing.assert_array_equal(seen_i, icnt)
