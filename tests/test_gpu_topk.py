"""GPU parity of K4 (score + rated mask + top-K) and K5 (hit counting), through the C ABI.

Index work: bit-exact.  Ranked id lists are compared
  * exactly against the golden lists captured from the reference (G4-G7, tie-free by construction),
  * exactly against the oracle on exact-arithmetic inputs (every partial sum representable ->
    summation order cannot matter) including deliberate score ties (canonical tie rule),
  * on generic fp32 inputs wherever the oracle's adjacent score gap exceeds 1e-6 relative.
Every test runs under the three score arithmetics (bound-and-refine = default, bf16-split products on the dense matrix pipe,
fp32 MFMA); test_score_error_against_fp64 states the floating-point tolerance of each, test_refine_is_the_fp32_arithmetic
that the first reproduces the third bit for bit and that both are the fma chain the oracle restates."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ref_np as R


@pytest.fixture(scope='module', params=['bf16x3', 'fp32', 'refine'])
def hip(request):
    """every test runs under all score arithmetics of K4 (include/tkr.h, tkr_topk_set_math); 'bf16x3' was measured and dropped: it
    lives in the lab library (make -C top-k-rec_amd/csrc LAB=1, TKR_HIP_LIB=.../libtkr_hip_lab.so) and is skipped otherwise"""
    import tkr_hip
    assert torch.cuda.is_available()
    tkr_hip.lib()
    if request.param == 'bf16x3' and not tkr_hip.lab():
        pytest.skip('bf16x3 is a lab form (make LAB=1)')
    tkr_hip.set_topk_math(request.param)
    yield tkr_hip
    tkr_hip.set_topk_math(tkr_hip.TOPK_MATH_DEFAULT)


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _oracle_lists(U, V, b, rated_cols, K):
    s = np.dot(U, V.T)
    if b is not None:
        s = s + b.reshape(1, -1)
    return [R.filtered_topk(s[r], set(rated_cols[r]), K, canonical=True) for r in range(len(U))], s


def _gpu_lists(hip, U, V, b, rated_cols, K, user_idx=None, want_scores=False):
    n_rows = len(rated_cols)
    ptr = np.zeros(n_rows + 1, np.int64)
    np.cumsum([len(x) for x in rated_cols], out=ptr[1:])
    flat = np.array([c for x in rated_cols for c in sorted(x)], dtype=np.int32)
    mask, pitch = hip.build_rated_mask(_dev(ptr), _dev(flat), n_rows, len(V))
    out = hip.score_topk(_dev(U), _dev(V), K, bias=None if b is None else _dev(b), mask=mask, mask_pitch=pitch,
                         user_idx=None if user_idx is None else _dev(user_idx.astype(np.int32)), want_scores=want_scores)
    torch.cuda.synchronize()
    return out


def _exact(rng, n, k, lim):
    return rng.integers(-lim, lim + 1, (n, k)).astype(np.float32) / 64.0


@pytest.mark.parametrize('n_rows,n_cols,k,K', [(96, 80, 8, 30), (300, 1000, 50, 30), (1000, 333, 128, 32), (70, 2500, 64, 5),
                                               (33, 40, 100, 1), (257, 95, 200, 30), (5, 17, 3, 30), (1, 1, 16, 4),
                                               (640, 70000, 16, 30), (130, 66000, 128, 30), (40, 70001, 100, 7),
                                               # widths above 256: the k dimension in slabs of 256 (csrc/topk.hip score_topk_slab_kernel);
                                               # 2 slabs to k = 512 (what the trainer goes to), 3 to 768; k % 4 != 0: the scalar staging
                                               (300, 1000, 512, 30), (257, 333, 300, 30), (130, 2500, 260, 7), (200, 700, 510, 30),
                                               (150, 66000, 384, 30), (140, 900, 768, 30), (70, 500, 515, 5),
                                               # above 768: the generic form (one wave per row, csrc/topk.hip score_topk_wide_kernel)
                                               (130, 700, 1024, 30), (40, 2100, 1001, 7)])
def test_exact_arithmetic_lists(hip, n_rows, n_cols, k, K):
    rng = np.random.Generator(np.random.PCG64(n_rows * 7 + n_cols))
    lim = max(1, int(np.sqrt((1 << 23) / k)) // 2)          # k * (lim/64)^2 * 4096 < 2^24: all partial sums exact
    U, V = _exact(rng, n_rows, k, min(lim, 512)), _exact(rng, n_cols, k, min(lim, 512))
    b = (rng.integers(-64, 65, n_cols).astype(np.float32) / 64.0) if n_cols % 2 else None
    rated = [rng.choice(n_cols, int(rng.integers(0, min(n_cols, 60))), replace=False).tolist() for _ in range(n_rows)]
    rated[0] = list(range(n_cols))                           # everything rated -> empty list
    if n_rows > 2:
        rated[1] = list(range(max(0, n_cols - 3)))           # fewer than K unrated columns
        rated[2] = []
    exp, s = _oracle_lists(U, V, b, rated, K)
    ids, scores = _gpu_lists(hip, U, V, b, rated, K, want_scores=True)
    ids, scores = ids.cpu().numpy(), scores.cpu().numpy()
    for r in range(n_rows):
        got = [int(c) for c in ids[r] if c >= 0]
        assert got == exp[r], 'row %d' % r
        assert np.all(ids[r, len(got):] == -1)
        np.testing.assert_array_equal(scores[r, :len(got)], s[r, got])


def test_ties_follow_the_canonical_rule(hip):
    """equal scores: higher column first (stable ascending argsort read backwards)"""
    rng = np.random.Generator(np.random.PCG64(5))
    n_rows, n_cols, k, K = 64, 300, 8, 30
    U = _exact(rng, n_rows, k, 8)
    V = _exact(rng, 20, k, 8)[rng.integers(0, 20, n_cols)]   # only 20 distinct item rows -> massive ties
    V[:, :] = V
    rated = [rng.choice(n_cols, 25, replace=False).tolist() for _ in range(n_rows)]
    U[3] = 0                                                 # all scores +0.0 / -0.0
    exp, _ = _oracle_lists(U, V, None, rated, K)
    ids = _gpu_lists(hip, U, V, None, rated, K).cpu().numpy()
    for r in range(n_rows):
        assert [int(c) for c in ids[r] if c >= 0] == exp[r], 'row %d' % r


def test_generic_fp32_lists_match_outside_near_ties(hip):
    rng = np.random.Generator(np.random.PCG64(11))
    n_rows, n_cols, k, K = 512, 4000, 128, 30
    U = (rng.standard_normal((n_rows, k)) * 0.01).astype(np.float32)
    V = (rng.standard_normal((n_cols, k)) * 0.01).astype(np.float32)
    b = (rng.standard_normal(n_cols) * 0.001).astype(np.float32)
    rated = [rng.choice(n_cols, 50, replace=False).tolist() for _ in range(n_rows)]
    exp, s = _oracle_lists(U, V, b, rated, K)
    s64 = U.astype(np.float64) @ V.astype(np.float64).T + b
    ids, scores = _gpu_lists(hip, U, V, b, rated, K, want_scores=True)
    ids, scores = ids.cpu().numpy(), scores.cpu().numpy()
    same = 0
    for r in range(n_rows):
        np.testing.assert_allclose(scores[r], s64[r, ids[r]], rtol=2e-5, atol=1e-9)   # fp32 dot, K=128
        assert not (set(ids[r].tolist()) & set(rated[r]))
        ref = exp[r]
        # positions whose fp64 score is separated from both neighbours by > 1e-6 relative must agree
        sref = s64[r, ref]
        gap_ok = np.ones(K, bool)
        d = np.abs(np.diff(sref)) > 1e-6 * np.abs(sref[:-1]) + 1e-10
        gap_ok[:-1] &= d
        gap_ok[1:] &= d
        kth_gap = abs(sref[-1] - np.sort(np.delete(s64[r], rated[r]))[-K - 1]) > 1e-6 * abs(sref[-1]) + 1e-10
        if kth_gap:
            assert np.all(ids[r][gap_ok] == np.array(ref)[gap_ok]), 'row %d' % r
        same += int(ids[r].tolist() == ref)
    assert same > 0.95 * n_rows


@pytest.mark.parametrize('n_rows,n_cols,k,bias,wide', [(300, 2000, 128, True, False), (1000, 700, 64, False, False),
                                                       (513, 1500, 100, True, False), (200, 900, 50, True, True),
                                                       (3000, 5000, 128, True, False), (64, 70000, 32, False, False)])
def test_refine_is_the_fp32_arithmetic(n_rows, n_cols, k, bias, wide):
    """Bound-and-refine (the default) returns the ids AND the score bits of the fp32-MFMA arithmetic on generic inputs -- the fp16
    pass only proposes, the fp32 fma chain decides -- and both are the chain oracle/ref_np.mfma_chain_scores restates.  'wide':
    factors over nine decades (the power-of-two scaling of the fp16 pass and its margin must hold there too)."""
    import tkr_hip
    rng = np.random.Generator(np.random.PCG64(n_rows + 3 * n_cols + k))
    U = (rng.standard_normal((n_rows, k)) * 0.01).astype(np.float32)
    V = (rng.standard_normal((n_cols, k)) * 0.01).astype(np.float32)
    if wide:
        U *= (10.0 ** rng.uniform(-4, 5, (n_rows, 1))).astype(np.float32)
        V *= (10.0 ** rng.uniform(-5, 4, (n_cols, k))).astype(np.float32)
    b = (rng.standard_normal(n_cols) * 0.002).astype(np.float32) if bias else None
    K = 30
    rated = [rng.choice(n_cols, 40, replace=False).tolist() for _ in range(n_rows)]
    out = {}
    try:
        for mode in ('refine', 'fp32'):
            tkr_hip.set_topk_math(mode)
            ids, sc = _gpu_lists(tkr_hip, U, V, b, rated, K, want_scores=True)
            out[mode] = (ids.cpu().numpy(), sc.cpu().numpy())
    finally:
        tkr_hip.set_topk_math(tkr_hip.TOPK_MATH_DEFAULT)
    np.testing.assert_array_equal(out['refine'][0], out['fp32'][0])
    np.testing.assert_array_equal(out['refine'][1].view(np.int32), out['fp32'][1].view(np.int32))
    if n_rows * n_cols <= 2_000_000:                          # the oracle's chain, on every returned score
        chain = R.mfma_chain_scores(U, V, b)
        ids, sc = out['fp32']
        exp = np.take_along_axis(chain, ids.astype(np.int64), axis=1)
        same = exp.view(np.int32) == sc.view(np.int32)
        assert same.mean() > 1 - 1e-4                         # float64 double rounding in the oracle, see its docstring
        np.testing.assert_allclose(sc, exp, rtol=3e-7, atol=0)
        # ... and the lists are the oracle's ranking of the chain scores wherever it is free of exact ties at the cut
        for r in range(0, n_rows, 7):
            ref = R.filtered_topk(chain[r], set(rated[r]), K, canonical=True)
            if same[r].all():
                assert ids[r].tolist() == ref, 'row %d' % r


def test_refine_overflowing_lists_fall_back():
    """Lists that cannot hold their margin: 400 exact copies of the row every user likes best (far more than the 64 slots of a
    list) plus 300 near-copies inside the fp16 margin.  The bound-and-refine kernel has to give those user blocks up and the fp32
    kernel redoes them: the result is the oracle's (ties -> higher column first), with and without item-range splitting."""
    import tkr_hip
    rng = np.random.Generator(np.random.PCG64(77))
    n_rows, n_cols, k, K = 700, 3000, 64, 30
    U = (np.abs(rng.standard_normal((n_rows, k))) * 0.01 + 0.001).astype(np.float32)
    V = (rng.standard_normal((n_cols, k)) * 0.002).astype(np.float32)
    best = np.full(k, 0.03, dtype=np.float32)                       # positive against every (positive) user row
    dup = rng.choice(n_cols, 700, replace=False)
    V[dup[:400]] = best
    V[dup[400:]] = best * (1.0 - rng.uniform(0, 2e-4, (300, 1)).astype(np.float32))      # within 2^-10 * |u||v| of the copies
    rated = [rng.choice(n_cols, 20, replace=False).tolist() for _ in range(n_rows)]
    exp, s = _oracle_lists(U, V, None, rated, K)
    chain = R.mfma_chain_scores(U, V, None)
    try:
        tkr_hip.set_topk_math('refine')
        ids = _gpu_lists(tkr_hip, U, V, None, rated, K).cpu().numpy()
        tkr_hip.set_topk_math('fp32')
        ids_f = _gpu_lists(tkr_hip, U, V, None, rated, K).cpu().numpy()
    finally:
        tkr_hip.set_topk_math(tkr_hip.TOPK_MATH_DEFAULT)
    np.testing.assert_array_equal(ids, ids_f)
    for r in range(0, n_rows, 9):                                   # the fma chain's ranking (exact copies tie exactly in any order of summation)
        assert ids[r].tolist() == R.filtered_topk(chain[r], set(rated[r]), K, canonical=True), 'row %d' % r
        assert set(ids[r].tolist()) <= set(dup[:400].tolist())


def test_item_range_split_is_invisible(hip):
    """the launch may split the catalogue into item ranges and merge: identical ids and scores either way"""
    rng = np.random.Generator(np.random.PCG64(21))
    for n_rows, n_cols, k in ((3000, 5000, 64), (70000, 2100, 16), (700, 20000, 128)):
        U = (rng.standard_normal((n_rows, k)) * 0.01).astype(np.float32)
        V = (rng.standard_normal((n_cols, k)) * 0.01).astype(np.float32)
        V[::7] = V[1]                                        # exact ties across item ranges
        ptr = np.arange(0, (n_rows + 1) * 20, 20, dtype=np.int64)
        cols = rng.integers(0, n_cols, n_rows * 20).astype(np.int32)
        mask, pitch = hip.build_rated_mask(_dev(ptr), _dev(cols), n_rows, n_cols)
        a = hip.score_topk(_dev(U), _dev(V), 30, mask=mask, mask_pitch=pitch, want_scores=True, split=True)
        b = hip.score_topk(_dev(U), _dev(V), 30, mask=mask, mask_pitch=pitch, want_scores=True, split=False)
        torch.cuda.synchronize()
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


@pytest.mark.parametrize('K', [33, 50, 64, 100])
def test_large_k_multi_pass(hip, K):
    """evaluate.py -t above 32: exact through repeated launches"""
    rng = np.random.Generator(np.random.PCG64(K))
    n_rows, n_cols, k = 200, 400, 16
    U, V = _exact(rng, n_rows, k, 32), _exact(rng, n_cols, k, 32)
    rated = [rng.choice(n_cols, int(rng.integers(0, 40)), replace=False).tolist() for _ in range(n_rows)]
    rated[0] = list(range(n_cols - 40))                     # fewer than K unrated columns
    exp, s = _oracle_lists(U, V, None, rated, K)
    ids, scores = _gpu_lists(hip, U, V, None, rated, K, want_scores=True)
    ids, scores = ids.cpu().numpy(), scores.cpu().numpy()
    assert ids.shape == (n_rows, K)
    for r in range(n_rows):
        got = [int(c) for c in ids[r] if c >= 0]
        assert got == exp[r], 'row %d' % r
        np.testing.assert_array_equal(scores[r, :len(got)], s[r, got])


def test_user_idx_gather(hip):
    rng = np.random.Generator(np.random.PCG64(2))
    U, V = _exact(rng, 500, 32, 16), _exact(rng, 700, 32, 16)
    pick = rng.choice(500, 130, replace=True)                # repeated users allowed (two test lines, one user)
    rated = [rng.choice(700, 10, replace=False).tolist() for _ in pick]
    exp, _ = _oracle_lists(U[pick], V, None, rated, 30)
    ids = _gpu_lists(hip, U, V, None, rated, 30, user_idx=pick).cpu().numpy()
    for r in range(len(pick)):
        assert ids[r].tolist() == exp[r]


def test_count_hits_matches_rank_walk(hip):
    rng = np.random.Generator(np.random.PCG64(3))
    n_rows, n_cols, K = 200, 120, 30
    ids = np.stack([rng.permutation(n_cols)[:K] for _ in range(n_rows)]).astype(np.int32)
    ids[5, 7:] = -1
    likes = [sorted(rng.choice(n_cols, int(rng.integers(1, 20)), replace=False).tolist()) for _ in range(n_rows)]
    ptr = np.zeros(n_rows + 1, np.int64)
    np.cumsum([len(x) for x in likes], out=ptr[1:])
    flat = np.array([c for x in likes for c in x], dtype=np.int32)
    for step, total in ((5, 30), (3, 10), (1, 4), (7, 30), (40, 30)):
        interval = total // step
        got = hip.count_hits(_dev(ids[:, :total].copy()), _dev(ptr), _dev(flat), step, interval).cpu().numpy()
        exp = np.zeros(interval, np.int64)
        for r in range(n_rows):
            kept = [int(c) for c in ids[r, :total] if c >= 0]
            exp += np.array(R.bucket_hits(kept, set(likes[r]), step, interval), dtype=np.int64)
        np.testing.assert_array_equal(got, exp)


@pytest.mark.parametrize('g,scs', [('g4', ['im', 'om']), ('g5', ['im', 'om']), ('g6', ['all'])])
def test_cli_matches_reference_stdout_and_lists(hip, golden_dir, g, scs, capsys):
    import evaluate as E
    d = os.path.join(golden_dir, g)
    exp = json.load(open(os.path.join(d, 'expected.json')))
    data, model = os.path.join(d, 'data'), os.path.join(d, 'model')
    assert E.main(['-d', data, '-m', model, '-sl'] + scs) == exp['stdout']
    assert capsys.readouterr().out.strip().split('\n') == exp['stdout']
    if 'stdout_s3_t10' in exp:
        assert E.main(['-d', data, '-m', model, '-s', '3', '-t', '10', '-sl', 'om', 'im']) == exp['stdout_s3_t10']
    uids, vids = E.read_ids(os.path.join(data, 'uid')), E.read_ids(os.path.join(data, 'vid'))
    U, V = E.read_matrix(os.path.join(model, 'final-U.dat'), uids), E.read_matrix(os.path.join(model, 'final-V.dat'), vids)
    bp = os.path.join(model, 'final-B.dat')
    b = E.read_matrix(bp, vids) if os.path.exists(bp) else None
    token = {v: t for t, v in uids.items()}
    for sc in scs:
        scen = E.load_scenario(data, 0, sc, uids)
        ids = E.rank_scenario(_dev(U), V, b, vids, scen, 30, torch.device('cuda')).cpu().numpy()
        assert len(scen.users) == len(exp['lists'][sc])
        for u, row in zip(scen.users, ids):
            assert [int(c) for c in row if c >= 0] == exp['lists'][sc][token[int(u)]], (sc, token[int(u)])


def test_cli_total_above_32_matches_oracle(hip, golden_dir):
    import evaluate as E
    d = os.path.join(golden_dir, 'g4')
    data, model = os.path.join(d, 'data'), os.path.join(d, 'model')
    for step, total in ((10, 50), (7, 64)):
        assert E.main(['-d', data, '-m', model, '-s', str(step), '-t', str(total), '-sl', 'im', 'om']) == \
            R.evaluate_cli(data, model, step=step, total=total, scenarios=('im', 'om'))


def test_cli_edge_cases_g7(hip, golden_dir):
    import evaluate as E
    d = os.path.join(golden_dir, 'g7')
    exp = json.load(open(os.path.join(d, 'expected.json')))
    data, model = os.path.join(d, 'data'), os.path.join(d, 'model')
    for run in exp['runs']:
        assert E.main(['-d', data, '-m', model, '-s', str(run['step']), '-t', str(run['total']), '-sl', 'sm']) == run['stdout']


def test_large_shape_properties(hip):
    """Netflix-width catalogue (17,770 items, k=128): sampled rows against the oracle, plus size-independent
    properties on every row: sorted descending, no rated column, no duplicates."""
    rng = np.random.Generator(np.random.PCG64(8))
    n_rows, n_cols, k, K = 20000, 17770, 128, 30
    U = (rng.standard_normal((n_rows, k)) * 0.01).astype(np.float32)
    V = (rng.standard_normal((n_cols, k)) * 0.01).astype(np.float32)
    rated = [rng.choice(n_cols, 100, replace=False).tolist() if r % 50 == 0 else [] for r in range(n_rows)]
    ids, scores = _gpu_lists(hip, U, V, None, rated, K, want_scores=True)
    ids, scores = ids.cpu().numpy(), scores.cpu().numpy()
    assert np.all(ids >= 0) and np.all(np.diff(scores, axis=1) <= 0)
    assert all(len(set(row)) == K for row in ids[::97].tolist())
    for r in range(0, n_rows, 50):
        assert not (set(ids[r].tolist()) & set(rated[r]))
    sample = np.arange(0, n_rows, 400)
    s64 = U[sample].astype(np.float64) @ V.astype(np.float64).T
    for q, r in enumerate(sample):
        s = s64[q].copy()
        s[rated[r]] = -np.inf
        top = np.argsort(-s, kind='stable')[:K]
        np.testing.assert_allclose(scores[r], s[ids[r]], rtol=2e-5, atol=1e-9)
        assert len(set(top.tolist()) & set(ids[r].tolist())) >= K - 1       # at most one near-tie swap at the cut


@pytest.mark.parametrize('k,scale', [(128, 'unit'), (64, 'wide'), (100, 'unit'), (16, 'wide')])
def test_score_error_against_fp64(hip, k, scale):
    """Tolerance of the scores themselves, both arithmetics: |s - s64| <= 1.2e-6 * sum_k |u_k v_k| (k <= 128).
    fp32 has eps = 1.19e-7; a sequential fp32 dot of length k may be off by k*eps/2 (7.6e-6 at k = 128) and typically
    is by ~sqrt(k)*eps; the split kernel drops < 2^-23 |u_k v_k| per product and rounds 6 accumulations per 16 k.
    Its MEAN error must stay within 4x of np.dot(fp32)'s.  'wide': factors spanning nine decades."""
    rng = np.random.Generator(np.random.PCG64(k))
    n_rows, n_cols, K = 256, 2048, 32
    U = rng.standard_normal((n_rows, k)).astype(np.float32)
    V = rng.standard_normal((n_cols, k)).astype(np.float32)
    if scale == 'wide':
        U *= (10.0 ** rng.uniform(-6, 3, (n_rows, k))).astype(np.float32)
        V *= (10.0 ** rng.uniform(-6, 3, (n_cols, k))).astype(np.float32)
    ids, scores = _gpu_lists(hip, U, V, None, [[] for _ in range(n_rows)], K, want_scores=True)
    ids, scores = ids.cpu().numpy(), scores.cpu().numpy()
    s64 = U.astype(np.float64) @ V.astype(np.float64).T
    mag = np.abs(U).astype(np.float64) @ np.abs(V).astype(np.float64).T
    rows = np.arange(n_rows)[:, None]
    err = np.abs(scores.astype(np.float64) - s64[rows, ids]) / mag[rows, ids]
    np_err = np.abs(np.dot(U, V.T).astype(np.float64) - s64)[rows, ids] / mag[rows, ids]
    assert err.max() <= 1.2e-6, err.max()
    assert err.mean() <= 4 * np_err.mean() + 1e-9, (err.mean(), np_err.mean())
    # and the selection itself: the kept set is the true top-K up to scores closer than the tolerance
    for r in range(0, n_rows, 17):
        order = np.argsort(-s64[r], kind='stable')[:K]
        cut = s64[r][order[-1]]
        missing = set(order.tolist()) - set(ids[r].tolist())
        assert all(abs(s64[r][c] - cut) <= 2.4e-6 * mag[r][c] for c in missing), (r, missing)
