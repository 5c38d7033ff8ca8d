"""-m gpu: the BPR hot path at BASELINE.json's full MovieLens-10M shape (69,878 users x 10,380 items, k = 128, batch 256,
one reference epoch = 10^6 // 256 = 3,906 batches): the whole epoch against the oracle replaying the same 999,936
triplets, plus the size-independent properties of the path (sampler invariants on every triplet, untouched rows
bit-identical, update counters = number of batches touching a row, run-to-run determinism, lr = 0 leaves parameters
bit-identical while the slots move)."""
import numpy as np
import pytest
import torch

from oracle import plan_np as P
from oracle import ref_np as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ml10m():
    import synth
    r = synth.make_ratings(seed=42, **synth.ML10M)
    row_ptr, pos, srt, tr_users = synth.positives_csr(r)
    return dict(n_users=r['n_users'], n_items=r['n_in'] + r['n_out'], row_ptr=row_ptr, pos=pos, srt=srt, tr_users=tr_users)


def _run(d, hp, nb, B, k, seed):
    from single import _engine
    dev = torch.device('cuda')
    eng = _engine.BprEngine(d['n_users'], d['n_items'], k, hp, dev, seed=seed)
    init = {n: eng.get(n)[0].cpu().numpy() for n in ('U', 'V', 'b')}
    csr = _engine.TrainingCSR.from_arrays(d['row_ptr'], d['pos'], d['tr_users'], dev)
    eng.run_batches(csr, nb, B, want_loss=False)
    torch.cuda.synchronize()
    return eng, init


@pytest.mark.parametrize('k', [128, 50])      # BASELINE.json configs[1] (k = 128) and configs[0]'s width (k = 50, the reference's train.py)
def test_full_epoch_matches_oracle_and_invariants(ml10m, k):
    d, B = ml10m, 256
    nb = 10 ** 6 // B                                                 # single/bpr.py:139: limit // batch_size
    hp = dict(lu=2.5e-3, li=2.5e-3, lj=2.5e-4, lb=0.0, lr=1.0e-4, mode='l2')       # the reference's defaults
    eng, init = _run(d, hp, nb, B, k, seed=2024)
    u, i, j = P.sample_triplets(d['tr_users'], d['row_ptr'], d['pos'], d['srt'], d['n_items'], 2024, 0, nb * B)
    # the last chunk of the device stream is what the oracle sampler says (K1 is bit-exact at full size too)
    tail = eng.plan.u.cpu().numpy()
    last = nb % 512 or 512
    np.testing.assert_array_equal(tail[: last * B], u[(nb - last) * B:])
    # sampler invariants on all 999,936 triplets: i is a train positive of u, j is not
    n_items = d['n_items']
    key_pos = np.sort(np.repeat(np.arange(d['n_users'], dtype=np.int64), np.diff(d['row_ptr'])) * n_items + d['pos'])
    def member(uu, cc):
        q = uu.astype(np.int64) * n_items + cc
        at = np.searchsorted(key_pos, q)
        return (at < len(key_pos)) & (key_pos[np.minimum(at, len(key_pos) - 1)] == q)
    assert member(u, i).all() and not member(u, j).any()
    assert np.isin(u, d['tr_users']).all()
    # update counters = number of batches that touch the row
    ub = np.unique(np.stack([np.repeat(np.arange(nb), B), u], 1), axis=0)
    ib = np.unique(np.stack([np.repeat(np.arange(nb), 2 * B), np.concatenate([i.reshape(nb, B), j.reshape(nb, B)], 1).ravel()], 1), axis=0)
    np.testing.assert_array_equal(eng.cnt.ucnt.cpu().numpy(), np.bincount(ub[:, 1], minlength=d['n_users']))
    np.testing.assert_array_equal(eng.cnt.icnt.cpu().numpy(), np.bincount(ib[:, 1], minlength=n_items))
    # rows never drawn are bit-identical to their initial values
    got = {n: eng.get(n)[0].cpu().numpy() for n in ('U', 'V', 'b')}
    never = eng.cnt.ucnt.cpu().numpy() == 0
    assert never.sum() > 0                                            # ~e^-14 of the users after 10^6 uniform draws
    np.testing.assert_array_equal(got['U'][never], init['U'][never])
    # the oracle replays the epoch on the same triplets
    ref = dict(U=init['U'].copy(), V=init['V'].copy(), b=init['b'].copy(), msU=np.ones_like(init['U']), msV=np.ones_like(init['V']),
               msb=np.ones_like(init['b']))
    for bb in range(nb):
        sl = slice(bb * B, (bb + 1) * B)
        R.bpr_step(ref, u[sl], i[sl], j[sl], hp)
    for n in ('U', 'V', 'b'):
        np.testing.assert_allclose(got[n], ref[n], rtol=2e-4, atol=1e-6, err_msg=n)
        np.testing.assert_allclose(eng.get(n)[1].cpu().numpy(), ref['ms' + n], rtol=2e-4, atol=1e-7, err_msg='ms' + n)
    assert np.abs(got['V'] - init['V']).max() > 1e-4                  # and it did train


def test_full_size_determinism_and_zero_learning_rate(ml10m):
    d, k, B, nb = ml10m, 128, 256, 1024
    hp = dict(lu=2.5e-3, li=2.5e-3, lj=2.5e-4, lb=1e-3, lr=1.0e-2, mode='l2')
    a, _ = _run(d, hp, nb, B, k, seed=5)
    b, _ = _run(d, hp, nb, B, k, seed=5)
    for n in ('U', 'V', 'b'):
        for x, y in zip(a.get(n), b.get(n)):
            assert torch.equal(x, y), n                               # bitwise, parameters and slots
    c, init = _run(d, dict(hp, lr=0.0), nb, B, k, seed=5)
    for n in ('U', 'V', 'b'):
        p, ms = c.get(n)
        np.testing.assert_array_equal(p.cpu().numpy(), init[n])       # lr = 0: no parameter moves ...
    assert float((c.get('V')[1] != 1).float().mean()) > 0.5           # ... but the RMSProp slots of touched rows do


def test_vbpr_full_size_sparse_view(ml10m):
    """BASELINE.json configs[2] shape: d = 20,000 content features (~100 nonzeros per item), k = 128, batch 256 --
    eight batches of the CSR/CSC path against the oracle's dense arithmetic"""
    from single import _engine
    d_, k, B, nb, dfeat = ml10m, 128, 256, 8, 20000
    kh = k // 2
    rng = np.random.Generator(np.random.PCG64(3))
    n_items = d_['n_items']
    feat = np.zeros((n_items, dfeat), dtype=np.float32)
    cols = rng.integers(0, dfeat, (n_items, 100))
    np.put_along_axis(feat, cols, (rng.random((n_items, 100)) + 0.1).astype(np.float32), axis=1)
    feat /= np.linalg.norm(feat, axis=1, keepdims=True)
    hp = dict(lu=2.5e-3, li=2.5e-3, lj=2.5e-4, lb=1e-3, le=1e-3, lr=0.01, mode='l2')
    dev = torch.device('cuda')
    eng = _engine.VbprEngine(d_['n_users'], n_items, k, dfeat, feat, hp, dev, seed=11)
    assert eng.sparse is not None                                    # 0.5 % dense: the sparse view selects itself
    eng.set_dense(cem=(rng.standard_normal((dfeat, kh)) * 0.05).astype(np.float32), icb=(rng.standard_normal(dfeat) * 0.05).astype(np.float32))
    U0 = eng.get('U')[0].cpu().numpy()
    ref = dict(ure=U0[:, :kh].copy(), uce=U0[:, kh:].copy(), ire=eng.get('I')[0].cpu().numpy(), irb=eng.get('irb')[0].cpu().numpy(),
               cem=eng.cem.cpu().numpy(), icb=eng.icb.cpu().numpy())
    for n in list(ref):
        ref['ms_' + n] = np.ones_like(ref[n])
    csr = _engine.TrainingCSR.from_arrays(d_['row_ptr'], d_['pos'], d_['tr_users'], dev)
    loss = eng.run_batches(csr, nb, B).cpu().numpy()
    torch.cuda.synchronize()
    u, i, j = P.sample_triplets(d_['tr_users'], d_['row_ptr'], d_['pos'], d_['srt'], n_items, 11, 0, nb * B)
    ref_loss = [R.vbpr_step(ref, feat, u[b * B:(b + 1) * B], i[b * B:(b + 1) * B], j[b * B:(b + 1) * B], hp) for b in range(nb)]
    tol = dict(rtol=3e-4, atol=2e-5)
    Uc = eng.get('U')[0].cpu().numpy()
    np.testing.assert_allclose(Uc[:, :kh], ref['ure'], err_msg='ure', **tol)
    np.testing.assert_allclose(Uc[:, kh:], ref['uce'], err_msg='uce', **tol)
    np.testing.assert_allclose(eng.get('I')[0].cpu().numpy(), ref['ire'], err_msg='ire', **tol)
    np.testing.assert_allclose(eng.cem.cpu().numpy(), ref['cem'], err_msg='cem', **tol)
    np.testing.assert_allclose(eng.icb.cpu().numpy(), ref['icb'], err_msg='icb', **tol)
    np.testing.assert_allclose(eng.mscem.cpu().numpy(), ref['ms_cem'], rtol=1e-3, atol=1e-6)
    np.testing.assert_allclose(loss, np.array(ref_loss), rtol=2e-4)


def test_vbpr_full_size_dense_dc128(ml10m):
    """the LITERAL reading of BASELINE.json configs[2] ("d=128"): DENSE content features of width 128 at the ML-10M shape, k = 128,
    batch 256 -- eight batches of the column-plan step (every feature column meets every triplet: 512-entry runs split over the
    groups of a workgroup; the pair sums by the last workgroup of the projection) against the oracle (this view was
    benchmarked but parity-tested at toy shape only)"""
    from single import _engine
    d_, k, B, nb, dfeat = ml10m, 128, 256, 8, 128
    kh = k // 2
    rng = np.random.Generator(np.random.PCG64(5))
    n_items = d_['n_items']
    feat = (rng.random((n_items, dfeat)) + 0.1).astype(np.float32)
    feat /= np.linalg.norm(feat, axis=1, keepdims=True)
    hp = dict(lu=2.5e-3, li=2.5e-3, lj=2.5e-4, lb=1e-3, le=1e-3, lr=0.01, mode='l2')
    dev = torch.device('cuda')
    eng = _engine.VbprEngine(d_['n_users'], n_items, k, dfeat, feat, hp, dev, seed=13)
    assert eng.sparse is not None and eng.wants_cols(B)             # a narrow feat takes the gather view + column plan whatever its density
    eng.set_dense(cem=(rng.standard_normal((dfeat, kh)) * 0.05).astype(np.float32), icb=(rng.standard_normal(dfeat) * 0.05).astype(np.float32))
    U0 = eng.get('U')[0].cpu().numpy()
    ref = dict(ure=U0[:, :kh].copy(), uce=U0[:, kh:].copy(), ire=eng.get('I')[0].cpu().numpy(), irb=eng.get('irb')[0].cpu().numpy(),
               cem=eng.cem.cpu().numpy(), icb=eng.icb.cpu().numpy())
    for n in list(ref):
        ref['ms_' + n] = np.ones_like(ref[n])
    csr = _engine.TrainingCSR.from_arrays(d_['row_ptr'], d_['pos'], d_['tr_users'], dev)
    loss = eng.run_batches(csr, nb, B).cpu().numpy()
    torch.cuda.synchronize()
    u, i, j = P.sample_triplets(d_['tr_users'], d_['row_ptr'], d_['pos'], d_['srt'], n_items, 13, 0, nb * B)
    ref_loss = [R.vbpr_step(ref, feat, u[b * B:(b + 1) * B], i[b * B:(b + 1) * B], j[b * B:(b + 1) * B], hp) for b in range(nb)]
    tol = dict(rtol=3e-4, atol=2e-5)
    Uc = eng.get('U')[0].cpu().numpy()
    np.testing.assert_allclose(Uc[:, :kh], ref['ure'], err_msg='ure', **tol)
    np.testing.assert_allclose(Uc[:, kh:], ref['uce'], err_msg='uce', **tol)
    np.testing.assert_allclose(eng.get('I')[0].cpu().numpy(), ref['ire'], err_msg='ire', **tol)
    np.testing.assert_allclose(eng.get('irb')[0].cpu().numpy(), ref['irb'], err_msg='irb', **tol)
    np.testing.assert_allclose(eng.cem.cpu().numpy(), ref['cem'], err_msg='cem', **tol)
    np.testing.assert_allclose(eng.icb.cpu().numpy(), ref['icb'], err_msg='icb', **tol)
    np.testing.assert_allclose(eng.mscem.cpu().numpy(), ref['ms_cem'], rtol=1e-3, atol=1e-6)
    np.testing.assert_allclose(loss, np.array(ref_loss), rtol=2e-4)


def test_netflix_shape_chunk_matches_oracle():
    """BASELINE.json configs[3] shape (480,189 users x 17,770 items, k = 128, batch 256) on one GPU: one full plan chunk (512 batches)
    against the oracle replaying the same 131,072 triplets, sampler invariants on every triplet, counters = batches touching a row"""
    import synth
    from single import _engine
    n_users, n_items, k, B, nb = 480189, 17770, 128, 256, 512
    row_ptr, pos, srt, tr_users = synth.train_csr_shape(n_users, n_items, mean_pos=90.0, seed=43)
    dev = torch.device('cuda')
    hp = dict(lu=2.5e-3, li=2.5e-3, lj=2.5e-4, lb=0.0, lr=1.0e-3, mode='l2')
    eng = _engine.BprEngine(n_users, n_items, k, hp, dev, seed=77)
    init = {n: eng.get(n)[0].cpu().numpy() for n in ('U', 'V', 'b')}
    csr = _engine.TrainingCSR.from_arrays(row_ptr, pos, tr_users, dev)
    eng.run_batches(csr, nb, B, want_loss=False)
    eng.check()
    # the device stream against the oracle sampler: membership by one searchsorted over (user, item) keys
    key_pos = np.repeat(np.arange(n_users, dtype=np.int64), np.diff(row_ptr)) * n_items + srt          # srt: ascending inside a row
    u = eng.plan.u.cpu().numpy()[: nb * B].astype(np.int64)
    i = eng.plan.i.cpu().numpy()[: nb * B].astype(np.int64)
    j = eng.plan.j.cpu().numpy()[: nb * B].astype(np.int64)

    def member(uu, cc):
        q = uu * n_items + cc
        at = np.minimum(np.searchsorted(key_pos, q), len(key_pos) - 1)
        return key_pos[at] == q
    assert member(u, i).all() and not member(u, j).any() and (np.diff(row_ptr)[u] > 0).all()
    first = 4096                                                       # bit-exact against the oracle sampler on the head of the stream
    ou, oi, oj = P.sample_triplets(tr_users, row_ptr, pos, srt, n_items, 77, 0, first)
    assert np.array_equal(u[:first], ou) and np.array_equal(i[:first], oi) and np.array_equal(j[:first], oj)
    ub = np.unique(np.stack([np.repeat(np.arange(nb), B), u], 1), axis=0)
    ib = np.unique(np.stack([np.repeat(np.arange(nb), 2 * B), np.concatenate([i.reshape(nb, B), j.reshape(nb, B)], 1).ravel()], 1), axis=0)
    np.testing.assert_array_equal(eng.cnt.ucnt.cpu().numpy(), np.bincount(ub[:, 1], minlength=n_users))
    np.testing.assert_array_equal(eng.cnt.icnt.cpu().numpy(), np.bincount(ib[:, 1], minlength=n_items))
    # the oracle replays the chunk on the device's triplets (sparse: only touched rows move)
    ref = dict(U=init['U'].copy(), V=init['V'].copy(), b=init['b'].copy(), msU=np.ones_like(init['U']), msV=np.ones_like(init['V']),
               msb=np.ones_like(init['b']))
    for bb in range(nb):
        sl = slice(bb * B, (bb + 1) * B)
        R.bpr_step(ref, u[sl].astype(np.int32), i[sl].astype(np.int32), j[sl].astype(np.int32), hp)
    got = {n: eng.get(n)[0].cpu().numpy() for n in ('U', 'V', 'b')}
    for n in ('U', 'V', 'b'):
        np.testing.assert_allclose(got[n], ref[n], rtol=2e-4, atol=1e-6, err_msg=n)
    untouched = np.setdiff1d(np.arange(n_users), u)
    np.testing.assert_array_equal(got['U'][untouched], init['U'][untouched])
    assert np.abs(got['V'] - init['V']).max() > 1e-4


def test_topk_full_netflix_shape_properties():
    """BASELINE.json configs[4] at FULL size on one GPU: top-30 of all 17,770 items for all 480,189 users (k = 128), every user with a
    rated mask.  Size-independent properties on EVERY row (sorted, no rated column, no duplicate, no padding), the checksum of the
    result against a second run (the kernel is deterministic) and against the fp32 arithmetic mode on the rows where that is decided
    (gap above 1e-6 relative), and 1,200 sampled rows against the fp64 oracle."""
    import tkr_hip
    n_users, n_items, k, K, deg = 480189, 17770, 128, 30, 60
    dev = torch.device('cuda')
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    U = (torch.randn((n_users, k), device=dev, generator=g) * 0.01 * 1e6).round() / 1e6       # '%f'-rounded factors, like the CLI's input
    V = (torch.randn((n_items, k), device=dev, generator=g) * 0.01 * 1e6).round() / 1e6
    ptr = torch.arange(0, (n_users + 1) * deg, deg, dtype=torch.int64, device=dev)
    cols = torch.randint(0, n_items, (n_users * deg,), device=dev, generator=g, dtype=torch.int32)
    cols = torch.sort(cols.view(n_users, deg), dim=1).values.contiguous().view(-1)
    mask, pitch = tkr_hip.build_rated_mask(ptr, cols, n_users, n_items)
    ids, scores = tkr_hip.score_topk(U, V, K, mask=mask, mask_pitch=pitch, want_scores=True)
    ids2 = tkr_hip.score_topk(U, V, K, mask=mask, mask_pitch=pitch)
    assert torch.equal(ids, ids2)                                                       # run-to-run identical
    assert bool((ids >= 0).all()) and bool((ids < n_items).all())
    assert bool((scores[:, 1:] <= scores[:, :-1]).all())                                # descending on every row
    srt = torch.sort(ids, dim=1).values
    assert bool((srt[:, 1:] != srt[:, :-1]).all())                                      # no duplicate column in any row
    rated = cols.view(n_users, deg).long()
    hit = torch.zeros(n_users, dtype=torch.bool, device=dev)
    for c in range(K):                                                                  # no rated column in any row: searchsorted per position
        at = torch.searchsorted(rated, ids[:, c:c + 1].long()).clamp(max=deg - 1)
        hit |= rated.gather(1, at).squeeze(1) == ids[:, c].long()
    assert not bool(hit.any())
    # sampled rows against fp64
    sample = torch.arange(0, n_users, 401, device=dev)
    s64 = U[sample].double() @ V.double().T
    s64.scatter_(1, rated[sample], float('-inf'))
    top_s, top_i = torch.topk(s64, K + 1, dim=1)
    got = ids[sample].long()
    np.testing.assert_allclose(scores[sample].cpu().numpy(), s64.gather(1, got).cpu().numpy(), rtol=2e-5, atol=1e-9)
    decided = ((top_s[:, :-1] - top_s[:, 1:]) > 1e-6 * top_s[:, :-1].abs()).all(1)      # rows without a near-tie inside or at the cut
    assert int(decided.sum()) > len(sample) // 2
    assert torch.equal(got[decided], top_i[decided, :K])
    # the default arithmetic (bound-and-refine) IS the fp32-MFMA arithmetic: all 480,189 x 30 ids and score bits
    # agree; the bf16-split arithmetic differs only in summation order (same rows wherever the cut is decided)
    try:
        tkr_hip.set_topk_math('fp32')
        ids_f, sc_f = tkr_hip.score_topk(U, V, K, mask=mask, mask_pitch=pitch, want_scores=True)
        ids_b = None
        if tkr_hip.lab():
            tkr_hip.set_topk_math('bf16x3')
            ids_b = tkr_hip.score_topk(U, V, K, mask=mask, mask_pitch=pitch)
    finally:
        tkr_hip.set_topk_math(tkr_hip.TOPK_MATH_DEFAULT)
    assert torch.equal(ids_f, ids) and torch.equal(sc_f.view(torch.int32), scores.view(torch.int32))
    if ids_b is not None:
        assert torch.equal(ids_b[sample][decided], got[decided])
        assert float((ids_b == ids).all(1).float().mean()) > 0.97
