"""GPU parity of K1 (sampler + planner) and K2 (BPR step) against the oracle, through the C ABI.

K1 is integer work: bit-exact against oracle/plan_np.py.  K2 is fp32: the tables after N
sequential mini-batches agree with oracle/ref_np.bpr_step on the same init and the same
(u,i,j) stream within the tolerance stated in each test."""
import ctypes as C
import os

import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from oracle import plan_np as P
from oracle import ref_np as R


@pytest.fixture(scope='module')
def hip():
    import tkr_hip
    assert torch.cuda.is_available(), 'GPU tests need a MI355X'
    tkr_hip.lib()
    return tkr_hip


def _toy(n_users, n_items, seed, max_deg=12, all_but_one=True):
    rng = np.random.Generator(np.random.PCG64(seed))
    tr = {}
    for u in rng.permutation(n_users)[: max(1, n_users - n_users // 8)]:
        tr[int(u)] = [int(x) for x in rng.integers(0, n_items, int(rng.integers(1, max_deg)))]
    if all_but_one:
        tr[5] = list(range(n_items - 1))
    return tr, list(tr.keys())


def _dev(a, dt=torch.int32):
    return torch.from_numpy(np.ascontiguousarray(a)).to('cuda', dt)


def _run_plan(hip, tr, tr_users, n_users, n_items, seed, first, nb, B, chunks=1):
    """run K1 for `chunks` consecutive calls of nb batches -> (got, exp) of the LAST call, plus objects"""
    from single import _engine
    row_ptr, pos, srt = P.build_csr(tr, n_users)
    csr = _engine.TrainingCSR.from_arrays(row_ptr, pos, np.asarray(tr_users, np.int32), torch.device('cuda'))
    np.testing.assert_array_equal(csr.cols_sorted.cpu().numpy(), srt)
    cnt = _engine.UpdateCounters(n_users, n_items, torch.device('cuda'))
    plan = _engine.PlanBuffers(nb, B, torch.device('cuda'))
    plan.rec.zero_()
    ucnt, icnt = np.zeros(n_users, np.int32), np.zeros(n_items, np.int32)
    for c in range(chunks):
        hip.sample_plan(csr, n_users, n_items, seed, first + c * nb * B, nb, B, cnt, plan)
        exp = P.sample_and_plan(tr_users, row_ptr, pos, srt, n_items, seed, first + c * nb * B, nb, B, ucnt, icnt)
    torch.cuda.synchronize()
    got = [plan.u.cpu().numpy(), plan.i.cpu().numpy(), plan.j.cpu().numpy(),
           plan.task.cpu().numpy().reshape(nb, 3 * B, 4), plan.occ.cpu().numpy().reshape(nb, 3 * B, 2),
           plan.rec.cpu().numpy().reshape(nb, -1, 16), plan.hdr.cpu().numpy().reshape(nb, 4),
           plan.occt.cpu().numpy().reshape(nb, 3 * B)]
    np.testing.assert_array_equal(plan.tpar.cpu().numpy()[: nb * B].reshape(nb, B), P.sample_and_plan.last_tpars, err_msg='tpar')
    np.testing.assert_array_equal(cnt.ucnt.cpu().numpy(), ucnt)
    np.testing.assert_array_equal(cnt.icnt.cpu().numpy(), icnt)
    assert int(cnt.touch_u.abs().sum()) == 0 and int(cnt.touch_i.abs().sum()) == 0
    return got, exp, plan


@pytest.mark.parametrize('n_users,n_items,B,nb,chunks', [(60, 40, 32, 5, 1), (300, 150, 256, 7, 3), (300, 150, 100, 3, 2),
                                                         (5000, 900, 1024, 3, 1), (5000, 900, 8192, 2, 2), (40, 30, 1, 4, 1),
                                                         (300, 150, 64, 512, 2),
                                                         # 1024 < batch <= 16,384: the counting planner (csrc/planner_mid.hip) -- several user ranges;
                                                         # runs of ~55 / ~140 occurrences (ranked by a wave) and of > 512 (the bitmap form)
                                                         (5000, 900, 2048, 3, 2), (70000, 10000, 4096, 3, 2), (300, 150, 4096, 2, 2),
                                                         (40, 30, 2048, 2, 1), (40, 6, 4096, 2, 1), (40, 30, 16384, 1, 2), (5000, 900, 1025, 2, 1),
                                                         # ... 512 batches over 150 items do not fit its batch-major touch maps: the row-major lookups
                                                         (300, 150, 1100, 512, 1), (70000, 10000, 2048, 70, 2),
                                                         # ... the Netflix shape's 59 user ranges per batch
                                                         (480189, 17770, 8192, 2, 1),
                                                         # 16,384 < batch <= 65,536: its wide form (run lists in the workspace); above: the grid-wide planner
                                                         # (csrc/planner_big.hip)
                                                         (40, 6, 32768, 1, 1), (300, 150, 65536, 1, 2), (70000, 10000, 32768, 2, 1),
                                                         (5000, 900, 8193, 3, 2), (20000, 3000, 16384, 2, 2), (50000, 9000, 65536, 2, 1),
                                                         (300, 150, 20000, 2, 1)])
def test_sample_plan_bit_exact(hip, n_users, n_items, B, nb, chunks):
    tr, tr_users = _toy(n_users, n_items, seed=n_users + B)
    got, exp, _ = _run_plan(hip, tr, tr_users, n_users, n_items, seed=0x1234567890ABCDEF, first=(1 << 33) + 17, nb=nb,
                            B=B, chunks=chunks)
    for name, g, e in zip(('u', 'i', 'j', 'task', 'occ', 'rec', 'hdr', 'occt'), got, exp):
        if name == 'rec':                       # only the used workgroups are defined
            for b in range(nb):
                used = exp[6][b, 0] * P.team_for(B)
                np.testing.assert_array_equal(g[b, :used], e[b, :used], err_msg='rec batch %d' % b)
        else:
            np.testing.assert_array_equal(g, e, err_msg=name)


def test_negative_draw_membership_at_every_search_width(hip):
    """the membership search of the negative draw (csrc/sampler_draw.h is_member, a k-ary search) on rows of every length around
    powers of two and of four, and on rows with few free items so that most candidates ARE members: (u, i, j) equal the oracle's"""
    from single import _engine
    n_items = 9000
    rng = np.random.Generator(np.random.PCG64(7))
    degs = [1, 2, 3, 15, 16, 17, 31, 32, 33, 34, 63, 64, 65, 511, 512, 513, 528, 529, 1000, 4095, 4096, 4097, 8191, 8192, 8193, 8990, 8999]
    tr = {u: [int(x) for x in rng.permutation(n_items)[:d]] for u, d in enumerate(degs)}
    # ... and rows that are dense runs with single gaps (candidates hit the pivots themselves)
    tr[len(degs)] = [x for x in range(n_items) if x % 16 != 0]
    tr[len(degs) + 1] = [x for x in range(n_items) if x % 563 != 1]
    tr_users = list(tr.keys())
    n_users = len(tr_users)
    row_ptr, pos, srt = P.build_csr(tr, n_users)
    dev = torch.device('cuda')
    csr = _engine.TrainingCSR.from_arrays(row_ptr, pos, np.asarray(tr_users, np.int32), dev)
    B, nb = 512, 8
    plan = _engine.PlanBuffers(nb, B, dev)
    hip.sample_plan(csr, n_users, n_items, 99, 12345, nb, B, _engine.UpdateCounters(n_users, n_items, dev), plan)
    u, i, j = P.sample_triplets(tr_users, row_ptr, pos, srt, n_items, 99, 12345, nb * B)
    np.testing.assert_array_equal(plan.u.cpu().numpy().reshape(-1), u)
    np.testing.assert_array_equal(plan.i.cpu().numpy().reshape(-1), i)
    np.testing.assert_array_equal(plan.j.cpu().numpy().reshape(-1), j)
    assert len(set(u.tolist())) == n_users            # every row length was drawn from


def test_sample_plan_ctl_offset(hip):
    """the device-side batch base (ctl) walks the same stream as first_triplet does"""
    from single import _engine
    tr, tr_users = _toy(200, 90, seed=1)
    row_ptr, pos, srt = P.build_csr(tr, 200)
    B, nb = 64, 4
    dev = torch.device('cuda')
    csr = _engine.TrainingCSR.from_arrays(row_ptr, pos, np.asarray(tr_users, np.int32), dev)
    ctl = torch.tensor([5], dtype=torch.int64, device='cuda')
    plan = _engine.PlanBuffers(nb, B, dev)
    hip.sample_plan(csr, 200, 90, 77, 1000, nb, B, _engine.UpdateCounters(200, 90, dev), plan, ctl=ctl)
    u, i, j = P.sample_triplets(tr_users, row_ptr, pos, srt, 90, 77, 1000 + 5 * B, nb * B)
    np.testing.assert_array_equal(plan.u.cpu().numpy(), u)
    np.testing.assert_array_equal(plan.j.cpu().numpy(), j)


def _state_struct(hip, T, n_users, n_items, k, hp):
    st = hip.BprState()
    st.U, st.msU = T['U'].data_ptr(), T['msU'].data_ptr()
    st.V, st.msV, st.b, st.msb = T['V'].data_ptr(), T['msV'].data_ptr(), T['b'].data_ptr(), T['msb'].data_ptr()
    st.n_users, st.n_items, st.k = n_users, n_items, k
    st.mode = 0 if hp['mode'] == 'l2' else 1
    st.lu, st.li, st.lj, st.lb, st.lr = hp['lu'], hp['li'], hp['lj'], hp['lb'], hp['lr']
    st.rho, st.eps = 0.9, 1e-10
    st.opt = 1 if hp.get('opt') == 'sgd' else 0
    return st


def _tables(ref, n_users, n_items, k):
    T = {}
    for name, n, kk in (('U', n_users, k), ('V', n_items, k), ('b', n_items, 0)):
        shape = (2, n, kk) if kk else (2, n)
        T[name] = torch.zeros(shape, device='cuda')
        T[name][0] = torch.from_numpy(ref[name]).cuda()
        T['ms' + name] = torch.ones(shape, device='cuda')
    return T


def _current(T, name, cnt):
    sel = (torch.from_numpy(cnt).cuda() & 1).long()
    idx = torch.arange(T[name].shape[1], device='cuda')
    return T[name][sel, idx].cpu().numpy()


@pytest.mark.parametrize('k,B,nb,mode,lr', [(16, 64, 12, 'l2', 0.05), (128, 256, 10, 'l2', 0.05), (300, 256, 5, 'l2', 0.05), (512, 1024, 3, 'l1', 0.05),
                                           (50, 256, 6, 'l1', 0.05), (200, 128, 4, 'l2', 1e-4),
                                           (128, 2048, 3, 'l2', 0.05), (64, 4096, 9, 'l2', 0.05), (128, 8192, 8, 'l1', 0.05),
                                           (128, 16384, 3, 'l2', 0.05), (64, 65536, 2, 'l1', 0.05), (32, 1048576, 2, 'l2', 0.05)])
def test_bpr_step_parity(hip, k, B, nb, mode, lr):
    n_users, n_items = (400, 120) if B <= 8192 else (20000, 3000)      # small tables: many in-batch duplicate rows (hundreds per row above 8192)
    tr, tr_users = _toy(n_users, n_items, seed=k + B, all_but_one=False)
    rng = np.random.Generator(np.random.PCG64(k))
    ref = R.init_bpr_state(n_users, n_items, k, rng)
    ref['b'][:] = (rng.standard_normal(n_items) * 0.01).astype(np.float32)
    hp = dict(lu=2.5e-3, li=2.5e-3, lj=2.5e-4, lb=1e-3, lr=lr, mode=mode)
    T = _tables(ref, n_users, n_items, k)
    got, exp, plan = _run_plan(hip, tr, tr_users, n_users, n_items, seed=42, first=0, nb=nb, B=B)
    loss = torch.zeros(nb, device='cuda')
    hip.bpr_run(_state_struct(hip, T, n_users, n_items, k, hp), plan, B, nb, loss)
    torch.cuda.synchronize()
    ref_loss = []
    u, i, j = exp[0], exp[1], exp[2]
    ucnt, icnt = np.zeros(n_users, np.int32), np.zeros(n_items, np.int32)
    for b in range(nb):
        sl = slice(b * B, (b + 1) * B)
        ref_loss.append(R.bpr_step(ref, u[sl], i[sl], j[sl], hp))
        ucnt[np.unique(u[sl])] += 1
        icnt[np.unique(np.concatenate([i[sl], j[sl]]))] += 1
    # fp32 tolerance: |dP| per step <= lr/sqrt(0.1) ~ 0.16 at lr=0.05, compared at 1e-5 abs + 2e-4 rel.  At batch 2^20 an item row
    # sums ~700 occurrence gradients of both signs: the oracle adds them one by one in batch order, the kernel in 16 interleaved partial
    # sums, and what cancels to ~1e-3 of the terms carries the difference of the two orders (4 of 3,000 biases at 5e-5): 1e-4 abs there
    atol = 1e-5 if B < (1 << 20) else 1e-4
    for name, cnt in (('U', ucnt), ('V', icnt), ('b', icnt)):
        np.testing.assert_allclose(_current(T, name, cnt), ref[name], rtol=2e-4, atol=atol, err_msg=name)
        np.testing.assert_allclose(_current(T, 'ms' + name, cnt), ref['ms' + name], rtol=2e-4 if B < (1 << 20) else 5e-4, atol=1e-7,
                                   err_msg='ms' + name)
    # the reported loss is one fp32 atomic add per row task: a million terms of ~0.7 keep 3-4 digits
    np.testing.assert_allclose(loss.cpu().numpy(), np.array(ref_loss), rtol=1e-4 if B < (1 << 20) else 1e-3)
    # rows never sampled keep their initial value in buffer 0 and an untouched buffer 1
    assert np.all(T['U'][1].cpu().numpy()[ucnt == 0] == 0)


def test_bpr_step_is_deterministic(hip):
    """same plan, same init -> bitwise identical tables (no float atomics on the parameters)"""
    n_users, n_items, k, B, nb = 300, 80, 128, 256, 8
    tr, tr_users = _toy(n_users, n_items, seed=9, all_but_one=False)
    hp = dict(lu=2.5e-3, li=2.5e-3, lj=2.5e-4, lb=0.0, lr=0.01, mode='l2')
    _, exp, plan = _run_plan(hip, tr, tr_users, n_users, n_items, seed=3, first=0, nb=nb, B=B)
    outs = []
    for _ in range(2):
        ref = R.init_bpr_state(n_users, n_items, k, np.random.Generator(np.random.PCG64(0)))
        T = _tables(ref, n_users, n_items, k)
        hip.bpr_run(_state_struct(hip, T, n_users, n_items, k, hp), plan, B, nb, None)
        torch.cuda.synchronize()
        outs.append([T[n].cpu().numpy() for n in ('U', 'V', 'b', 'msU', 'msV', 'msb')])
    for a, b in zip(*outs):
        np.testing.assert_array_equal(a, b)


def test_abi_rejects_bad_arguments(hip):
    st = hip.BprState()
    assert hip.lib().tkr_bpr_run(C.byref(st), None, None, None, 256, 1, None, None) == -1
    args = [None] * 24
    assert hip.lib().tkr_sample_plan(None, 0, None, None, None, 5, 10, 0, 0, None, 1, 256, *([None] * 15), None, C.c_int64(0), None) == -1
    assert hip.lib().tkr_sample_plan(None, 1, None, None, None, 5, 10, 0, 0, None, 513, 256, *([None] * 15), None, C.c_int64(0), None) == -2
    # batches up to 1024 plan inside one workgroup; above, a few KB of range sums (csrc/planner_mid.hip); from 4096 on also the key arrays of the
    # grid-wide planner (csrc/planner_big.hip: what takes over when the counting planner cannot)
    assert hip.lib().tkr_plan_workspace_bytes(1024, 128) == 0 and 0 < hip.lib().tkr_plan_workspace_bytes(2048, 128) <= 128 * 128 * 16
    assert hip.plan_workspace_bytes(16384, 4) > 16384 * 4 * 8 * 6


@pytest.mark.parametrize('k,B,nb', [(16, 64, 12), (128, 256, 10), (50, 512, 5), (128, 2048, 3), (64, 2048, 9)])
def test_sgd_step_parity(hip, k, B, nb):
    """tkr_bpr_state.opt = 1: the legacy update P -= lr*g (old/methods/bpr.py:57-61), no slot traffic at all --
    the ms pointers are NULL"""
    n_users, n_items = 400, 120
    tr, tr_users = _toy(n_users, n_items, seed=k + B + 1, all_but_one=False)
    rng = np.random.Generator(np.random.PCG64(k + 1))
    ref = R.init_bpr_state(n_users, n_items, k, rng)
    ref['b'][:] = (rng.standard_normal(n_items) * 0.01).astype(np.float32)
    hp = dict(lu=2.5e-3, li=2.5e-3, lj=2.5e-4, lb=1e-3, lr=0.05, mode='l2', opt='sgd')
    T = _tables(ref, n_users, n_items, k)
    _, exp, plan = _run_plan(hip, tr, tr_users, n_users, n_items, seed=43, first=0, nb=nb, B=B)
    st = _state_struct(hip, T, n_users, n_items, k, hp)
    st.msU = st.msV = st.msb = None
    loss = torch.zeros(nb, device='cuda')
    hip.bpr_run(st, plan, B, nb, loss)
    torch.cuda.synchronize()
    u, i, j = exp[0], exp[1], exp[2]
    ucnt, icnt = np.zeros(n_users, np.int32), np.zeros(n_items, np.int32)
    ref_loss = []
    for b in range(nb):
        sl = slice(b * B, (b + 1) * B)
        ref_loss.append(R.bpr_step(ref, u[sl], i[sl], j[sl], hp))
        ucnt[np.unique(u[sl])] += 1
        icnt[np.unique(np.concatenate([i[sl], j[sl]]))] += 1
    for name, cnt in (('U', ucnt), ('V', icnt), ('b', icnt)):
        np.testing.assert_allclose(_current(T, name, cnt), ref[name], rtol=2e-4, atol=1e-5, err_msg=name)
        assert torch.all(T['ms' + name] == 1.0)                       # never written
    np.testing.assert_allclose(loss.cpu().numpy(), np.array(ref_loss), rtol=1e-4)
    # an RMSProp state without slots is rejected, an unknown optimiser too
    st.opt = 0
    assert hip.lib().tkr_bpr_run(C.byref(st), plan.rec.data_ptr(), plan.occ.data_ptr(), plan.hdr.data_ptr(), B, 1, None, None) == -1
    st.opt = 7
    assert hip.lib().tkr_bpr_run(C.byref(st), plan.rec.data_ptr(), plan.occ.data_ptr(), plan.hdr.data_ptr(), B, 1, None, None) == -1


def test_legacy_bpr_api(hip, capfd):
    """old/methods/bpr.py surface: BPR(K, users, items, ...).train(data, epochs, batch_size); W/H/B.get_value()"""
    from old.methods.bpr import BPR as LegacyBPR
    rng = np.random.Generator(np.random.PCG64(5))
    users = {1000 + 3 * x: x for x in range(300)}
    items = {50 + x: x for x in range(90)}
    P, Q = rng.standard_normal((300, 4)), rng.standard_normal((90, 4))
    data = []
    for uid, ux in users.items():
        top = np.argsort(-(P[ux] @ Q.T))[:8]
        data += [(uid, 50 + int(t)) for t in top]
    m = LegacyBPR(16, users, items, learning_rate=0.05, seed=11)
    W0, H0 = m.W.get_value().copy(), m.H.get_value().copy()
    assert W0.shape == (300, 16) and H0.shape == (90, 16) and np.all(m.B.get_value() == 0)
    assert abs(W0.std() - 0.01) < 2e-3
    m.train(data, epochs=20, batch_size=256)
    err = capfd.readouterr().err
    assert 'Generating 48000 random training samples' in err and 'Total training time' in err
    n_batches = (len(data) * 20 - 1) // 256                           # old/methods/bpr.py:72
    assert m._engine.triplets_drawn == n_batches * 256 and m.losses is not None
    W, H, B = m.W.get_value(), m.H.get_value(), m.B.get_value()
    assert np.isfinite(W).all() and np.isfinite(H).all() and not np.array_equal(W, W0)
    # the model learned the planted preferences: positives outrank the rest for most users
    s = W @ H.T + B
    auc = []
    for uid, ux in users.items():
        pos = sorted({items[i] for u, i in data if u == uid})
        neg = np.setdiff1d(np.arange(90), pos)
        auc.append((s[ux][pos][:, None] > s[ux][neg][None, :]).mean())
    assert np.mean(auc) > 0.8
    # small data: batch size clipped with the reference's warning (old/methods/bpr.py:64-66)
    m2 = LegacyBPR(8, users, items, seed=1)
    m2.train(data[:40], epochs=3, batch_size=256)
    assert 'switching to a batch size of 40' in capfd.readouterr().err
    m2.W.set_value(np.ones((300, 8), np.float32))
    assert np.all(m2.W.get_value() == 1.0)


@pytest.mark.parametrize('k,B,nb', [(300, 256, 6), (600, 256, 4), (1000, 128, 3), (300, 2048, 2), (700, 2048, 2)])
def test_wide_factors_through_the_class(tmp_path, k, B, nb):
    """single/bpr.py:20 takes any k: BPR(k).train runs at every width -- plain tables + K2 up to k = 512 (256 above batch 1024),
    the generic row form (csrc/bpr_step.hip bpr_wide_kernel) beyond, announced by one warning -- and equals the oracle on its stream"""
    sys.path.insert(0, os.path.join(ROOT, 'top-k-rec_amd'))
    import warnings
    import synth
    import tkr_hip
    from single import BPR, _engine
    r = synth.make_ratings(150, 60, 0, seed=13, mu=2.6, sigma=0.4, min_r=4, max_r=25)
    data = str(tmp_path / 'data')
    synth.write_dataset(data, r)
    m = BPR(k=k, lr=0.02, lambda_b=1e-3)
    m.load_training_data(data + '/uid', data + '/vid', data + '/f0tr.txt')
    rng = np.random.Generator(np.random.PCG64(0))
    init = [(rng.standard_normal((m.n_users, k)) * 0.1).astype(np.float32), (rng.standard_normal((m.n_items, k)) * 0.1).astype(np.float32),
            np.zeros((m.n_items, 1), np.float32)]
    m.fue, m.fie, m.fib = (a.copy() for a in init)
    generic = k > 512 or (k > 256 and B > 1024)
    with warnings.catch_warnings(record=True) as seen:
        warnings.simplefilter('always')
        m.train(epochs=1, batch_size=B, epoch_sample_limit=B * nb, seed=11, verbose=False)
    assert any('generic row form' in str(w.message) for w in seen) == generic
    assert m._eng.layout == 'bulk'
    hp = dict(lu=m.lu, li=m.li, lj=m.lj, lb=m.lb, lr=0.02, mode='l2')
    row_ptr, pos, srt = P.build_csr(m.tr_data, m.n_users)
    st = dict(U=init[0].copy(), V=init[1].copy(), b=init[2].ravel().copy(), msU=np.ones_like(init[0]), msV=np.ones_like(init[1]),
              msb=np.ones(m.n_items, np.float32))
    u, i, j = P.sample_triplets(m.tr_users, row_ptr, pos, srt, m.n_items, 11, 0, nb * B)
    last = None
    for s in range(nb):
        last = R.bpr_step(st, u[s * B:(s + 1) * B], i[s * B:(s + 1) * B], j[s * B:(s + 1) * B], hp)
    np.testing.assert_allclose(m.fue, st['U'], rtol=2e-4, atol=1e-5)
    np.testing.assert_allclose(m.fie, st['V'], rtol=2e-4, atol=1e-5)
    np.testing.assert_allclose(m.fib.ravel(), st['b'], rtol=2e-4, atol=1e-5)
    np.testing.assert_allclose(m.last_epoch_loss, float(last), rtol=1e-4)
    if k != 300 or B != 256:
        return
    dev = torch.device('cuda')
    # ... and what the trainer holds, the scorer ranks (evaluate.py:78 on the k = 300 model above)
    ids = tkr_hip.score_topk(torch.from_numpy(m.fue).to(dev), torch.from_numpy(m.fie).to(dev), 5, bias=torch.from_numpy(m.fib.ravel().copy()).to(dev))
    s = m.fue.astype(np.float64) @ m.fie.astype(np.float64).T + m.fib.ravel()
    best = np.argsort(-s, axis=1, kind='stable')[:, :5]
    assert (ids.cpu().numpy() == best).mean() > 0.98                      # fp32 near-ties aside
