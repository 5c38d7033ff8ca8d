"""GPU parity of K1 (sampler + planner) and K2 (BPR step) against the oracle, through the C ABI.

K1 is integer work: bit-exact against oracle/plan_np.py.  K2 is fp32: the tables after N
sequential mini-batches agree with oracle/ref_np.bpr_step on the same init and the same
(u,i,j) stream within the tolerance stated in each test."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

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


def _run_plan(hip, tr, tr_users, n_users, n_items, seed, first, nb, B):
    row_ptr, pos, srt = P.build_csr(tr, n_users)
    out = [torch.empty(nb * B, dtype=torch.int32, device='cuda') for _ in range(3)]
    task = torch.empty(nb * 3 * B * 4, dtype=torch.int32, device='cuda')
    occ = torch.empty(nb * 3 * B * 2, dtype=torch.int32, device='cuda')
    hip.sample_plan(_dev(np.asarray(tr_users, np.int32)), _dev(row_ptr), _dev(pos), _dev(srt), n_items, seed, first,
                    nb, B, out[0], out[1], out[2], task, occ)
    torch.cuda.synchronize()
    exp = P.sample_and_plan(tr_users, row_ptr, pos, srt, n_items, seed, first, nb, B)
    got = [o.cpu().numpy() for o in out] + [task.cpu().numpy().reshape(nb, 3 * B, 4), occ.cpu().numpy().reshape(nb, 3 * B, 2)]
    return got, exp


@pytest.mark.parametrize('n_users,n_items,B,nb', [(60, 40, 32, 5), (300, 150, 256, 7), (300, 150, 100, 3),
                                                  (5000, 900, 1024, 3), (5000, 900, 8192, 2), (40, 30, 1, 4)])
def test_sample_plan_bit_exact(hip, n_users, n_items, B, nb):
    tr, tr_users = _toy(n_users, n_items, seed=n_users + B)
    got, exp = _run_plan(hip, tr, tr_users, n_users, n_items, seed=0x1234567890ABCDEF, first=(1 << 33) + 17, nb=nb, B=B)
    for name, g, e in zip(('u', 'i', 'j', 'task', 'occ'), got, exp):
        np.testing.assert_array_equal(g, e, err_msg=name)


def test_sample_plan_ctl_offset(hip):
    """the device-side batch base (ctl) walks the same stream as first_triplet does"""
    tr, tr_users = _toy(200, 90, seed=1)
    row_ptr, pos, srt = P.build_csr(tr, 200)
    B, nb = 64, 4
    ctl = torch.tensor([5], dtype=torch.int64, device='cuda')
    out = [torch.empty(nb * B, dtype=torch.int32, device='cuda') for _ in range(3)]
    task = torch.empty(nb * 3 * B * 4, dtype=torch.int32, device='cuda')
    occ = torch.empty(nb * 3 * B * 2, dtype=torch.int32, device='cuda')
    hip.sample_plan(_dev(np.asarray(tr_users, np.int32)), _dev(row_ptr), _dev(pos), _dev(srt), 90, 77, 1000, nb, B,
                    out[0], out[1], out[2], task, occ, ctl=ctl)
    u, i, j = P.sample_triplets(tr_users, row_ptr, pos, srt, 90, 77, 1000 + 5 * B, nb * B)
    np.testing.assert_array_equal(out[0].cpu().numpy(), u)
    np.testing.assert_array_equal(out[2].cpu().numpy(), j)


def _state_struct(hip, T, n_users, n_items, k, hp):
    st = hip.BprState()
    st.U, st.msU, st.ustamp = T['U'].data_ptr(), T['msU'].data_ptr(), T['ustamp'].data_ptr()
    st.V, st.msV, st.b, st.msb = T['V'].data_ptr(), T['msV'].data_ptr(), T['b'].data_ptr(), T['msb'].data_ptr()
    st.istamp = T['istamp'].data_ptr()
    st.n_users, st.n_items, st.k = n_users, n_items, k
    st.mode = 0 if hp['mode'] == 'l2' else 1
    st.lu, st.li, st.lj, st.lb, st.lr = hp['lu'], hp['li'], hp['lj'], hp['lb'], hp['lr']
    st.rho, st.eps = 0.9, 1e-10
    return st


def _tables(ref, n_users, n_items, k):
    T = {}
    for name, n, kk in (('U', n_users, k), ('V', n_items, k), ('b', n_items, 0)):
        shape = (2, n, kk) if kk else (2, n)
        T[name] = torch.zeros(shape, device='cuda')
        T[name][0] = torch.from_numpy(ref[name]).cuda()
        T['ms' + name] = torch.ones(shape, device='cuda')
    T['ustamp'] = torch.zeros(n_users, dtype=torch.int32, device='cuda')
    T['istamp'] = torch.zeros(n_items, dtype=torch.int32, device='cuda')
    return T


def _current(T, name, stamp):
    sel = (T[stamp] & 1).long()
    idx = torch.arange(T[name].shape[1], device='cuda')
    return T[name][sel, idx].cpu().numpy()


@pytest.mark.parametrize('k,B,nb,mode,lr', [(16, 64, 12, 'l2', 0.05), (128, 256, 10, 'l2', 0.05),
                                           (50, 256, 6, 'l1', 0.05), (200, 128, 4, 'l2', 1e-4),
                                           (128, 2048, 3, 'l2', 0.05)])
def test_bpr_step_parity(hip, k, B, nb, mode, lr):
    n_users, n_items = 400, 120               # small tables: many in-batch duplicate rows
    tr, tr_users = _toy(n_users, n_items, seed=k + B, all_but_one=False)
    rng = np.random.Generator(np.random.PCG64(k))
    ref = R.init_bpr_state(n_users, n_items, k, rng)
    ref['b'][:] = (rng.standard_normal(n_items) * 0.01).astype(np.float32)
    hp = dict(lu=2.5e-3, li=2.5e-3, lj=2.5e-4, lb=1e-3, lr=lr, mode=mode)
    T = _tables(ref, n_users, n_items, k)
    got, exp = _run_plan(hip, tr, tr_users, n_users, n_items, seed=42, first=0, nb=nb, B=B)
    task, occ = _dev(exp[3].reshape(-1)), _dev(exp[4].reshape(-1))
    loss = torch.zeros(nb, device='cuda')
    hip.bpr_run(_state_struct(hip, T, n_users, n_items, k, hp), task, occ, B, nb, 1, loss)
    torch.cuda.synchronize()
    ref_loss = []
    u, i, j = exp[0], exp[1], exp[2]
    for b in range(nb):
        sl = slice(b * B, (b + 1) * B)
        ref_loss.append(R.bpr_step(ref, u[sl], i[sl], j[sl], hp))
    # fp32 tolerance: |dP| per step <= lr/sqrt(0.1) ~ 0.16 at lr=0.05, compared at 1e-5 abs + 2e-4 rel
    for name, stamp in (('U', 'ustamp'), ('V', 'istamp'), ('b', 'istamp')):
        np.testing.assert_allclose(_current(T, name, stamp), ref[name], rtol=2e-4, atol=1e-5, err_msg=name)
        np.testing.assert_allclose(_current(T, 'ms' + name, stamp), ref['ms' + name], rtol=2e-4, atol=1e-7, err_msg='ms' + name)
    np.testing.assert_allclose(loss.cpu().numpy(), np.array(ref_loss), rtol=1e-4)
    # untouched rows are bit-identical to the init and their stamps are still zero
    touched_u = np.zeros(n_users, bool); touched_u[u] = True
    assert np.all(T['ustamp'].cpu().numpy()[~touched_u] == 0)


def test_bpr_step_is_deterministic(hip):
    """same plan, same init -> bitwise identical tables (no float atomics on the parameters)"""
    n_users, n_items, k, B, nb = 300, 80, 128, 256, 8
    tr, tr_users = _toy(n_users, n_items, seed=9, all_but_one=False)
    hp = dict(lu=2.5e-3, li=2.5e-3, lj=2.5e-4, lb=0.0, lr=0.01, mode='l2')
    _, exp = _run_plan(hip, tr, tr_users, n_users, n_items, seed=3, first=0, nb=nb, B=B)
    outs = []
    for _ in range(2):
        ref = R.init_bpr_state(n_users, n_items, k, np.random.Generator(np.random.PCG64(0)))
        T = _tables(ref, n_users, n_items, k)
        hip.bpr_run(_state_struct(hip, T, n_users, n_items, k, hp), _dev(exp[3].reshape(-1)), _dev(exp[4].reshape(-1)), B, nb, 1, None)
        torch.cuda.synchronize()
        outs.append([_current(T, n, s) for n, s in (('U', 'ustamp'), ('V', 'istamp'), ('b', 'istamp'))])
    for a, b in zip(*outs):
        np.testing.assert_array_equal(a, b)


def test_abi_rejects_bad_arguments(hip):
    st = hip.BprState()
    rc = hip.lib().tkr_bpr_step(C.byref(st), None, None, 256, 1, None, None)
    assert rc == -1
    assert hip.lib().tkr_sample_plan(None, 0, None, None, None, 10, 0, 0, None, 1, 256, None, None, None, None, None, None) == -1
    assert hip.lib().tkr_sample_plan(None, 1, None, None, None, 10, 0, 0, None, 1, 16384, None, None, None, None, None, None) == -2
