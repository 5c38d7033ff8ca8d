"""The binding a reference maintainer would write: raw ctypes on libtkr_hip.so, no helper
module of this repo in between.  Trains a few batches and ranks, checks against the oracle."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import plan_np as P
from oracle import ref_np as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class tkr_bpr_state(C.Structure):                # include/tkr.h
    _fields_ = [(n, C.c_void_p) for n in ('U', 'msU', 'V', 'msV', 'b', 'msb')] + \
               [(n, C.c_int32) for n in ('n_users', 'n_items', 'k', 'mode')] + \
               [(n, C.c_float) for n in ('lu', 'li', 'lj', 'lb', 'lr', 'rho', 'eps')] + [('opt', C.c_int32)]


def test_integration_md_sequence():
    lib = C.CDLL(os.path.join(ROOT, 'top-k-rec_amd', 'libtkr_hip.so'))
    ptr = lambda t: C.c_void_p(t.data_ptr())
    dev = 'cuda'
    rng = np.random.Generator(np.random.PCG64(0))
    n_users, n_items, k, B, nb, seed = 500, 200, 64, 128, 6, 99
    tr = {u: [int(x) for x in rng.integers(0, n_items, int(rng.integers(1, 9)))] for u in range(n_users)}
    tr_users_l = list(tr.keys())
    row_ptr_n, pos_n, srt_n = P.build_csr(tr, n_users)
    i32 = dict(dtype=torch.int32, device=dev)
    tr_users, row_ptr = torch.tensor(tr_users_l, **i32), torch.from_numpy(row_ptr_n).to(dev)
    pos_cols, cols_sorted = torch.from_numpy(pos_n).to(dev), torch.from_numpy(srt_n).to(dev)
    U = torch.zeros(2, n_users, k, device=dev); U[0].normal_(0, 0.01); msU = torch.ones_like(U)
    V = torch.zeros(2, n_items, k, device=dev); V[0].normal_(0, 0.01); msV = torch.ones_like(V)
    b = torch.zeros(2, n_items, device=dev); msb = torch.ones_like(b)
    ref = dict(U=U[0].cpu().numpy(), V=V[0].cpu().numpy(), b=np.zeros(n_items, np.float32), msU=np.ones((n_users, k), np.float32),
               msV=np.ones((n_items, k), np.float32), msb=np.ones(n_items, np.float32))
    ucnt, icnt = torch.zeros(n_users, **i32), torch.zeros(n_items, **i32)
    touch_u, touch_i = torch.zeros(n_users * 16, **i32), torch.zeros(n_items * 16, **i32)
    out_u, out_i, out_j = (torch.empty(nb * B, **i32) for _ in range(3))
    task, occ = torch.empty(nb * 3 * B * 4, **i32), torch.empty(nb * 3 * B * 2, **i32)
    rec = torch.empty(nb * lib.tkr_plan_max_blocks(B) * lib.tkr_plan_team(B) * 16, **i32)
    hdr, occt = torch.empty(nb * 4, **i32), torch.empty(nb * 3 * B, **i32)
    loss = torch.zeros(nb, device=dev)
    hp = dict(lu=2.5e-3, li=2.5e-3, lj=2.5e-4, lb=0.0, lr=0.05, mode='l2')
    st = tkr_bpr_state(ptr(U), ptr(msU), ptr(V), ptr(msV), ptr(b), ptr(msb), n_users, n_items, k, 0,
                       hp['lu'], hp['li'], hp['lj'], hp['lb'], hp['lr'], 0.9, 1e-10)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.tkr_sample_plan(ptr(tr_users), len(tr_users_l), ptr(row_ptr), ptr(pos_cols), ptr(cols_sorted), n_users, n_items,
                               C.c_uint64(seed), C.c_uint64(0), None, nb, B, ptr(ucnt), ptr(icnt), ptr(touch_u), ptr(touch_i),
                               ptr(out_u), ptr(out_i), ptr(out_j), ptr(task), ptr(occ), ptr(rec), ptr(hdr), ptr(occt), None, None, None,
                               None, C.c_int64(0), stream) == 0             # tpar, prec, pocc, workspace unused: one launch per batch (K2), B <= 4096
    assert lib.tkr_bpr_run(C.byref(st), ptr(rec), ptr(occ), ptr(hdr), B, nb, ptr(loss), stream) == 0
    torch.cuda.synchronize()
    u, i, j = P.sample_triplets(tr_users_l, row_ptr_n, pos_n, srt_n, n_items, seed, 0, nb * B)
    np.testing.assert_array_equal(out_u.cpu().numpy(), u)
    ref_loss = [R.bpr_step(ref, u[q * B:(q + 1) * B], i[q * B:(q + 1) * B], j[q * B:(q + 1) * B], hp) for q in range(nb)]
    fue = U[(ucnt & 1).long(), torch.arange(n_users, device=dev)].cpu().numpy()          # bpr.py:151
    fie = V[(icnt & 1).long(), torch.arange(n_items, device=dev)].cpu().numpy()
    np.testing.assert_allclose(fue, ref['U'], rtol=2e-4, atol=1e-5)
    np.testing.assert_allclose(fie, ref['V'], rtol=2e-4, atol=1e-5)
    np.testing.assert_allclose(loss.cpu().numpy(), np.array(ref_loss), rtol=1e-4)

    # ---- score seam: all users against all items, their train positives masked
    total, step = 30, 5
    rated = [sorted(set(tr[uu])) for uu in range(n_users)]
    rptr = torch.tensor(np.r_[0, np.cumsum([len(x) for x in rated])], dtype=torch.int64, device=dev)
    rcols = torch.tensor([c for x in rated for c in x], **i32)
    pitch = (n_users + 31) // 32 * 32
    mask = torch.zeros(((n_items + 31) // 32) * pitch, **i32)
    assert lib.tkr_build_rated_mask(ptr(rptr), ptr(rcols), n_users, n_items, ptr(mask), pitch, stream) == 0
    ids = torch.empty(n_users, total, **i32)
    Ud, Vd = torch.from_numpy(fue).to(dev), torch.from_numpy(fie).to(dev)
    assert lib.tkr_score_topk(ptr(Ud), None, n_users, ptr(Vd), None, n_items, k, ptr(mask), pitch, total, ptr(ids), None,
                              None, C.c_int64(0), stream) == 0
    first = torch.zeros(total // step, dtype=torch.int64, device=dev)
    likes = [sorted(rng.choice(n_items, 5, replace=False).tolist()) for _ in range(n_users)]
    lptr = torch.arange(0, (n_users + 1) * 5, 5, dtype=torch.int64, device=dev)
    lcols = torch.tensor([c for x in likes for c in x], **i32)
    assert lib.tkr_count_hits(ptr(ids), n_users, total, ptr(lptr), ptr(lcols), step, total // step, ptr(first), stream) == 0
    torch.cuda.synchronize()
    s = fue.astype(np.float64) @ fie.astype(np.float64).T
    hits = np.zeros(total // step, np.int64)
    got = ids.cpu().numpy()
    for uu in range(0, n_users, 7):
        sc = s[uu].copy()
        sc[rated[uu]] = -np.inf
        assert len(set(got[uu].tolist()) & set(np.argsort(-sc, kind='stable')[:total].tolist())) >= total - 1
    for uu in range(n_users):
        hits += np.array(R.bucket_hits([int(c) for c in got[uu] if c >= 0], set(likes[uu]), step, total // step))
    np.testing.assert_array_equal(torch.cumsum(first, 0).cpu().numpy(), hits)


class tkr_flow_state(C.Structure):                # include/tkr.h
    _fields_ = [(n, C.c_void_p) for n in ('U', 'msU', 'tailU', 'rdU', 'V', 'msV', 'tailV', 'rdV')] + \
               [(n, C.c_int32) for n in ('n_users', 'n_items', 'k', 'mode')] + \
               [(n, C.c_float) for n in ('lu', 'li', 'lj', 'lb', 'lr', 'rho', 'eps')] + [('opt', C.c_int32), ('item_bufs', C.c_int32)]


def test_integration_md_persistent_sequence():
    """INTEGRATION.md §B.1b with raw ctypes: granule tables, the dataflow form of the plan, ONE persistent launch for the chunk, a
    second launch for the rest of the plan, rollback of what was planned but not run -- against the oracle"""
    lib = C.CDLL(os.path.join(ROOT, 'top-k-rec_amd', 'libtkr_hip.so'))
    lib.tkr_flow_row_granules.restype = C.c_int32
    ptr = lambda t: C.c_void_p(t.data_ptr())
    dev = 'cuda'
    rng = np.random.Generator(np.random.PCG64(1))
    n_users, n_items, k, B, nb, run, seed = 500, 200, 50, 128, 8, 6, 4242
    tr = {u: [int(x) for x in rng.integers(0, n_items, int(rng.integers(1, 9)))] for u in range(n_users)}
    tr_users_l = list(tr.keys())
    row_ptr_n, pos_n, srt_n = P.build_csr(tr, n_users)
    i32 = dict(dtype=torch.int32, device=dev)
    tr_users, row_ptr = torch.tensor(tr_users_l, **i32), torch.from_numpy(row_ptr_n).to(dev)
    pos_cols, cols_sorted = torch.from_numpy(pos_n).to(dev), torch.from_numpy(srt_n).to(dev)
    ref = R.init_bpr_state(n_users, n_items, k, rng)
    kp = lib.tkr_flow_row_granules(k)
    assert kp == 128

    def granules(n, w, init=None, pad=0.0):         # [2][n][w] x {fp32 value, uint32 tag}: version 0 in buffer 0, no version in buffer 1
        t = torch.zeros(2, n, w, 2, device=dev)
        t[0, :, :, 0] = pad
        if init is not None:
            t[0, :, :init.shape[1], 0] = torch.from_numpy(init).to(dev)
        t.view(torch.int32)[1, :, :, 1] = -1
        return t
    U, msU = granules(n_users, kp, ref['U']), granules(n_users, kp, ref['msU'], pad=1.0)
    V, msV = granules(n_items, kp, ref['V']), granules(n_items, kp, ref['msV'], pad=1.0)
    tailU, tailV = granules(n_users, 4), granules(n_items, 4)
    tailV[0, :, 1, 0] = 1.0                         # the slot of the (zero) bias
    rdU, rdV = torch.zeros(2 * n_users, **i32), torch.zeros(2 * n_items, **i32)
    lib.tkr_flow_ctl_words.restype = C.c_int32
    ctl = torch.zeros(lib.tkr_flow_ctl_words(), **i32)
    ucnt, icnt = torch.zeros(n_users, **i32), torch.zeros(n_items, **i32)
    touch_u, touch_i = torch.zeros(n_users * 16, **i32), torch.zeros(n_items * 16, **i32)
    out_u, out_i, out_j = (torch.empty(nb * B, **i32) for _ in range(3))
    task, occ, occt = torch.empty(nb * 3 * B * 4, **i32), torch.empty(nb * 3 * B * 2, **i32), torch.empty(nb * 3 * B, **i32)
    prec, pocc = torch.empty(nb * 3 * B * 32, **i32), torch.empty(nb * 3 * B * 4, **i32)
    loss = torch.zeros(nb, device=dev)
    hp = dict(lu=2.5e-3, li=2.5e-3, lj=2.5e-4, lb=1e-3, lr=0.05, mode='l2')
    st = tkr_flow_state(ptr(U), ptr(msU), ptr(tailU), ptr(rdU), ptr(V), ptr(msV), ptr(tailV), ptr(rdV), n_users, n_items, k, 0,
                        hp['lu'], hp['li'], hp['lj'], hp['lb'], hp['lr'], 0.9, 1e-10, 0, 2)      # opt = RMSProp, two buffers per item row
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.tkr_sample_plan(ptr(tr_users), len(tr_users_l), ptr(row_ptr), ptr(pos_cols), ptr(cols_sorted), n_users, n_items,
                               C.c_uint64(seed), C.c_uint64(0), None, nb, B, ptr(ucnt), ptr(icnt), ptr(touch_u), ptr(touch_i),
                               ptr(out_u), ptr(out_i), ptr(out_j), ptr(task), ptr(occ), None, None, ptr(occt), None, ptr(prec), ptr(pocc),
                               None, C.c_int64(0), stream) == 0
    at = lambda first: C.c_void_p(prec.data_ptr() + first * 3 * B * 32 * 4)              # the record of the first task of batch `first`
    assert lib.tkr_bpr_flow_run(C.byref(st), at(0), ptr(pocc), B, 2, ptr(ctl), ptr(loss), 0, stream) == 0          # batches 0, 1
    assert lib.tkr_bpr_flow_run(C.byref(st), at(2), ptr(pocc), B, run - 2, ptr(ctl), ptr(loss), 0, stream) == 0    # batches 2 .. 5
    assert lib.tkr_plan_rollback(ptr(task), B, run, nb - run, ptr(ucnt), ptr(icnt), stream) == 0                            # 6, 7 never run
    torch.cuda.synchronize()
    assert int(ctl[1026]) == 0                      # TKR_FLOW_CTL_STATUS
    u, i, j = P.sample_triplets(tr_users_l, row_ptr_n, pos_n, srt_n, n_items, seed, 0, nb * B)
    np.testing.assert_array_equal(out_u.cpu().numpy(), u)
    ref_loss = [R.bpr_step(ref, u[q * B:(q + 1) * B], i[q * B:(q + 1) * B], j[q * B:(q + 1) * B], hp) for q in range(run)]
    cu, ci = np.zeros(n_users, np.int32), np.zeros(n_items, np.int32)
    for q in range(run):
        cu[np.unique(u[q * B:(q + 1) * B])] += 1
        ci[np.unique(np.concatenate([i[q * B:(q + 1) * B], j[q * B:(q + 1) * B]]))] += 1
    np.testing.assert_array_equal(ucnt.cpu().numpy(), cu)                                  # counters = batches that RAN
    np.testing.assert_array_equal(icnt.cpu().numpy(), ci)
    fue = U[(ucnt & 1).long(), torch.arange(n_users, device=dev), :k, 0].cpu().numpy()     # bpr.py:151: the current buffer's values
    fie = V[(icnt & 1).long(), torch.arange(n_items, device=dev), :k, 0].cpu().numpy()
    fib = tailV[(icnt & 1).long(), torch.arange(n_items, device=dev), 0, 0].cpu().numpy()
    np.testing.assert_allclose(fue, ref['U'], rtol=2e-4, atol=1e-5)
    np.testing.assert_allclose(fie, ref['V'], rtol=2e-4, atol=1e-5)
    np.testing.assert_allclose(fib, ref['b'], rtol=2e-4, atol=1e-5)
    np.testing.assert_allclose(loss.cpu().numpy()[:run], np.array(ref_loss), rtol=1e-4)


class tkr_plan_call(C.Structure):                   # include/tkr.h, field for field
    _fields_ = [(n, C.c_void_p) for n in ('tr_users', 'row_ptr', 'pos_cols', 'cols_sorted', 'ucnt', 'icnt', 'touch_u', 'touch_i', 'out_u', 'out_i',
                                           'out_j', 'task', 'occ', 'occt', 'prec', 'pocc', 'ohdr')] + \
               [('seed', C.c_uint64), ('first_triplet', C.c_uint64)] + \
               [(n, C.c_int32) for n in ('n_tr', 'n_users', 'n_items', 'n_batches', 'batch_size', 'n_owner', 'ohdr_stride', 'reserved')]


def test_integration_md_owned_rows_in_one_call():
    """INTEGRATION.md §B.1b (its K2o part) with raw ctypes: K1 of a short chunk and the persistent step with owned item rows (K2o) in ONE C call
    (tkr_bpr_own_plan_run over a tkr_plan_call), twice in a row on the same tables, two HIP events of the caller recorded around the
    step launch -- against the oracle"""
    lib = C.CDLL(os.path.join(ROOT, 'top-k-rec_amd', 'libtkr_hip.so'))
    lib.tkr_flow_row_granules.restype = C.c_int32
    lib.tkr_flow_ctl_words.restype = C.c_int32
    lib.tkr_bpr_own_owners.restype = C.c_int32
    ptr = lambda t: C.c_void_p(t.data_ptr())
    dev = 'cuda'
    rng = np.random.Generator(np.random.PCG64(2))
    n_users, n_items, k, B, nb, seed = 500, 200, 50, 128, 5, 777
    tr = {u: [int(x) for x in rng.integers(0, n_items, int(rng.integers(1, 9)))] for u in range(n_users)}
    tr_users_l = list(tr.keys())
    row_ptr_n, pos_n, srt_n = P.build_csr(tr, n_users)
    i32 = dict(dtype=torch.int32, device=dev)
    tr_users, row_ptr = torch.tensor(tr_users_l, **i32), torch.from_numpy(row_ptr_n).to(dev)
    pos_cols, cols_sorted = torch.from_numpy(pos_n).to(dev), torch.from_numpy(srt_n).to(dev)
    ref = R.init_bpr_state(n_users, n_items, k, rng)
    kp = lib.tkr_flow_row_granules(k)
    n_owner = lib.tkr_bpr_own_owners(n_items, k)
    assert n_owner > 0

    def granules(n, w, init=None, pad=0.0):
        t = torch.zeros(2, n, w, 2, device=dev)
        t[0, :, :, 0] = pad
        if init is not None:
            t[0, :, :init.shape[1], 0] = torch.from_numpy(init).to(dev)
        t.view(torch.int32)[1, :, :, 1] = -1
        return t
    U, msU = granules(n_users, kp, ref['U']), granules(n_users, kp, ref['msU'], pad=1.0)
    V, msV = granules(n_items, kp, ref['V']), granules(n_items, kp, ref['msV'], pad=1.0)
    tailU, tailV = granules(n_users, 4), granules(n_items, 4)
    tailV[0, :, 1, 0] = 1.0
    rdU, rdV = torch.zeros(2 * n_users, **i32), torch.zeros(2 * n_items, **i32)
    ctl = torch.zeros(lib.tkr_flow_ctl_words(), **i32)
    ucnt, icnt = torch.zeros(n_users, **i32), torch.zeros(n_items, **i32)
    touch_u, touch_i = torch.zeros(n_users * 16, **i32), torch.zeros(n_items * 16, **i32)
    out_u, out_i, out_j = (torch.empty(nb * B, **i32) for _ in range(3))
    task, occ, occt = torch.empty(nb * 3 * B * 4, **i32), torch.empty(nb * 3 * B * 2, **i32), torch.empty(nb * 3 * B, **i32)
    prec, pocc = torch.empty(nb * 3 * B * 32, **i32), torch.empty(nb * 3 * B * 4, **i32)
    ohdr, xch = torch.zeros(n_owner * nb, **i32), torch.zeros(nb * B * 4, **i32)
    loss = torch.zeros(nb, device=dev)
    hp = dict(lu=2.5e-3, li=2.5e-3, lj=2.5e-4, lb=1e-3, lr=0.05, mode='l2')
    st = tkr_flow_state(ptr(U), ptr(msU), ptr(tailU), ptr(rdU), ptr(V), ptr(msV), ptr(tailV), ptr(rdV), n_users, n_items, k, 0,
                        hp['lu'], hp['li'], hp['lj'], hp['lb'], hp['lr'], 0.9, 1e-10, 0, 2)
    pc = tkr_plan_call()
    for name, t in (('tr_users', tr_users), ('row_ptr', row_ptr), ('pos_cols', pos_cols), ('cols_sorted', cols_sorted), ('ucnt', ucnt), ('icnt', icnt),
                    ('touch_u', touch_u), ('touch_i', touch_i), ('out_u', out_u), ('out_i', out_i), ('out_j', out_j), ('task', task), ('occ', occ),
                    ('occt', occt), ('prec', prec), ('pocc', pocc), ('ohdr', ohdr)):
        setattr(pc, name, t.data_ptr())
    pc.seed, pc.n_tr, pc.n_users, pc.n_items, pc.batch_size, pc.n_owner, pc.ohdr_stride = seed, len(tr_users_l), n_users, n_items, B, n_owner, nb
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for e in ev:
        e.record()                                  # a torch event exists on the device from its first record() on
    ref_loss, got_loss, u_all = [], [], []
    for call, n in enumerate((nb, 3)):              # two short calls: 5 batches, then 3 more from where the stream stands
        pc.first_triplet, pc.n_batches = sum((nb, 3)[:call]) * B, n
        assert lib.tkr_bpr_own_plan_run(C.byref(pc), C.byref(st), 0, n, ptr(ctl), ptr(loss), 0, ptr(xch), C.c_uint32(call + 1),
                                        C.c_void_p(ev[0].cuda_event), C.c_void_p(ev[1].cuda_event), stream) == 0
        torch.cuda.synchronize()
        assert int(ctl[1026]) == 0 and ev[0].elapsed_time(ev[1]) > 0.0
        u, i, j = P.sample_triplets(tr_users_l, row_ptr_n, pos_n, srt_n, n_items, seed, int(pc.first_triplet), n * B)
        np.testing.assert_array_equal(out_u.cpu().numpy()[:n * B], u)
        ref_loss += [R.bpr_step(ref, u[q * B:(q + 1) * B], i[q * B:(q + 1) * B], j[q * B:(q + 1) * B], hp) for q in range(n)]
        got_loss += loss.cpu().numpy()[:n].tolist()
        loss.zero_()
    fue = U[(ucnt & 1).long(), torch.arange(n_users, device=dev), :k, 0].cpu().numpy()
    fie = V[(icnt & 1).long(), torch.arange(n_items, device=dev), :k, 0].cpu().numpy()
    fib = tailV[(icnt & 1).long(), torch.arange(n_items, device=dev), 0, 0].cpu().numpy()
    np.testing.assert_allclose(fue, ref['U'], rtol=2e-4, atol=1e-5)
    np.testing.assert_allclose(fie, ref['V'], rtol=2e-4, atol=1e-5)
    np.testing.assert_allclose(fib, ref['b'], rtol=2e-4, atol=1e-5)
    np.testing.assert_allclose(np.array(got_loss), np.array(ref_loss), rtol=1e-4)
