"""-m gpu: the persistent dataflow forms of the BPR step (K2f, csrc/bpr_flow.hip; K2o with owned item rows, csrc/bpr_own.hip)
and the dataflow form of K1's plan.

K1's extra outputs are integer work: bit-exact against oracle/plan_np.flow_records.  K2f is held to the SAME bars as K2
(tests/test_gpu_bpr.py): tables after N sequential mini-batches within 1e-5 + 2e-4*|x| of oracle/ref_np.bpr_step on the
same init and (u,i,j) stream, bitwise run-to-run determinism, plus what only a persistent kernel can get wrong: a chunk
cut into several launches equals one launch, rows that are hammered in every batch (long hand-off chains), rows with many
occurrences per batch (one wave walks them), and the version / acknowledge bookkeeping left in the tables."""
import ctypes as C

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


def _toy(n_users, n_items, seed, max_deg=12):
    rng = np.random.Generator(np.random.PCG64(seed))
    tr = {}
    for u in rng.permutation(n_users)[: max(1, n_users - n_users // 8)]:
        tr[int(u)] = [int(x) for x in rng.integers(0, n_items, int(rng.integers(1, max_deg)))]
    return tr, list(tr.keys())


KERNEL_TUNE = {'f': 0, 'o': 0, 's': 0x8000, 'w': 0xc000, 'l': 0x0100}      # K2f | K2o, item tasks read rows (default) | ... exchange scalars | ... on 16 waves | K2o with a loader wave (partner rows staged in LDS)


def _owners(hip, which, n_items, k):
    """'f': K2f (no owners); 'o' / 's' / 'w': K2o (see KERNEL_TUNE) with the device's owner count; 'o8', 's3', ...: on 8 / 3 workgroups
    (several rows per owner)"""
    if which == 'f':
        return 0
    if which[0] in 'swl' and not hip.lab():
        pytest.skip('the scalar-exchange / 16-wave / loader forms of K2o are lab forms (make -C top-k-rec_amd/csrc LAB=1)')
    n = hip.bpr_own_owners(n_items, k)
    assert n > 0
    return n if len(which) == 1 else int(which[1:])


def _plan(hip, tr, tr_users, n_users, n_items, seed, first, nb, B, chunks=1, owners=0):
    from single import _engine
    dev = torch.device('cuda')
    row_ptr, pos, srt = P.build_csr(tr, n_users)
    csr = _engine.TrainingCSR.from_arrays(row_ptr, pos, np.asarray(tr_users, np.int32), dev)
    cnt = _engine.UpdateCounters(n_users, n_items, dev)
    plan = _engine.PlanBuffers(nb, B, dev, flow=True, owners=owners)
    ucnt, icnt = np.zeros(n_users, np.int32), np.zeros(n_items, np.int32)
    for c in range(chunks):
        hip.sample_plan(csr, n_users, n_items, seed, first + c * nb * B, nb, B, cnt, plan)
        exp = P.sample_and_plan(tr_users, row_ptr, pos, srt, n_items, seed, first + c * nb * B, nb, B, ucnt, icnt, n_owner=owners)
    torch.cuda.synchronize()
    return plan, exp, cnt, (ucnt, icnt)


@pytest.mark.parametrize('owners', [0, 1, 7, 256])
@pytest.mark.parametrize('n_users,n_items,B,nb,chunks', [(60, 40, 32, 5, 1), (300, 150, 256, 7, 3), (300, 150, 100, 3, 2),
                                                         (5000, 900, 1024, 3, 1), (40, 30, 1, 4, 1), (300, 150, 64, 512, 2),
                                                         # K1's forms: sorts in registers (128 < B, keys per thread 1+2 / 2+4 / 4+8) or in LDS; the
                                                         # records of a short call (<= 64 batches of <= 256: one thread per slot) or of a long one
                                                         (300, 150, 256, 80, 1), (2000, 600, 512, 4, 2), (2000, 600, 200, 5, 1), (700, 300, 129, 66, 1)])
def test_flow_plan_bit_exact(hip, n_users, n_items, B, nb, chunks, owners):
    tr, tr_users = _toy(n_users, n_items, seed=n_users + B)
    plan, exp, cnt, (ucnt, icnt) = _plan(hip, tr, tr_users, n_users, n_items, 0x1234567890ABCDEF, (1 << 33) + 17, nb, B, chunks, owners)
    flow = P.sample_and_plan.last_flow
    np.testing.assert_array_equal(plan.u.cpu().numpy(), exp[0])
    np.testing.assert_array_equal(plan.j.cpu().numpy(), exp[2])
    np.testing.assert_array_equal(plan.pocc.cpu().numpy().reshape(nb, 3 * B, 4), flow['pocc'], err_msg='pocc')
    np.testing.assert_array_equal(plan.prec.cpu().numpy().reshape(nb, 3 * B, 32), flow['prec'], err_msg='prec')
    if owners:
        np.testing.assert_array_equal(plan.ohdr.cpu().numpy().reshape(owners, plan.cap)[:, :nb], flow['ohdr'], err_msg='ohdr')
        live = flow['prec'][:, :, 0] != -1
        assert ((flow['prec'][:, :, 5] >= -1) & (flow['prec'][:, :, 5] < np.arange(nb)[:, None]))[live].all()
    np.testing.assert_array_equal(plan.task.cpu().numpy().reshape(nb, 3 * B, 4)[:, :, :3], flow['task'][:, :, :3], err_msg='task')
    np.testing.assert_array_equal(cnt.ucnt.cpu().numpy(), ucnt)
    np.testing.assert_array_equal(cnt.icnt.cpu().numpy(), icnt)
    assert int(cnt.touch_u.abs().sum()) == 0 and int(cnt.touch_i.abs().sum()) == 0


@pytest.mark.parametrize('owners', ['dev', 7, 64])
@pytest.mark.parametrize('n_users,n_items,B,nb,k', [(300, 150, 256, 7, 64), (2000, 600, 256, 20, 128), (300, 150, 100, 3, 32), (40, 30, 1, 4, 16),
                                                    (700, 300, 129, 64, 128), (5000, 4000, 200, 33, 100), (60, 40, 32, 5, 16)])
def test_planner_prologue_of_the_step_is_bit_exact(hip, n_users, n_items, B, nb, k, owners):
    """K1 INSIDE the step's launch (csrc/bpr_own.hip PLAN = true, reached through tkr_bpr_own_plan_run for a short call): the same plan
    words as the planner kernels and the oracle -- triplets, tasks, occurrences, records, owner runs, update counters, a clean touch
    bitmap -- for two calls in a row (the second one's versions start from the first one's counters), and the tables the step
    leaves are the oracle's"""
    from single import _engine
    owners = hip.bpr_own_owners(n_items, k) if owners == 'dev' else owners
    if nb > owners:
        pytest.skip('one planner workgroup per batch: a call of more batches than owners plans with the kernels of csrc/sampler.hip')
    tr, tr_users = _toy(n_users, n_items, seed=n_users + B)
    dev = torch.device('cuda')
    row_ptr, pos, srt = P.build_csr(tr, n_users)
    csr = _engine.TrainingCSR.from_arrays(row_ptr, pos, np.asarray(tr_users, np.int32), dev)
    cnt = _engine.UpdateCounters(n_users, n_items, dev)
    plan = _engine.PlanBuffers(nb, B, dev, flow=True, owners=owners)
    rng = np.random.Generator(np.random.PCG64(k))
    ref = R.init_bpr_state(n_users, n_items, k, rng)
    hp = dict(lu=2.5e-3, li=2.5e-3, lj=2.5e-4, lb=1e-3, lr=0.02, mode='l2')
    F = _Flow(hip, ref, n_users, n_items, k, hp, bufs=4)
    step = hip.own_stepper(F.st, B, F.ctl)
    call = hip.plan_call(csr, n_users, n_items, 0xABCDEF12345, B, cnt, plan)
    ucnt, icnt = np.zeros(n_users, np.int32), np.zeros(n_items, np.int32)
    tot_u, tot_i = np.zeros(n_users, np.int32), np.zeros(n_items, np.int32)
    uocc, iocc = np.zeros(n_users, np.int64), np.zeros(n_items, np.int64)
    first = (1 << 33) + 5
    for c in range(2):
        call.first_triplet, call.n_batches = first + c * nb * B, nb
        loss = torch.zeros(nb, device=dev)
        step.plan_and_run(plan, call, 0, nb, loss)
        exp = P.sample_and_plan(tr_users, row_ptr, pos, srt, n_items, 0xABCDEF12345, first + c * nb * B, nb, B, ucnt, icnt, n_owner=owners)
        flow = P.sample_and_plan.last_flow
        torch.cuda.synchronize()
        assert int(F.ctl[hip.FLOW_CTL_STATUS]) == 0
        for name, got, want in (('u', plan.u, exp[0]), ('i', plan.i, exp[1]), ('j', plan.j, exp[2])):
            np.testing.assert_array_equal(got.cpu().numpy()[:nb * B], want, err_msg=name)
        np.testing.assert_array_equal(plan.pocc.cpu().numpy().reshape(nb, 3 * B, 4), flow['pocc'], err_msg='pocc')
        np.testing.assert_array_equal(plan.prec.cpu().numpy().reshape(nb, 3 * B, 32), flow['prec'], err_msg='prec')
        np.testing.assert_array_equal(plan.ohdr.cpu().numpy().reshape(owners, plan.cap)[:, :nb], flow['ohdr'], err_msg='ohdr')
        np.testing.assert_array_equal(plan.task.cpu().numpy().reshape(nb, 3 * B, 4)[:, :, :3], flow['task'][:, :, :3], err_msg='task')
        np.testing.assert_array_equal(cnt.ucnt.cpu().numpy(), ucnt)
        np.testing.assert_array_equal(cnt.icnt.cpu().numpy(), icnt)
        assert int(cnt.touch_u.abs().sum()) == 0 and int(cnt.touch_i.abs().sum()) == 0
        a, b, uo, io, ref_loss = _oracle(ref, exp, n_users, n_items, nb, B, hp)
        tot_u += a; tot_i += b; uocc += uo; iocc += io
        np.testing.assert_allclose(loss.cpu().numpy(), ref_loss, rtol=1e-4)
    F.item_readers = 2
    _check(F, ref, tot_u, tot_i, uocc, iocc)


class _Flow:
    """granule tables of one model, initialised from an oracle state dict"""

    def __init__(self, hip, ref, n_users, n_items, k, hp, bufs=None):
        """bufs: buffers per item row (2 or 4; default: alternating between the tests' parameter sets by the parity of k // 16 + n_items)"""
        from single import _engine
        dev = torch.device('cuda')
        bufs = bufs or (4 if (k // 16 + n_items) % 2 == 0 else 2)
        self.hip, self.n_users, self.n_items, self.k, self.bufs = hip, n_users, n_items, k, bufs
        self.U, self.V = _engine.FlowTable(n_users, k, dev), _engine.FlowTable(n_items, k, dev, bufs)
        self.tU, self.tV = _engine.FlowTail(n_users, dev), _engine.FlowTail(n_items, dev, bufs)
        self.U.assign(torch.from_numpy(ref['U']).cuda(), torch.from_numpy(ref['msU']).cuda())
        self.V.assign(torch.from_numpy(ref['V']).cuda(), torch.from_numpy(ref['msV']).cuda())
        self.tU.assign()
        self.tV.assign(torch.from_numpy(ref['b']).cuda(), torch.from_numpy(ref['msb']).cuda())
        self.ctl = torch.zeros(hip.flow_ctl_words(), dtype=torch.int32, device=dev)
        st = hip.FlowState()
        st.U, st.msU, st.tailU, st.rdU = self.U.p.data_ptr(), self.U.ms.data_ptr(), self.tU.t.data_ptr(), self.tU.rd.data_ptr()
        st.V, st.msV, st.tailV, st.rdV = self.V.p.data_ptr(), self.V.ms.data_ptr(), self.tV.t.data_ptr(), self.tV.rd.data_ptr()
        st.n_users, st.n_items, st.k = n_users, n_items, k
        st.mode = 0 if hp['mode'] == 'l2' else 1
        st.lu, st.li, st.lj, st.lb, st.lr = hp['lu'], hp['li'], hp['lj'], hp['lb'], hp['lr']
        st.rho, st.eps = 0.9, 1e-10
        st.opt = 1 if hp.get('opt') == 'sgd' else 0
        st.item_bufs = bufs
        self.st = st

    def run(self, plan, B, nb, loss=None, first=0, waves_per_cu=0, kernel='o'):
        self.item_readers = 1 if plan.owners and kernel[0] in 'sw' else 2      # scalar exchange: an item row is read by the user tasks only
        if plan.owners:                  # K2o; waves_per_cu = owner waves per workgroup here
            self.hip.bpr_own_run(self.st, plan, B, nb, self.ctl, loss, first=first, owner_waves=waves_per_cu | KERNEL_TUNE[kernel[0]])
        else:
            self.hip.bpr_flow_run(self.st, plan, B, nb, self.ctl, loss, first=first, waves_per_cu=waves_per_cu)

    def status(self):
        torch.cuda.synchronize()
        c = self.ctl.cpu().numpy()
        return int(c[self.hip.FLOW_CTL_STATUS]), c

    def current(self, ucnt, icnt):
        cu, ci = torch.from_numpy(ucnt).cuda(), torch.from_numpy(icnt).cuda()
        out = {}
        out['U'], out['msU'] = (t.cpu().numpy() for t in self.U.current(cu))
        out['V'], out['msV'] = (t.cpu().numpy() for t in self.V.current(ci))
        out['b'], out['msb'] = (t.cpu().numpy() for t in self.tV.current(ci))
        return out

    def raw(self):
        return [t.view(torch.int32).clone()           # bits: a tag of 0xffffffff is a NaN as fp32
                for t in (self.U.p, self.U.ms, self.V.p, self.V.ms, self.tU.t, self.tV.t, self.tU.rd, self.tV.rd)]


def _oracle(ref, exp, n_users, n_items, nb, B, hp):
    u, i, j = exp[0], exp[1], exp[2]
    ucnt, icnt = np.zeros(n_users, np.int32), np.zeros(n_items, np.int32)
    uocc, iocc = np.zeros(n_users, np.int64), np.zeros(n_items, np.int64)
    losses = []
    for b in range(nb):
        sl = slice(b * B, (b + 1) * B)
        losses.append(R.bpr_step(ref, u[sl], i[sl], j[sl], hp))
        ucnt[np.unique(u[sl])] += 1
        icnt[np.unique(np.concatenate([i[sl], j[sl]]))] += 1
        np.add.at(uocc, u[sl], 1)
        np.add.at(iocc, np.concatenate([i[sl], j[sl]]), 1)
    return ucnt, icnt, uocc, iocc, np.array(losses)


def _check(F, ref, ucnt, icnt, uocc, iocc, tol=dict(rtol=2e-4, atol=1e-5), slots=True):
    status, ctl = F.status()
    assert status == 0, 'a bounded spin ran out'
    A = F.hip.FLOW_CTL_ARRIVE
    assert not ctl[:A + 2].any()        # the ticket, arrival and leave words are back at zero: the next launch needs no memset
    got = F.current(ucnt, icnt)
    for name in ('U', 'V', 'b'):
        np.testing.assert_allclose(got[name], ref[name], err_msg=name, **tol)
        if slots:
            np.testing.assert_allclose(got['ms' + name], ref['ms' + name], rtol=2e-4, atol=1e-7, err_msg='ms' + name)
    # bookkeeping the kernel leaves behind: every partner read acknowledged (2 per occurrence), current tags = update counts
    np.testing.assert_array_equal(F.tU.rd.cpu().numpy().reshape(-1, 2).sum(1), 2 * uocc)
    np.testing.assert_array_equal(F.tV.rd.cpu().numpy().reshape(-1, F.bufs).sum(1), F.item_readers * iocc)
    from single._engine import _tags
    for tab, cnt in ((F.U.p, ucnt), (F.V.p, icnt), (F.tV.t, icnt), (F.tU.t, ucnt)):
        sel = torch.from_numpy(cnt & (tab.shape[0] - 1)).cuda().long()
        idx = torch.arange(len(cnt), device='cuda')
        tags = _tags(tab)[sel, idx].cpu().numpy()
        assert (tags == cnt.reshape(-1, 1)).all()
    for tail, rd, cnt in ((F.tU, F.tU.rd, ucnt), (F.tV, F.tV.rd, icnt)):              # expect[buffer] of the current version = rd[buffer]
        cur = tail.t.view(torch.int32)[torch.from_numpy(cnt & (tail.bufs - 1)).cuda().long(), torch.arange(len(cnt), device='cuda')]
        np.testing.assert_array_equal(cur[:, 2:2 + tail.bufs, 0].cpu().numpy(), rd.cpu().numpy().reshape(-1, tail.bufs))


@pytest.mark.parametrize('k,B,nb,mode,lr', [(16, 64, 12, 'l2', 0.05), (128, 256, 10, 'l2', 0.05), (50, 256, 6, 'l1', 0.05),
                                           (200, 128, 4, 'l2', 1e-4), (64, 1024, 5, 'l2', 0.05), (128, 256, 40, 'l1', 0.02),
                                           (256, 64, 6, 'l2', 0.05)])
@pytest.mark.parametrize('kernel', ['f', 'o', 'o8', 's', 's8', 'w', 'l', 'l8'])
def test_bpr_flow_parity(hip, k, B, nb, mode, lr, kernel):
    n_users, n_items = 400, 120               # small tables: every item is updated in (almost) every batch, many rows have > 4 occurrences
    tr, tr_users = _toy(n_users, n_items, seed=k + B)
    rng = np.random.Generator(np.random.PCG64(k))
    ref = R.init_bpr_state(n_users, n_items, k, rng)
    ref['b'][:] = (rng.standard_normal(n_items) * 0.01).astype(np.float32)
    hp = dict(lu=2.5e-3, li=2.5e-3, lj=2.5e-4, lb=1e-3, lr=lr, mode=mode)
    F = _Flow(hip, ref, n_users, n_items, k, hp)
    plan, exp, _, _ = _plan(hip, tr, tr_users, n_users, n_items, 42, 0, nb, B, owners=_owners(hip, kernel, n_items, k))
    loss = torch.zeros(nb, device='cuda')
    F.run(plan, B, nb, loss, kernel=kernel)
    ucnt, icnt, uocc, iocc, ref_loss = _oracle(ref, exp, n_users, n_items, nb, B, hp)
    # (the scalar-exchange forms round <u, v_i> + b_i and <u, v_j> + b_j separately before they subtract: a few of 30,000 elements land
    # 1.6e-5 from the oracle at lr = 0.05, where RMSProp's first steps are +-lr whatever the gradient's size)
    _check(F, ref, ucnt, icnt, uocc, iocc, **(dict(tol=dict(rtol=2e-4, atol=3e-5)) if kernel[0] in 'sw' else {}))
    np.testing.assert_allclose(loss.cpu().numpy(), ref_loss, rtol=1e-4)
    heavy = (P.sample_and_plan.last_flow['prec'][:, :, 2] > 4).sum()
    assert B < 256 or heavy > 0               # the walk over more than 4 occurrences is exercised


@pytest.mark.parametrize('bufs', [2, 4])
@pytest.mark.parametrize('kernel', ['f', 'o', 'o3', 's', 's3', 'w', 'l', 'l3'])
def test_flow_is_deterministic_and_launch_split_invariant(hip, kernel, bufs):
    """bitwise: two runs of one launch, and the same chunk cut into launches of 1 + 3 + the rest and 2 + 1 + 4 + the rest (K2o: the
    rows an owner holds in LDS do not outlive a launch; the first task of a row in the next launch takes it from the tables again)"""
    n_users, n_items, k, B, nb = 300, 80, 128, 256, 12
    tr, tr_users = _toy(n_users, n_items, seed=9)
    hp = dict(lu=2.5e-3, li=2.5e-3, lj=2.5e-4, lb=0.0, lr=0.01, mode='l2')
    plan, exp, _, _ = _plan(hip, tr, tr_users, n_users, n_items, 3, 0, nb, B, owners=_owners(hip, kernel, n_items, k))
    outs = []
    for cuts in ((nb,), (nb,), (1, 3, nb - 4), (2, 1, 4, nb - 7)):
        ref = R.init_bpr_state(n_users, n_items, k, np.random.Generator(np.random.PCG64(0)))
        F = _Flow(hip, ref, n_users, n_items, k, hp, bufs)
        at = 0
        for m in cuts:
            F.run(plan, B, m, None, first=at, kernel=kernel)
            at += m
        assert F.status()[0] == 0
        outs.append(F.raw())
    for a, b, c, d in zip(*outs):
        assert torch.equal(a, b) and torch.equal(a, c) and torch.equal(a, d)


@pytest.mark.parametrize('bufs', [2, 4])
@pytest.mark.parametrize('kernel,waves_per_cu', [('f', 4), ('f', 8), ('f', 12), ('o', 0), ('o', 1), ('o', 6), ('o3', 2), ('s', 0), ('s', 1), ('s3', 2),
                                                 ('w', 0), ('w', 13), ('l', 0), ('l', 1), ('l3', 2)])
def test_flow_few_waves_and_hot_rows(hip, kernel, waves_per_cu, bufs):
    """12 items: every item row is rewritten in every batch (a hand-off chain through all 64 batches), all of them with dozens of
    occurrences; and the result must not depend on how many waves run"""
    n_users, n_items, k, B, nb = 500, 12, 64, 128, 64
    rng = np.random.Generator(np.random.PCG64(1))
    tr = {u: [int(x) for x in rng.choice(n_items, 3, replace=False)] for u in range(n_users)}
    tr_users = list(tr.keys())
    ref = R.init_bpr_state(n_users, n_items, k, rng)
    hp = dict(lu=2.5e-3, li=2.5e-3, lj=2.5e-4, lb=1e-3, lr=0.02, mode='l2')
    F = _Flow(hip, ref, n_users, n_items, k, hp, bufs)
    plan, exp, _, _ = _plan(hip, tr, tr_users, n_users, n_items, 77, 0, nb, B, owners=_owners(hip, kernel, n_items, k))
    F.run(plan, B, nb, None, waves_per_cu=waves_per_cu, kernel=kernel)
    ucnt, icnt, uocc, iocc, _ = _oracle(ref, exp, n_users, n_items, nb, B, hp)
    assert icnt.min() >= nb - 2
    _check(F, ref, ucnt, icnt, uocc, iocc, tol=dict(rtol=5e-4, atol=2e-5))


@pytest.mark.parametrize('kernel', ['f', 'o', 's', 'l'])
def test_flow_sgd(hip, kernel):
    n_users, n_items, k, B, nb = 400, 120, 128, 256, 8
    tr, tr_users = _toy(n_users, n_items, seed=5)
    rng = np.random.Generator(np.random.PCG64(2))
    ref = R.init_bpr_state(n_users, n_items, k, rng)
    hp = dict(lu=2.5e-3, li=2.5e-3, lj=2.5e-4, lb=1e-3, lr=0.05, mode='l2', opt='sgd')
    F = _Flow(hip, ref, n_users, n_items, k, hp)
    F.st.msU = F.st.msV = None                # the slots are neither read nor written
    plan, exp, _, _ = _plan(hip, tr, tr_users, n_users, n_items, 43, 0, nb, B, owners=_owners(hip, kernel, n_items, k))
    loss = torch.zeros(nb, device='cuda')
    F.run(plan, B, nb, loss, kernel=kernel)
    ucnt, icnt, uocc, iocc, ref_loss = _oracle(ref, exp, n_users, n_items, nb, B, hp)
    _check(F, ref, ucnt, icnt, uocc, iocc, slots=False)
    np.testing.assert_allclose(loss.cpu().numpy(), ref_loss, rtol=1e-4)
    st = hip.FlowState()
    assert hip.lib().tkr_bpr_flow_run(C.byref(st), None, None, 256, 1, None, None, 0, None) == -1


def test_engine_layouts_and_piecewise_plans():
    """BprEngine: batch 256 runs on the granule layout, batch 4096 on the plain one, the model survives the conversions; a plan
    consumed in pieces equals one consumed at once; the counters seen from outside are those of the batches that RAN"""
    from single import _engine
    n_users, n_items, k = 900, 200, 64
    tr, tr_users = _toy(n_users, n_items, seed=3)
    row_ptr, pos, srt = P.build_csr(tr, n_users)
    dev = torch.device('cuda')
    csr = _engine.TrainingCSR.from_arrays(row_ptr, pos, np.asarray(tr_users, np.int32), dev)
    hp = dict(lu=2.5e-3, li=2.5e-3, lj=2.5e-4, lb=1e-3, lr=0.02, mode='l2')

    def fresh():
        return _engine.BprEngine(n_users, n_items, k, hp, dev, seed=21)

    a = fresh()
    init = {n: a.get(n)[0].cpu().numpy() for n in ('U', 'V', 'b')}
    a.run_batches(csr, 30, 256, want_loss=False)
    assert a.layout == 'flow'
    b = fresh()
    for m in (1, 4, 25):
        b.run_batches(csr, m, 256, want_loss=False)
        b.check()
    for n in ('U', 'V', 'b'):
        for x, y in zip(a.get(n), b.get(n)):
            assert torch.equal(x, y), n
    u, i, j = P.sample_triplets(tr_users, row_ptr, pos, srt, n_items, 21, 0, 30 * 256)
    ub = np.unique(np.stack([np.repeat(np.arange(30), 256), u], 1), axis=0)
    np.testing.assert_array_equal(a.cnt.ucnt.cpu().numpy(), np.bincount(ub[:, 1], minlength=n_users))     # 30 batches, not the 512 planned
    assert a.triplets_drawn == 30 * 256
    # the oracle on the same stream
    ref = dict(U=init['U'].copy(), V=init['V'].copy(), b=init['b'].copy(), msU=np.ones_like(init['U']), msV=np.ones_like(init['V']),
               msb=np.ones_like(init['b']))
    for bb in range(30):
        sl = slice(bb * 256, (bb + 1) * 256)
        R.bpr_step(ref, u[sl], i[sl], j[sl], hp)
    for n in ('U', 'V', 'b'):
        np.testing.assert_allclose(a.get(n)[0].cpu().numpy(), ref[n], rtol=2e-4, atol=1e-5, err_msg=n)
    # on to a large batch: tables convert to the plain layout, values and slots carried over, and back again
    before = {n: [t.clone() for t in a.get(n)] for n in ('U', 'V', 'b')}
    a.prepare(4096)
    assert a.layout == 'bulk'
    for n in ('U', 'V', 'b'):
        for x, y in zip(before[n], a.get(n)):
            assert torch.equal(x, y), n
    a.run_batches(csr, 3, 4096, want_loss=False)
    a.run_batches(csr, 5, 256, want_loss=False)
    a.check()
    assert a.layout == 'flow' and a.triplets_drawn == 30 * 256 + 3 * 4096 + 5 * 256
    u2, i2, j2 = P.sample_triplets(tr_users, row_ptr, pos, srt, n_items, 21, 30 * 256, 3 * 4096 + 5 * 256)
    for lo, B_ in [(q * 4096, 4096) for q in range(3)] + [(3 * 4096 + q * 256, 256) for q in range(5)]:
        R.bpr_step(ref, u2[lo:lo + B_], i2[lo:lo + B_], j2[lo:lo + B_], hp)
    for n in ('U', 'V', 'b'):
        np.testing.assert_allclose(a.get(n)[0].cpu().numpy(), ref[n], rtol=3e-4, atol=2e-5, err_msg=n)


def test_short_calls_in_one_c_call_equal_two(monkeypatch):
    """a short call sends K1 and the step out in ONE C call (tkr_bpr_own_plan_run); a long one plans and steps separately: the same
    tables AND losses bit for bit (K2o's per-task loss sums are added up in a fixed order: csrc/bpr_own.hip own_loss_kernel), the same counters -- also across a settle() that finds a chunk whose K1 never left"""
    from single import _engine
    n_users, n_items, k = 900, 200, 64
    tr, tr_users = _toy(n_users, n_items, seed=3)
    row_ptr, pos, srt = P.build_csr(tr, n_users)
    dev = torch.device('cuda')
    csr = _engine.TrainingCSR.from_arrays(row_ptr, pos, np.asarray(tr_users, np.int32), dev)
    hp = dict(lu=2.5e-3, li=2.5e-3, lj=2.5e-4, lb=1e-3, lr=0.02, mode='l2')
    out = {}
    for fuse in (True, False):
        monkeypatch.setenv('TKR_FUSE_SHORT', '1' if fuse else '0')
        e = _engine.BprEngine(n_users, n_items, k, hp, dev, seed=21)
        losses = []
        for m in (1, 20, 7, 3):
            losses.append(e.run_batches(csr, m, 256, want_loss=True).clone())
            e.settle()
        assert e._plan_owners(256) > 0
        out[fuse] = ([t.clone() for n in ('U', 'V', 'b') for t in e.get(n)], torch.cat(losses), e.cnt.ucnt.clone(), e.cnt.icnt.clone())
    for x, y in zip(out[True][0], out[False][0]):
        assert torch.equal(x, y)
    assert torch.equal(out[True][1], out[False][1])
    assert torch.equal(out[True][2], out[False][2]) and torch.equal(out[True][3], out[False][3])


def test_the_same_short_call_again_skips_planning_decisions_not_work(monkeypatch):
    """BprEngine._run_again: a short call repeated (same data, batch size and count) goes straight to the C call of the other plan
    buffer.  Tables, losses, counters and stream position equal those of an engine that plans and steps in separate calls, also
    with a settle(), another batch count, other training data and a changed switch in between (each must take the long way once)."""
    from single import _engine
    n_users, n_items, k = 900, 200, 64
    tr, tr_users = _toy(n_users, n_items, seed=3)
    row_ptr, pos, srt = P.build_csr(tr, n_users)
    dev = torch.device('cuda')
    csr = _engine.TrainingCSR.from_arrays(row_ptr, pos, np.asarray(tr_users, np.int32), dev)
    csr2 = _engine.TrainingCSR.from_arrays(row_ptr, pos, np.asarray(tr_users, np.int32), dev)
    hp = dict(lu=2.5e-3, li=2.5e-3, lj=2.5e-4, lb=1e-3, lr=0.02, mode='l2')
    out = {}
    for fuse in (True, False):
        monkeypatch.setenv('TKR_FUSE_SHORT', '1' if fuse else '0')
        e = _engine.BprEngine(n_users, n_items, k, hp, dev, seed=21)
        real, taken = e._run_again, []

        def spy(*a, _real=real, _taken=taken):
            r = _real(*a)
            _taken.append(r is not False)
            return r
        e._run_again = spy
        losses = []
        script = ([(csr, 20)] * 4 + [('settle', 0)] + [(csr, 20)] * 2 + [(csr, 7)] * 3 + [(csr2, 7)] * 2 + [('waves', 0)] + [(csr2, 7)] * 2 +
                  [(csr2, 600), (csr2, 5)])
        for what, m in script:
            if what == 'settle':
                e.settle()
            elif what == 'waves':
                e.cfg.own_waves = 7
            else:
                losses.append(e.run_batches(what, m, 256, want_loss=True).clone())
        e.check()
        if fuse:
            # asked (what the last call left behind fits) and answered: 4x20: -, no (the other buffer's call is not there yet), yes, yes |
            # settle | yes, yes | 3x7: yes, yes, yes | other data: -, no (the other buffer's call names the first data), | switch: -, yes |
            # more than a chunk: no | -
            assert taken == [False, True, True, True, True, True, True, True, False, True, False], taken
        else:
            assert not any(taken)
        out[fuse] = ([t.clone() for n in ('U', 'V', 'b') for t in e.get(n)], torch.cat(losses), e.cnt.ucnt.clone(), e.cnt.icnt.clone(),
                     e.triplets_drawn)
    for x, y in zip(out[True][0], out[False][0]):
        assert torch.equal(x, y)
    assert torch.equal(out[True][1], out[False][1])
    assert torch.equal(out[True][2], out[False][2]) and torch.equal(out[True][3], out[False][3])
    assert out[True][4] == out[False][4] == (6 * 20 + 7 * 7 + 605) * 256


@pytest.mark.parametrize('bufs', [2, 4])
def test_fused_exchange_of_the_granule_tables(monkeypatch, bufs):
    """dist.ItemSync on the dataflow layout (tkr_sync_flow_snapshot / pack / unpack) against the same exchange through get /
    set_replicated: two 'ranks' that did the same work (the all-reduce doubles the packed vector) must leave V = start + 2 * delta,
    b alike, the slots unchanged, and the tables in the freshly assigned state -- bit for bit, and training goes on from there
    exactly as from tables assigned the slow way."""
    from single import _engine
    import dist as tdist
    n_users, n_items, k = 700, 300, 64
    tr, tr_users = _toy(n_users, n_items, seed=5)
    row_ptr, pos, srt = P.build_csr(tr, n_users)
    dev = torch.device('cuda')
    csr = _engine.TrainingCSR.from_arrays(row_ptr, pos, np.asarray(tr_users, np.int32), dev)
    hp = dict(lu=2.5e-3, li=2.5e-3, lj=2.5e-4, lb=1e-3, lr=0.02, mode='l2')
    monkeypatch.setenv('TKR_FLOW_ITEM_BUFS', str(bufs))
    monkeypatch.setattr(tdist, 'world', lambda: (0, 2))
    monkeypatch.setattr(tdist.dist, 'all_reduce', lambda t, op=None, group=None: t.mul_(2.0))     # two ranks with identical deltas

    def run(fused):
        eng = _engine.BprEngine(n_users, n_items, k, hp, dev, seed=9)
        eng.run_batches(csr, 7, 256, want_loss=False)
        if not fused:
            monkeypatch.setattr(eng, 'flow_sync_tables', lambda: None)
        sync = tdist.ItemSync(eng)
        assert (sync.flow is not None) == fused
        sync.begin()
        start = {n: [t.clone() for t in eng.get(n)] for n in ('V', 'b')}
        eng.run_batches(csr, 9, 256, want_loss=False)
        cur = {n: [t.clone() for t in eng.get(n)] for n in ('V', 'b')}
        sync.end()
        eng.check()
        for n in ('V', 'b'):
            got_p, got_ms = eng.get(n)
            assert torch.equal(got_p, start[n][0] + 2.0 * (cur[n][0] - start[n][0])), n
            assert torch.equal(got_ms, 2.0 * (cur[n][1] * 0.5)), n
        assert int(eng.cnt.icnt.abs().sum()) == 0
        eng.run_batches(csr, 11, 256, want_loss=False)
        eng.check()
        return eng

    a, b = run(True), run(False)
    for x, y in ((a.V.p, b.V.p), (a.V.ms, b.V.ms), (a.tailV.t, b.tailV.t), (a.U.p, b.U.p)):       # granules incl. tags, both buffers
        assert torch.equal(x.view(torch.int32), y.view(torch.int32))
    assert torch.equal(a.tailV.rd, b.tailV.rd) and torch.equal(a.cnt.icnt, b.cnt.icnt)
    # the unpack leaves the next epoch's start in the snapshot buffer and begin() then launches nothing -- unless somebody wrote the
    # item tables in between: through the engine (set_items) or behind its back (a direct assign on the table objects)
    for how in ('set_items', 'assign'):
        eng = _engine.BprEngine(n_users, n_items, k, hp, dev, seed=9)
        eng.run_batches(csr, 3, 256, want_loss=False)
        sync = tdist.ItemSync(eng)
        sync.begin()
        eng.run_batches(csr, 4, 256, want_loss=False)
        sync.end()
        newV = torch.randn(n_items, k, device=dev) * 0.03
        if how == 'set_items':
            eng.set_items(V=newV)
        else:
            eng.settle()
            eng.V.assign(newV, torch.ones(n_items, k, device=dev))
        sync.begin()
        assert torch.equal(sync.start_flat[:n_items * k].view(n_items, k), newV), how


def test_engine_picks_the_persistent_step_by_shape_and_batch():
    """BprEngine: K2o (owned item rows) up to batch 256 where the item table fits the CUs' LDS, K2f above that batch size and for
    item tables that do not fit, K2 (plain layout) for large batches -- and every one of them trains the same model"""
    import tkr_hip
    from single import _engine
    dev = torch.device('cuda')
    hp = dict(lu=2.5e-3, li=2.5e-3, lj=2.5e-4, lb=1e-3, lr=0.02, mode='l2')
    assert tkr_hip.bpr_own_owners(10380, 128) > 0 and tkr_hip.bpr_own_owners(17770, 128) > 0        # the benchmark shapes fit
    assert tkr_hip.bpr_own_owners(400000, 128) == 0 and tkr_hip.bpr_own_owners(40000, 256) == 0       # ... these do not: K2f
    n_users, n_items, k = 900, 300, 64
    tr, tr_users = _toy(n_users, n_items, seed=4)
    row_ptr, pos, srt = P.build_csr(tr, n_users)
    csr = _engine.TrainingCSR.from_arrays(row_ptr, pos, np.asarray(tr_users, np.int32), dev)
    eng = _engine.BprEngine(n_users, n_items, k, hp, dev, seed=5)
    init = {n: eng.get(n)[0].cpu().numpy() for n in ('U', 'V', 'b')}
    eng.run_batches(csr, 6, 256, want_loss=False)
    assert eng.layout == 'flow' and eng._plan_owners(256) > 0 and eng.plan.owners > 0          # K2o
    eng.run_batches(csr, 3, 512, want_loss=False)
    assert eng.layout == 'flow' and eng._plan_owners(512) == 0 and eng.plan.owners == 0        # K2f
    eng.run_batches(csr, 4, 128, want_loss=False)
    assert eng.plan.owners > 0                                                                 # K2o again, same tables
    eng.check()
    u, i, j = P.sample_triplets(tr_users, row_ptr, pos, srt, n_items, 5, 0, 6 * 256 + 3 * 512 + 4 * 128)
    ref = dict(U=init['U'].copy(), V=init['V'].copy(), b=init['b'].copy(), msU=np.ones_like(init['U']), msV=np.ones_like(init['V']),
               msb=np.ones_like(init['b']))
    at = 0
    for nb, B in ((6, 256), (3, 512), (4, 128)):
        for _ in range(nb):
            R.bpr_step(ref, u[at:at + B], i[at:at + B], j[at:at + B], hp)
            at += B
    for n in ('U', 'V', 'b'):
        np.testing.assert_allclose(eng.get(n)[0].cpu().numpy(), ref[n], rtol=3e-4, atol=2e-5, err_msg=n)


def test_a_timed_out_spin_steps_down_k2o_k2f_k2():
    """a bounded spin that runs out (injected through the status word) is reported at the next check and moves the engine one step
    down: K2o (a 12-wave workgroup resident on every CU) -> K2f (4 waves per CU) -> K2 (one launch per batch); the engine stays
    usable once the caller has put valid tables back (a launch is not transactional)"""
    import tkr_hip
    from single import _engine
    dev = torch.device('cuda')
    hp = dict(lu=2.5e-3, li=2.5e-3, lj=2.5e-4, lb=1e-3, lr=0.02, mode='l2')
    n_users, n_items, k = 700, 250, 64
    tr, tr_users = _toy(n_users, n_items, seed=6)
    row_ptr, pos, srt = P.build_csr(tr, n_users)
    csr = _engine.TrainingCSR.from_arrays(row_ptr, pos, np.asarray(tr_users, np.int32), dev)
    eng = _engine.BprEngine(n_users, n_items, k, hp, dev, seed=8)
    keep = {n: [t.clone() for t in eng.get(n)] for n in ('U', 'V', 'b')}

    def restore():
        eng.set_users(U=keep['U'][0], msU=keep['U'][1])
        eng.set_items(V=keep['V'][0], b=keep['b'][0], msV=keep['V'][1], msb=keep['b'][1])

    def inject():
        eng._flow_ran = True
        eng.ctl[tkr_hip.FLOW_CTL_STATUS] = 1
        with pytest.raises(tkr_hip.TkrError):
            eng.check()
        restore()
    eng.run_batches(csr, 5, 256, want_loss=False)
    eng.check()
    assert eng.layout == 'flow' and eng.plan.owners > 0
    inject()
    eng.run_batches(csr, 5, 256, want_loss=False)
    eng.check()
    assert eng.layout == 'flow' and eng.plan.owners == 0                 # K2f
    inject()
    eng.run_batches(csr, 5, 256, want_loss=False)
    eng.check()
    assert eng.layout == 'bulk'                                          # K2
    # ... and what it trains from the restored tables is the model: against the oracle on the stream the engine drew last
    ref = dict(U=keep['U'][0].cpu().numpy().copy(), V=keep['V'][0].cpu().numpy().copy(), b=keep['b'][0].cpu().numpy().copy(),
               msU=keep['U'][1].cpu().numpy().copy(), msV=keep['V'][1].cpu().numpy().copy(), msb=keep['b'][1].cpu().numpy().copy())
    u, i, j = (eng.plan.u[:5 * 256].cpu().numpy(), eng.plan.i[:5 * 256].cpu().numpy(), eng.plan.j[:5 * 256].cpu().numpy())
    for q in range(5):
        R.bpr_step(ref, u[q * 256:(q + 1) * 256], i[q * 256:(q + 1) * 256], j[q * 256:(q + 1) * 256], hp)
    for n in ('U', 'V', 'b'):
        np.testing.assert_allclose(eng.get(n)[0].cpu().numpy(), ref[n], rtol=3e-4, atol=2e-5, err_msg=n)


def test_k2o_at_netflix_width_and_k256(monkeypatch):
    """17,770 items x k = 256: 70 rows of 2 KB per owner = 154 KB of a CU's 160 KB of LDS, the 8-wave form of K2o (k > 128) -- against
    K2f on the same stream (the two group the occurrences of a heavy row differently: equal to rounding)"""
    import synth
    from single import _engine
    dev = torch.device('cuda')
    n_users, n_items, k = 30000, 17770, 256
    row_ptr, pos, _, tr_users = synth.train_csr_shape(n_users, n_items, mean_pos=30.0, seed=3)
    csr = _engine.TrainingCSR.from_arrays(row_ptr, pos, tr_users, dev)
    hp = dict(lu=2.5e-3, li=2.5e-3, lj=2.5e-4, lb=1e-3, lr=0.01, mode='l2')
    outs = []
    for own in ('1', '0'):
        monkeypatch.setenv('TKR_OWN', own)
        eng = _engine.BprEngine(n_users, n_items, k, hp, dev, seed=11)
        eng.run_batches(csr, 300, 256, want_loss=False)
        eng.check()
        assert eng.layout == 'flow' and (eng.plan.owners > 0) == (own == '1')
        outs.append([eng.get(n)[0].clone() for n in ('U', 'V', 'b')])
        del eng
    for a, b, n in zip(outs[0], outs[1], 'UVb'):
        assert torch.allclose(a, b, rtol=2e-4, atol=1e-6), n
