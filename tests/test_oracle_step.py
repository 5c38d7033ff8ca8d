"""Analytic checks of the UNPINNED part of the oracle (the TF-1.15 train step cannot run
here): gradients vs torch-CPU autograd on the literal loss expressions of
single/bpr.py:93-99 and single/vbpr.py:64-72, and a hand-worked RMSProp known answer."""
import numpy as np
import pytest
import torch

from oracle import ref_np as R

HP = dict(lu=2.5e-3, li=2.5e-3, lj=2.5e-4, lb=1e-3, le=1e-3, lr=0.05)   # large lr: P-P' resolves g in fp32


def _batch(rng, n_users, n_items, B):
    ub = rng.integers(0, n_users, B)
    ib = rng.integers(0, n_items, B)
    jb = (ib + 1 + rng.integers(0, n_items - 1, B)) % n_items
    return ub, ib, jb


def _implied_grad(before, after, ms_after, lr):
    """invert P' = P - lr*g/sqrt(ms'+eps) on touched rows."""
    return (before - after) * np.sqrt(ms_after + R.EPS) / np.float32(lr)


@pytest.mark.parametrize('mode', ['l2', 'l1'])
def test_bpr_step_matches_autograd(mode):
    rng = np.random.Generator(np.random.PCG64(0))
    n_users, n_items, k, B = 50, 30, 16, 64          # many duplicate rows inside the batch
    st = R.init_bpr_state(n_users, n_items, k, rng)
    st['b'][:] = (rng.standard_normal(n_items) * 0.01).astype(np.float32)
    ub, ib, jb = _batch(rng, n_users, n_items, B)
    hp = dict(HP, mode=mode)
    U = torch.tensor(st['U'], dtype=torch.float64, requires_grad=True)
    V = torch.tensor(st['V'], dtype=torch.float64, requires_grad=True)
    b = torch.tensor(st['b'], dtype=torch.float64, requires_grad=True)
    tu, ti, tj = (torch.tensor(x) for x in (ub, ib, jb))
    ue, ie, je, bi, bj = U[tu], V[ti], V[tj], b[ti], b[tj]
    x = bi - bj + (ue * ie).sum(1) - (ue * je).sum(1)
    obj = torch.log(1 + torch.exp(-x)).sum()
    if mode == 'l2':
        obj = obj + 0.5 * (ue ** 2 * hp['lu'] + ie ** 2 * hp['li'] + je ** 2 * hp['lj']).sum() \
            + 0.5 * (bi ** 2 + bj ** 2).sum() * hp['lb']
    else:
        obj = obj + (ue.abs() * hp['lu'] + ie.abs() * hp['li'] + je.abs() * hp['lj']).sum() \
            + (bi.abs() + bj.abs()).sum() * hp['lb']
    obj.backward()
    before = {n: st[n].copy() for n in ('U', 'V', 'b')}
    loss = R.bpr_step(st, ub, ib, jb, hp)
    assert abs(float(loss) - float(obj.detach())) < 2e-4 * max(1.0, abs(float(obj.detach())))
    for name, grad, ms in (('U', U.grad, 'msU'), ('V', V.grad, 'msV'), ('b', b.grad, 'msb')):
        g = grad.numpy()
        touched = np.abs(g).reshape(len(g), -1).sum(1) > 0
        got = _implied_grad(before[name], st[name], st[ms], hp['lr'])
        np.testing.assert_allclose(got[touched], g[touched], rtol=2e-3, atol=2e-6)
        # untouched rows: parameter and slot unchanged (sparse / lazy update)
        np.testing.assert_array_equal(st[name][~touched], before[name][~touched])
        assert np.all(st[ms][~touched] == 1.0)
        # ms of touched rows: 0.9*1 + 0.1*g^2
        np.testing.assert_allclose(st[ms][touched], 0.9 + 0.1 * g[touched] ** 2, rtol=1e-5)


def test_rmsprop_known_answer():
    """Hand-worked: one triplet, k=1, U=[1], V=[2,0], b=0, lambda=0 -> x = 2, s = 1/(1+e^2).
    g_U = -s*(2-0); ms = 0.9 + 0.1 g^2; U' = 1 - lr*g/sqrt(ms + 1e-10)."""
    st = dict(U=np.array([[1.0]], np.float32), V=np.array([[2.0], [0.0]], np.float32), b=np.zeros(2, np.float32),
              msU=np.ones((1, 1), np.float32), msV=np.ones((2, 1), np.float32), msb=np.ones(2, np.float32))
    hp = dict(lu=0, li=0, lj=0, lb=0, lr=0.1, mode='l2')
    loss = R.bpr_step(st, [0], [0], [1], hp)
    s = 1.0 / (1.0 + np.exp(2.0))
    assert abs(loss - np.log1p(np.exp(-2.0))) < 1e-6
    gU, gVi, gVj, gbi, gbj = -2 * s, -s, s, -s, s
    for val, g, p0 in ((st['U'][0, 0], gU, 1.0), (st['V'][0, 0], gVi, 2.0), (st['V'][1, 0], gVj, 0.0),
                       (st['b'][0], gbi, 0.0), (st['b'][1], gbj, 0.0)):
        ms = 0.9 + 0.1 * g * g
        assert abs(val - (p0 - 0.1 * g / np.sqrt(ms + 1e-10))) < 1e-6
    # first-step magnitude: ms0 = 1 -> |dP| ~ lr*|g|/sqrt(0.9 + 0.1 g^2)  
    assert abs((st['U'][0, 0] - 1.0) - 0.1 * abs(gU) / np.sqrt(0.9 + 0.1 * gU * gU)) < 1e-6   # g<0: U grows


@pytest.mark.parametrize('mode', ['l2', 'l1'])
def test_vbpr_step_matches_autograd(mode):
    rng = np.random.Generator(np.random.PCG64(1))
    n_users, n_items, k, d, B = 40, 25, 8, 12, 48
    st = R.init_vbpr_state(n_users, n_items, k, d, rng)
    st['irb'][:] = (rng.standard_normal(n_items) * 0.01).astype(np.float32)
    st['icb'][:] = (rng.standard_normal(d) * 0.01).astype(np.float32)
    st['cem'][:] = (rng.standard_normal((d, k // 2)) * 0.05).astype(np.float32)
    feat = np.abs(rng.standard_normal((n_items, d))).astype(np.float32)
    ub, ib, jb = _batch(rng, n_users, n_items, B)
    hp = dict(HP, mode=mode)
    # the variables WITH the reference's shapes: item_rating_bias [n_items, 1], item_content_bias [d, 1] (vbpr.py:43,47) -- the
    # expressions below are vbpr.py:50-72 token for token, and torch broadcasts like TensorFlow: x_uij comes out [B, B]
    T = {n: torch.tensor(st[n].reshape(-1, 1) if n in ('irb', 'icb') else st[n], dtype=torch.float64, requires_grad=True)
         for n in ('ure', 'uce', 'ire', 'irb', 'cem', 'icb')}
    F = torch.tensor(feat, dtype=torch.float64)
    tu, ti, tj = (torch.tensor(x) for x in (ub, ib, jb))
    ur, uc, ir, jr, bi, bj = T['ure'][tu], T['uce'][tu], T['ire'][ti], T['ire'][tj], T['irb'][ti], T['irb'][tj]
    ic, jc = F[ti], F[tj]
    ice, jce = ic @ T['cem'], jc @ T['cem']
    x_ui = (ur * ir + uc * ice).sum(1)
    x_uj = (ur * jr + uc * jce).sum(1)
    x = bi - bj + x_ui - x_uj + torch.matmul(ic - jc, T['icb'])
    assert x.shape == (B, B) and bi.shape == (B, 1) and x_ui.shape == (B,)
    obj = torch.log(1 + torch.exp(-x)).sum()
    if mode == 'l2':
        obj = obj + 0.5 * (T['cem'] ** 2).sum() * hp['le'] \
            + 0.5 * ((ur ** 2 + uc ** 2) * hp['lu'] + ir ** 2 * hp['li'] + jr ** 2 * hp['lj']).sum() \
            + 0.5 * ((bi ** 2 + bj ** 2).sum() + (T['icb'] ** 2).sum()) * hp['lb']
    else:
        obj = obj + T['cem'].abs().sum() * hp['le'] \
            + ((ur.abs() + uc.abs()) * hp['lu'] + ir.abs() * hp['li'] + jr.abs() * hp['lj']).sum() \
            + ((bi.abs() + bj.abs()).sum() + T['icb'].abs().sum()) * hp['lb']
    obj.backward()
    before = {n: st[n].copy() for n in T}
    loss = R.vbpr_step(st, feat, ub, ib, jb, hp)
    assert abs(float(loss) - float(obj.detach())) < 2e-4 * max(1.0, abs(float(obj.detach())))
    for name in T:
        g = T[name].grad.numpy().reshape(st[name].shape)
        got = _implied_grad(before[name], st[name], st['ms_' + name], hp['lr'])
        if name in ('cem', 'icb'):                         # dense: every element updated
            np.testing.assert_allclose(got, g, rtol=3e-3, atol=3e-6)
            np.testing.assert_allclose(st['ms_' + name], 0.9 + 0.1 * g ** 2, rtol=1e-5)
        else:
            touched = np.abs(g).reshape(len(g), -1).sum(1) > 0
            np.testing.assert_allclose(got[touched], g[touched], rtol=3e-3, atol=3e-6)
            np.testing.assert_array_equal(st[name][~touched], before[name][~touched])
    fue, fie, fib = R.vbpr_fold(st, feat)
    assert fue.shape == (n_users, k) and fie.shape == (n_items, k) and fib.shape == (n_items, 1)
    # folded factors reproduce x_ui up to the user-independent terms
    xui = (st['ure'][ub] * st['ire'][ib]).sum(1) + (st['uce'][ub] * (feat[ib] @ st['cem'])).sum(1) \
        + st['irb'][ib] + feat[ib] @ st['icb']
    np.testing.assert_allclose((fue[ub] * fie[ib]).sum(1) + fib[ib, 0], xui, rtol=1e-4, atol=1e-6)


def test_sgd_step_matches_autograd():
    """legacy optimiser (old/methods/bpr.py:43-61): P' = P - lr * dcost/dP, exactly the autograd gradient"""
    rng = np.random.Generator(np.random.PCG64(4))
    n_users, n_items, k, B = 40, 25, 12, 64
    st = R.init_bpr_state(n_users, n_items, k, rng)
    st['b'][:] = (rng.standard_normal(n_items) * 0.01).astype(np.float32)
    ub, ib, jb = _batch(rng, n_users, n_items, B)
    hp = dict(HP, mode='l2', opt='sgd')
    U = torch.tensor(st['U'], dtype=torch.float64, requires_grad=True)
    V = torch.tensor(st['V'], dtype=torch.float64, requires_grad=True)
    b = torch.tensor(st['b'], dtype=torch.float64, requires_grad=True)
    tu, ti, tj = (torch.tensor(x) for x in (ub, ib, jb))
    x = b[ti] - b[tj] + (U[tu] * V[ti]).sum(1) - (U[tu] * V[tj]).sum(1)
    # the literal legacy objective: -(sum log sigmoid(x) - regularisers)
    cost = -(torch.log(torch.sigmoid(x)).sum() - hp['lu'] * 0.5 * (U[tu] ** 2).sum() - hp['li'] * 0.5 * (V[ti] ** 2).sum()
             - hp['lj'] * 0.5 * (V[tj] ** 2).sum() - hp['lb'] * 0.5 * (b[ti] ** 2 + b[tj] ** 2).sum())
    cost.backward()
    before = {n: st[n].copy() for n in ('U', 'V', 'b', 'msU', 'msV', 'msb')}
    loss = R.bpr_step(st, ub, ib, jb, hp)
    assert abs(float(loss) - float(cost.detach())) < 2e-4 * max(1.0, abs(float(cost.detach())))
    for name, grad in (('U', U.grad), ('V', V.grad), ('b', b.grad)):
        want = before[name].astype(np.float64) - hp['lr'] * grad.numpy()
        np.testing.assert_allclose(st[name], want, rtol=1e-5, atol=2e-7)
        np.testing.assert_array_equal(st['ms' + name], before['ms' + name])       # no slot in this mode


# ---- several steps in a row: the oracle (fp32, hand-derived gradients, segment sums) against an independent fp64 model ----------
# The fp64 model never looks at the oracle's formulas: torch autograd differentiates the literal objective w.r.t. the FULL tables
# (which sums the contributions of duplicate rows by itself), and TF-1.15's RMSProp recurrences are applied as documented
# (SURVEY.md A.2): sparse variables -- only rows that occur in the batch, ms <- 0.9 ms + 0.1 g^2, P <- P - lr g / sqrt(ms + 1e-10),
# rms slot initialised to ONE; dense variables (VBPR cem, icb) -- every element, every step.
def _rmsprop64(P, ms, g, lr, rows=None):
    if rows is None:
        ms[...] = 0.9 * ms + 0.1 * g * g
        P[...] = P - lr * g / np.sqrt(ms + 1e-10)
    else:
        ms[rows] = 0.9 * ms[rows] + 0.1 * g[rows] * g[rows]
        P[rows] = P[rows] - lr * g[rows] / np.sqrt(ms[rows] + 1e-10)


@pytest.mark.parametrize('mode', ['l2', 'l1'])
def test_bpr_many_steps_duplicate_heavy_against_fp64_recurrence(mode):
    rng = np.random.Generator(np.random.PCG64(7))
    n_users, n_items, k, B, steps = 12, 9, 8, 96, 6             # ~8 occurrences of every user and ~21 of every item per batch
    st = R.init_bpr_state(n_users, n_items, k, rng)
    st['U'] *= 10; st['V'] *= 10
    st['b'][:] = (rng.standard_normal(n_items) * 0.05).astype(np.float32)
    hp = dict(lu=2.5e-3, li=2.5e-3, lj=2.5e-4, lb=1e-3, lr=0.01, mode=mode)
    P64 = {n: st[n].astype(np.float64) for n in ('U', 'V', 'b')}
    M64 = {n: np.ones_like(P64[n]) for n in P64}
    for step in range(steps):
        ub, ib, jb = _batch(rng, n_users, n_items, B)
        T = {n: torch.tensor(P64[n], requires_grad=True) for n in P64}
        tu, ti, tj = (torch.tensor(x) for x in (ub, ib, jb))
        ue, ie, je, bi, bj = T['U'][tu], T['V'][ti], T['V'][tj], T['b'][ti], T['b'][tj]
        x = bi - bj + (ue * ie).sum(1) - (ue * je).sum(1)
        obj = torch.log(1 + torch.exp(-x)).sum()
        if mode == 'l2':
            obj = obj + 0.5 * (ue ** 2 * hp['lu'] + ie ** 2 * hp['li'] + je ** 2 * hp['lj']).sum() + 0.5 * (bi ** 2 + bj ** 2).sum() * hp['lb']
        else:
            obj = obj + (ue.abs() * hp['lu'] + ie.abs() * hp['li'] + je.abs() * hp['lj']).sum() + (bi.abs() + bj.abs()).sum() * hp['lb']
        obj.backward()
        loss = R.bpr_step(st, ub, ib, jb, hp)
        assert abs(float(loss) - float(obj.detach())) < 3e-4 * abs(float(obj.detach()))
        items = np.unique(np.concatenate([ib, jb]))
        _rmsprop64(P64['U'], M64['U'], T['U'].grad.numpy(), hp['lr'], np.unique(ub))
        _rmsprop64(P64['V'], M64['V'], T['V'].grad.numpy(), hp['lr'], items)
        _rmsprop64(P64['b'], M64['b'], T['b'].grad.numpy(), hp['lr'], items)
    for n in ('U', 'V', 'b'):
        np.testing.assert_allclose(st[n], P64[n], rtol=2e-4, atol=2e-6, err_msg=n)
        np.testing.assert_allclose(st['ms' + n], M64[n], rtol=2e-4, err_msg='ms' + n)


@pytest.mark.parametrize('mode', ['l2', 'l1'])
def test_vbpr_many_steps_against_fp64_recurrence(mode):
    rng = np.random.Generator(np.random.PCG64(8))
    n_users, n_items, k, d, B, steps = 10, 8, 6, 9, 40, 5
    st = R.init_vbpr_state(n_users, n_items, k, d, rng)
    for n in ('ure', 'uce', 'ire'):
        st[n] *= 10
    st['irb'][:] = (rng.standard_normal(n_items) * 0.05).astype(np.float32)
    st['icb'][:] = (rng.standard_normal(d) * 0.05).astype(np.float32)
    st['cem'][:] = (rng.standard_normal((d, k // 2)) * 0.1).astype(np.float32)
    feat = (np.abs(rng.standard_normal((n_items, d))) * (rng.random((n_items, d)) < 0.5)).astype(np.float32)
    hp = dict(lu=2.5e-3, li=2.5e-3, lj=2.5e-4, lb=1e-3, le=1e-3, lr=0.002, mode=mode)
    names = ('ure', 'uce', 'ire', 'irb', 'cem', 'icb')
    P64 = {n: st[n].astype(np.float64) for n in names}
    M64 = {n: np.ones_like(P64[n]) for n in names}
    F = torch.tensor(feat, dtype=torch.float64)
    for step in range(steps):
        ub, ib, jb = _batch(rng, n_users, n_items, B)
        T = {n: torch.tensor(P64[n].reshape(-1, 1) if n in ('irb', 'icb') else P64[n], requires_grad=True) for n in names}
        tu, ti, tj = (torch.tensor(x) for x in (ub, ib, jb))
        ur, uc, ir, jr, bi, bj = T['ure'][tu], T['uce'][tu], T['ire'][ti], T['ire'][tj], T['irb'][ti], T['irb'][tj]
        ic, jc = F[ti], F[tj]
        x_ui = (ur * ir + uc * (ic @ T['cem'])).sum(1)
        x_uj = (ur * jr + uc * (jc @ T['cem'])).sum(1)
        x = bi - bj + x_ui - x_uj + torch.matmul(ic - jc, T['icb'])                  # [B, B]: vbpr.py:59-61 with the reference's shapes
        obj = torch.log(1 + torch.exp(-x)).sum()
        if mode == 'l2':
            obj = obj + 0.5 * (T['cem'] ** 2).sum() * hp['le'] + 0.5 * ((ur ** 2 + uc ** 2) * hp['lu'] + ir ** 2 * hp['li'] + jr ** 2 * hp['lj']).sum() \
                + 0.5 * ((bi ** 2 + bj ** 2).sum() + (T['icb'] ** 2).sum()) * hp['lb']
        else:
            obj = obj + T['cem'].abs().sum() * hp['le'] + ((ur.abs() + uc.abs()) * hp['lu'] + ir.abs() * hp['li'] + jr.abs() * hp['lj']).sum() \
                + ((bi.abs() + bj.abs()).sum() + T['icb'].abs().sum()) * hp['lb']
        obj.backward()
        loss = R.vbpr_step(st, feat, ub, ib, jb, hp)
        assert abs(float(loss) - float(obj.detach())) < 3e-4 * abs(float(obj.detach()))
        items, users = np.unique(np.concatenate([ib, jb])), np.unique(ub)
        g = {n: T[n].grad.numpy().reshape(P64[n].shape) for n in names}
        for n, rows in (('ure', users), ('uce', users), ('ire', items), ('irb', items), ('cem', None), ('icb', None)):
            _rmsprop64(P64[n], M64[n], g[n], hp['lr'], rows)
    for n in names:
        np.testing.assert_allclose(st[n], P64[n], rtol=3e-4, atol=3e-6, err_msg=n)
        np.testing.assert_allclose(st['ms_' + n], M64[n], rtol=3e-4, err_msg='ms_' + n)
