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
    T = {n: torch.tensor(st[n], dtype=torch.float64, requires_grad=True) for n in ('ure', 'uce', 'ire', 'irb', 'cem', 'icb')}
    F = torch.tensor(feat, dtype=torch.float64)
    tu, ti, tj = (torch.tensor(x) for x in (ub, ib, jb))
    ur, uc, ir, jr, bi, bj = T['ure'][tu], T['uce'][tu], T['ire'][ti], T['ire'][tj], T['irb'][ti], T['irb'][tj]
    ic, jc = F[ti], F[tj]
    ice, jce = ic @ T['cem'], jc @ T['cem']
    x = bi - bj + (ur * ir + uc * ice).sum(1) - (ur * jr + uc * jce).sum(1) + (ic - jc) @ T['icb']
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
        g = T[name].grad.numpy()
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
