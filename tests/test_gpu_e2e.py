"""End-to-end on the GPU through the reference's class API and CLI: BPR.train on text files ->
export_embeddings -> evaluate.py, against the oracle run on the SAME init and the SAME (u,i,j)
stream (the device stream is counter-based, so oracle/plan_np reproduces it from the seed).

Stated tolerances: trained embeddings |d| <= 1e-5 + 2e-4*|x| (fp32, a few hundred sequential RMSProp
steps); accuracy@{5..30} within +-0.001 (north_star)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import plan_np as P
from oracle import ref_np as R


def _dataset(tmp_path, seed=3):
    import synth
    r = synth.make_ratings(300, 120, 30, seed=seed, mu=3.2, sigma=0.5, min_r=6, max_r=50, om_per_user=4)
    data = str(tmp_path / 'data')
    synth.write_dataset(data, r)
    return data


def _oracle_train(m, init, seed, epochs, n_batches, B, hp, ms=None):
    row_ptr, pos, srt = P.build_csr(m.tr_data, m.n_users)
    st = dict(U=init[0].copy(), V=init[1].copy(), b=init[2].reshape(-1).copy(),
              msU=np.ones_like(init[0]), msV=np.ones_like(init[1]), msb=np.ones(len(init[1]), np.float32))
    if ms is not None:
        st.update(msU=ms[0].copy(), msV=ms[1].copy(), msb=ms[2].copy())
    u, i, j = P.sample_triplets(m.tr_users, row_ptr, pos, srt, m.n_items, seed, 0, epochs * n_batches * B)
    loss = None
    for s in range(epochs * n_batches):
        sl = slice(s * B, (s + 1) * B)
        loss = R.bpr_step(st, u[sl], i[sl], j[sl], hp)
    return st, loss


def test_train_export_evaluate_matches_oracle(tmp_path, capsys):
    from single import BPR
    import evaluate as E
    data = _dataset(tmp_path)
    k, B, epochs, lr = 16, 64, 3, 0.02
    m = BPR(k=k, lambda_b=1e-3, lr=lr)
    m.load_training_data(os.path.join(data, 'uid'), os.path.join(data, 'vid'), os.path.join(data, 'f0tr.txt'))
    rng = np.random.Generator(np.random.PCG64(0))
    init = [(rng.standard_normal((m.n_users, k)) * 0.1).astype(np.float32),
            (rng.standard_normal((m.n_items, k)) * 0.1).astype(np.float32), np.zeros((m.n_items, 1), np.float32)]
    m.fue, m.fie, m.fib = (a.copy() for a in init)               # warm-start path (bpr.py:127-135) = shared init
    limit = 64 * 40 + 17                                         # 40 batches per epoch, remainder dropped (F7)
    m.train(epochs=epochs, batch_size=B, epoch_sample_limit=limit, seed=77, verbose=False)
    assert m.fue.shape == (m.n_users, k) and m.fie.shape == (m.n_items, k) and m.fib.shape == (m.n_items, 1)
    hp = dict(lu=m.lu, li=m.li, lj=m.lj, lb=m.lb, lr=lr, mode='l2')
    st, loss = _oracle_train(m, init, 77, epochs, 40, B, hp)
    np.testing.assert_allclose(m.fue, st['U'], rtol=2e-4, atol=1e-5)
    np.testing.assert_allclose(m.fie, st['V'], rtol=2e-4, atol=1e-5)
    np.testing.assert_allclose(m.fib.ravel(), st['b'], rtol=2e-4, atol=1e-5)
    assert abs(m.last_epoch_loss - float(loss)) <= 1e-4 * abs(float(loss))
    # ---- export through the text format, evaluate with the CLI, compare with the oracle pipeline
    gpu_dir, ref_dir = str(tmp_path / 'gpu_model'), str(tmp_path / 'ref_model')
    m.export_embeddings(gpu_dir)
    assert os.path.exists(os.path.join(gpu_dir, 'weights'))
    os.mkdir(ref_dir)
    R.write_embed_text(os.path.join(ref_dir, 'final-U.dat'), st['U'])
    R.write_embed_text(os.path.join(ref_dir, 'final-V.dat'), st['V'])
    R.write_embed_text(os.path.join(ref_dir, 'final-B.dat'), st['b'].reshape(-1, 1))
    got = E.main(['-d', data, '-m', gpu_dir, '-sl', 'im', 'om'])
    exp = R.evaluate_cli(data, ref_dir, scenarios=('im', 'om'))
    same_model = R.evaluate_cli(data, gpu_dir, scenarios=('im', 'om'))       # oracle scorer on the GPU-trained model
    for g, e, s in zip(got, exp, same_model):
        gv, ev, sv = ([float(x) for x in line.split(',')[1:]] for line in (g, e, s))
        assert g.split(',')[0] == e.split(',')[0]
        assert max(abs(a - b) for a, b in zip(gv, ev)) <= 1e-3               # north_star: +-0.001
        assert gv == sv                                                       # same model -> identical metric
    # the model learned something: train positives outrank sampled negatives
    assert float(loss) < 0.69 * B

    # ---- resume (bpr.py:120-135): text values override, RMSProp slots come from the checkpoint
    m2 = BPR(k=k, lambda_b=1e-3, lr=lr)
    m2.load_training_data(os.path.join(data, 'uid'), os.path.join(data, 'vid'), os.path.join(data, 'f0tr.txt'))
    m2.train(epochs=1, batch_size=B, epoch_sample_limit=limit, model_path=gpu_dir, seed=5, verbose=False)
    blob = torch.load(os.path.join(gpu_dir, 'weights'))
    init2 = [R.read_embed_text(os.path.join(gpu_dir, f), ids) for f, ids in
             (('final-U.dat', m.uids), ('final-V.dat', m.iids), ('final-B.dat', m.iids))]
    st2, _ = _oracle_train(m2, init2, 5, 1, 40, B, hp,
                           ms=[blob['ms_U'].numpy(), blob['ms_V'].numpy(), blob['ms_b'].numpy()])
    np.testing.assert_allclose(m2.fue, st2['U'], rtol=2e-4, atol=1e-5)
    np.testing.assert_allclose(m2.fie, st2['V'], rtol=2e-4, atol=1e-5)


def test_reference_defaults_and_float_limit(tmp_path):
    """train.py's literal call: epochs=5, batch_size=256, epoch_sample_limit=10e5 (a float)"""
    from single import BPR
    data = _dataset(tmp_path, seed=4)
    m = BPR(k=50)
    m.load_training_data(os.path.join(data, 'uid'), os.path.join(data, 'vid'), os.path.join(data, 'f0tr.txt'))
    m.train(epochs=1, batch_size=256, epoch_sample_limit=10e3, verbose=False)       # 39 batches
    assert m.epoch_sample_limit == 10000 and np.isfinite(m.fue).all() and np.isfinite(m.fie).all()
    assert np.any(m.fib != 0) and abs(m.fue).max() < 0.1
    gen = m._uniform_user_sampling(32)
    ub, ib, jb = next(gen)
    assert ub.dtype == np.int64 and ib.dtype == np.int32 and jb.dtype == np.int32 and len(ub) == 32
    for u, i, j in zip(ub, ib, jb):
        assert i in m.tr_data[u] and j not in m.tr_data[u]


def test_train_restarts_one_kernel_down_when_a_step_gives_up(tmp_path, monkeypatch):
    """a launch of a persistent step is not transactional: when a bounded spin runs out (injected through the status word after
    the first epoch) BPR.train puts back the state it started from and runs again on the next kernel down -- K2o -> K2f -> K2 --
    instead of raising with half-updated tables.  Same start, same counter-based stream: the result is bit for
    bit what a run that never had the upper kernels produces -- also when the launch gave up inside its planner prologue and left
    K1's touch bitmaps dirty."""
    import tkr_hip
    from single import BPR, _engine
    data = _dataset(tmp_path, seed=6)

    def model():
        m = BPR(k=16, lambda_b=1e-3, lr=0.02)
        m.load_training_data(os.path.join(data, 'uid'), os.path.join(data, 'vid'), os.path.join(data, 'f0tr.txt'))
        return m

    orig = _engine.BprEngine.run_batches
    for failures, env, layout, owners in ((1, dict(TKR_OWN='0'), 'flow', 0), (2, dict(TKR_FLOW='0'), 'bulk', None)):
        calls = {'n': 0}

        def flaky(self, *a, **kw):
            out = orig(self, *a, **kw)
            if calls['n'] < failures and self.layout == 'flow':          # as if a spin of this epoch's launch had run out
                self.ctl[tkr_hip.FLOW_CTL_STATUS] = 1
                # ... in the planner prologue of the launch: its commit never ran and K1's touch bits of the call are still set
                # (restore() has to clear them, or the retry plans against wrong versions and buffer parities)
                self._cnt.touch_u[::3] = 0x5a5a
                self._cnt.touch_i[::2] = 0x0f0f
                calls['n'] += 1
            return out
        monkeypatch.setattr(_engine.BprEngine, 'run_batches', flaky)
        m = model()
        with pytest.warns(UserWarning, match='restarts from its initial state'):
            m.train(epochs=2, batch_size=64, epoch_sample_limit=64 * 30, seed=9, verbose=False)
        assert calls['n'] == failures and m._eng.layout == layout and (owners is None or m._eng.plan.owners == owners)
        assert m._eng.triplets_drawn == 2 * 30 * 64
        monkeypatch.setattr(_engine.BprEngine, 'run_batches', orig)
        for key, val in env.items():
            monkeypatch.setenv(key, val)
        clean = model()
        clean.train(epochs=2, batch_size=64, epoch_sample_limit=64 * 30, seed=9, verbose=False)
        for key in env:
            monkeypatch.delenv(key)
        assert clean._eng.layout == layout
        for a, b in ((m.fue, clean.fue), (m.fie, clean.fie), (m.fib, clean.fib)):
            assert np.array_equal(a, b)
        # (K2f adds a batch's loss up with LDS atomics, K2 with a sliced sum: the tables are bitwise, the reported loss to its last bits)
        assert abs(m.last_epoch_loss - clean.last_epoch_loss) <= 1e-5 * abs(clean.last_epoch_loss)


def test_call_ahead_plans_the_next_call_behind_this_one():
    """plain layout (batch > 512), TKR_CALL_AHEAD=1: the last chunk of a call plans the next call's first chunk on the side stream.
    Same tables, counters and stream position as without, whether the caller comes back for more, for less, or looks at the counters
    (settle() rolls the unused plan back) in between"""
    import numpy as np
    from oracle import plan_np as P
    from single import _engine
    from single._config import Tuning
    rng = np.random.Generator(np.random.PCG64(4))
    n_users, n_items, k, B = 3000, 700, 64, 2048
    tr = {int(u): [int(x) for x in rng.integers(0, n_items, int(rng.integers(1, 12)))] for u in range(n_users)}
    row_ptr, pos, srt = P.build_csr(tr, n_users)
    dev = torch.device('cuda')
    csr = _engine.TrainingCSR.from_arrays(row_ptr, pos, np.asarray(list(tr.keys()), np.int32), dev)
    hp = dict(lu=2.5e-3, li=2.5e-3, lj=2.5e-4, lb=1e-3, lr=0.02, mode='l2')
    out = {}
    for ahead in (False, True):
        cfg = Tuning.from_env({})
        cfg.call_ahead = ahead
        e = _engine.BprEngine(n_users, n_items, k, hp, dev, seed=9, tuning=cfg)
        e.run_batches(csr, 20, B, want_loss=False)
        assert (e._ahead is not None) == ahead
        e.run_batches(csr, 20, B, want_loss=False)          # comes back for the same
        e.run_batches(csr, 9, B, want_loss=False)           # ... for less than was planned
        ucnt = e.cnt.ucnt.clone()                           # looks at the counters: settle() drops what was planned and did not run
        e.run_batches(csr, 20, B, want_loss=False)
        e.check()
        out[ahead] = ([t.clone() for n in ('U', 'V', 'b') for t in e.get(n)], ucnt, e.cnt.icnt.clone(), e.triplets_drawn)
    for x, y in zip(out[False][0], out[True][0]):
        assert torch.equal(x, y)
    assert torch.equal(out[False][1], out[True][1]) and torch.equal(out[False][2], out[True][2])
    assert out[False][3] == out[True][3] == 69 * B
