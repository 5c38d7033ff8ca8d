"""GPU parity of K3 (VBPR step) against oracle/ref_np.vbpr_step through the C ABI, and the VBPR class
end to end (train -> folded export, vbpr.py:124-126).  fp32 tolerance stated per assert."""
import os
import pickle

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import plan_np as P
from oracle import ref_np as R


def _toy(n_users, n_items, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    tr = {int(u): [int(x) for x in rng.integers(0, n_items, int(rng.integers(1, 10)))] for u in rng.permutation(n_users)[: n_users - 5]}
    return tr, list(tr.keys())


@pytest.mark.parametrize('k,d,B,nb,mode,dense', [(16, 40, 64, 6, 'l2', True), (128, 700, 256, 4, 'l2', False),
                                                 (50, 333, 128, 3, 'l1', False), (128, 1030, 1024, 2, 'l2', False),
                                                 (200, 130, 96, 3, 'l2', True)])
@pytest.mark.parametrize('view', ['dense', 'sparse'])
def test_vbpr_step_parity(k, d, B, nb, mode, dense, view):
    """both implementations of the feature contraction: the dense MFMA kernels (V1/V3) and the CSR/CSC view (S1/S3)"""
    import tkr_hip
    from single import _engine
    n_users, n_items = 300, 90
    kh = k // 2
    tr, tr_users = _toy(n_users, n_items, seed=k + d)
    rng = np.random.Generator(np.random.PCG64(d))
    feat = np.abs(rng.standard_normal((n_items, d))).astype(np.float32)
    if not dense:
        feat *= rng.random((n_items, d)) < 0.1                       # tf-idf-like sparsity, stored dense
    feat /= np.maximum(np.linalg.norm(feat, axis=1, keepdims=True), 1e-6)
    feat = feat.astype(np.float32)
    hp = dict(lu=2.5e-3, li=2.5e-3, lj=2.5e-4, lb=1e-3, le=1e-3, lr=0.02, mode=mode)
    dev = torch.device('cuda')
    eng = _engine.VbprEngine(n_users, n_items, k, d, feat, hp, dev, seed=5, sparse=(view == 'sparse'))
    assert (eng.sparse is not None) == (view == 'sparse')
    # start from a non-degenerate cem / icb so every term of x is exercised
    eng.set_dense(cem=(rng.standard_normal((d, kh)) * 0.05).astype(np.float32), icb=(rng.standard_normal(d) * 0.05).astype(np.float32))
    eng.set_items(irb=(rng.standard_normal(n_items) * 0.01).astype(np.float32))
    U0 = eng.get('U')[0].cpu().numpy()
    ref = dict(ure=U0[:, :kh].copy(), uce=U0[:, kh:].copy(), ire=eng.get('I')[0].cpu().numpy(), irb=eng.get('irb')[0].cpu().numpy(),
               cem=eng.cem.cpu().numpy(), icb=eng.icb.cpu().numpy())
    for n in list(ref):
        ref['ms_' + n] = np.ones_like(ref[n])
    row_ptr, pos, srt = P.build_csr(tr, n_users)
    csr = _engine.TrainingCSR.from_arrays(row_ptr, pos, np.asarray(tr_users, np.int32), dev)
    loss = eng.run_batches(csr, nb, B).cpu().numpy()
    torch.cuda.synchronize()
    u, i, j = P.sample_triplets(tr_users, row_ptr, pos, srt, n_items, 5, 0, nb * B)
    np.testing.assert_array_equal(eng.plan.u.cpu().numpy()[: nb * B], u)
    ref_loss = [R.vbpr_step(ref, feat, u[b * B:(b + 1) * B], i[b * B:(b + 1) * B], j[b * B:(b + 1) * B], hp) for b in range(nb)]
    Uc, msU = (t.cpu().numpy() for t in eng.get('U'))
    tol = dict(rtol=3e-4, atol=2e-5)        # fp32; dense RMSProp moves every element by ~lr/sqrt(0.1) per batch
    np.testing.assert_allclose(Uc[:, :kh], ref['ure'], err_msg='ure', **tol)
    np.testing.assert_allclose(Uc[:, kh:], ref['uce'], err_msg='uce', **tol)
    np.testing.assert_allclose(msU[:, :kh], ref['ms_ure'], rtol=1e-3, atol=1e-6)
    np.testing.assert_allclose(eng.get('I')[0].cpu().numpy(), ref['ire'], err_msg='ire', **tol)
    np.testing.assert_allclose(eng.get('irb')[0].cpu().numpy(), ref['irb'], err_msg='irb', **tol)
    np.testing.assert_allclose(eng.cem.cpu().numpy(), ref['cem'], err_msg='cem', **tol)
    np.testing.assert_allclose(eng.icb.cpu().numpy(), ref['icb'], err_msg='icb', **tol)
    np.testing.assert_allclose(eng.mscem.cpu().numpy(), ref['ms_cem'], rtol=1e-3, atol=1e-6)
    np.testing.assert_allclose(loss, np.array(ref_loss), rtol=2e-4)


def test_vbpr_class_end_to_end(tmp_path):
    import synth
    from single import VBPR
    r = synth.make_ratings(200, 80, 20, seed=9, mu=3.0, sigma=0.5, min_r=5, max_r=40, om_per_user=3)
    data = str(tmp_path / 'data')
    synth.write_dataset(data, r)
    d, k = 64, 16
    feats = synth.make_content(100, d, nnz_per_row=12, seed=7)
    pickle.dump(feats, open(tmp_path / 'meta.pkl', 'wb'))
    m = VBPR(k=k, d=d, lambda_e=1e-3, lr=0.02)
    m.load_training_data(os.path.join(data, 'uid'), os.path.join(data, 'vid'), os.path.join(data, 'f0tr.txt'))
    m.load_content_data(str(tmp_path / 'meta.pkl'), os.path.join(data, 'vid'))
    assert m.feat.shape == (100, d)
    m.train(epochs=2, batch_size=64, epoch_sample_limit=64 * 20, seed=3, verbose=False)
    kh = k // 2
    assert m.fue.shape == (200, k) and m.fie.shape == (100, k) and m.fib.shape == (100, 1)
    # oracle on the same init (tables before training are reproducible from the seed) and stream
    from single import _engine
    eng0 = _engine.VbprEngine(m.n_users, m.n_items, k, d, m.feat, m._hyper(), torch.device('cuda'), seed=3)
    U0 = eng0.get('U')[0].cpu().numpy()
    ref = dict(ure=U0[:, :kh].copy(), uce=U0[:, kh:].copy(), ire=eng0.get('I')[0].cpu().numpy(), irb=np.zeros(100, np.float32),
               cem=eng0.cem.cpu().numpy(), icb=np.zeros(d, np.float32))
    for n in list(ref):
        ref['ms_' + n] = np.ones_like(ref[n])
    row_ptr, pos, srt = P.build_csr(m.tr_data, m.n_users)
    u, i, j = P.sample_triplets(m.tr_users, row_ptr, pos, srt, m.n_items, 3, 0, 2 * 20 * 64)
    for b in range(40):
        R.vbpr_step(ref, m.feat, u[b * 64:(b + 1) * 64], i[b * 64:(b + 1) * 64], j[b * 64:(b + 1) * 64], m._hyper())
    fue, fie, fib = R.vbpr_fold(ref, m.feat)
    np.testing.assert_allclose(m.fue, fue, rtol=3e-4, atol=2e-5)
    np.testing.assert_allclose(m.fie, fie, rtol=3e-4, atol=2e-5)
    np.testing.assert_allclose(m.fib, fib, rtol=3e-4, atol=2e-5)
    # export + resume path runs (weights restores the dense content tables)
    m.export_embeddings(str(tmp_path / 'vb'))
    m.train(epochs=1, batch_size=64, epoch_sample_limit=64 * 5, model_path=str(tmp_path / 'vb'), seed=4, verbose=False)
    assert np.isfinite(m.fie).all()


def test_vbpr_sparse_view_is_deterministic_and_auto_selected():
    """tf-idf-like features pick the sparse view on their own; two runs from the same state are bitwise identical
    (the column walk adds the batch's items in ascending item order, no float atomics on parameters)"""
    from single import _engine
    n_users, n_items, k, d, B, nb = 200, 150, 64, 2000, 256, 5
    tr, tr_users = _toy(n_users, n_items, seed=1)
    rng = np.random.Generator(np.random.PCG64(2))
    feat = (np.abs(rng.standard_normal((n_items, d))) * (rng.random((n_items, d)) < 0.01)).astype(np.float32)
    feat[7] = 0.0                                                    # an item without any feature
    hp = dict(lu=2.5e-3, li=2.5e-3, lj=2.5e-4, lb=1e-3, le=1e-3, lr=0.02, mode='l2')
    dev = torch.device('cuda')
    row_ptr, pos, srt = P.build_csr(tr, n_users)
    outs = []
    for _ in range(2):
        eng = _engine.VbprEngine(n_users, n_items, k, d, feat, hp, dev, seed=9)
        assert eng.sparse is not None and int(eng.sparse['f_ptr'][-1]) == int(np.count_nonzero(feat))
        csr = _engine.TrainingCSR.from_arrays(row_ptr, pos, np.asarray(tr_users, np.int32), dev)
        eng.run_batches(csr, nb, B, want_loss=False)
        torch.cuda.synchronize()
        outs.append([eng.cem.cpu().numpy(), eng.icb.cpu().numpy(), eng.get('U')[0].cpu().numpy(), eng.get('I')[0].cpu().numpy()])
    for a, b in zip(*outs):
        np.testing.assert_array_equal(a, b)
    assert np.abs(outs[0][0] - 2.0 / (d * k)).max() > 0            # cem moved
