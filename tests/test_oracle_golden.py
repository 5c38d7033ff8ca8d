"""Pin the CPU oracle (oracle/ref_np.py) to the golden vectors captured from the
reference's own code (tests/golden/make_golden.py): G1 loader, G2 legacy sampler,
G3 text I/O, G4-G7 evaluate CLI (stdout + per-user filtered lists)."""
import json
import os

import numpy as np
import pytest

from oracle import ref_np as R


def test_g1_loader(golden_dir):
    d = os.path.join(golden_dir, 'g1')
    exp = json.load(open(os.path.join(d, 'expected.json')))
    got = R.load_training(os.path.join(d, 'uid'), os.path.join(d, 'vid'), os.path.join(d, 'tr.txt'))
    assert got['uids'] == exp['uids'] and got['iids'] == exp['iids']
    assert [list(p) for p in got['data']] == exp['data']
    assert got['n_users'] == exp['n_users'] and got['n_items'] == exp['n_items']
    assert got['epoch_sample_limit'] == exp['epoch_sample_limit']
    assert {str(k): list(v) for k, v in got['tr_data'].items()} == exp['tr_data']
    assert list(got['tr_data'].keys()) == [int(k) for k in exp['tr_data'].keys()]   # insertion order
    assert got['tr_users'] == exp['tr_users']
    # the duplicate-id quirk (utils.py:10-16): u2 re-pointed, u5 shares its index
    assert exp['uids']['u2'] == exp['uids']['u5'] == 4 and exp['n_users'] == 5


def test_g2_legacy_sampler_stream(golden_dir):
    d = os.path.join(golden_dir, 'g2')
    exp = np.load(os.path.join(d, 'expected.npz'))
    m = R.load_training(os.path.join(d, 'uid'), os.path.join(d, 'vid'), os.path.join(d, 'f0tr.txt'))
    np.random.seed(123)
    gen = R.legacy_uniform_user_sampler(m['tr_users'], m['tr_data'], m['n_items'], 16)
    for b in range(exp['ub'].shape[0]):
        ub, ib, jb = next(gen)
        assert str(ub.dtype) == str(exp['ub_dtype']) and str(ib.dtype) == str(exp['ib_dtype'])
        np.testing.assert_array_equal(ub, exp['ub'][b])
        np.testing.assert_array_equal(ib, exp['ib'][b])
        np.testing.assert_array_equal(jb, exp['jb'][b])


def test_g3_text_io(golden_dir, tmp_path):
    d = os.path.join(golden_dir, 'g3')
    exp = np.load(os.path.join(d, 'expected.npz'))
    ids = json.load(open(os.path.join(d, 'ids.json')))
    R.write_embed_text(str(tmp_path / 'm' / 'mat.dat'), exp['mat'])      # also creates the parent dir
    R.write_embed_text(str(tmp_path / 'm' / 'bias.dat'), exp['bias'])
    for name in ('mat.dat', 'bias.dat'):
        assert open(tmp_path / 'm' / name, 'rb').read() == open(os.path.join(d, name), 'rb').read()
    np.testing.assert_array_equal(R.read_embed_text(os.path.join(d, 'mat.dat')), exp['back_all'])
    np.testing.assert_array_equal(R.read_embed_text(os.path.join(d, 'mat.dat'), ids), exp['back_ids'])
    np.testing.assert_array_equal(R.read_embed_text(os.path.join(d, 'bias.dat'), ids), exp['back_bias'])
    assert R.read_embed_text(os.path.join(d, 'missing.dat')) is None


def _lists(data, model, sc, total, step=5):
    uids = R.read_id_list(os.path.join(data, 'uid'))
    vids = R.read_id_list(os.path.join(data, 'vid'))
    rated = R.read_history(os.path.join(data, 'f0tr.txt'))
    U = R.read_embed_text(os.path.join(model, 'final-U.dat'), uids)
    V = R.read_embed_text(os.path.join(model, 'final-V.dat'), vids)
    bp = os.path.join(model, 'final-B.dat')
    b = R.read_embed_text(bp, vids) if os.path.exists(bp) else None
    idl = os.path.join(data, 'f0te.%s.idl' % sc)
    teids, teivt = R.read_id_list(idl), R.read_inverse_id_list(idl)
    tests = R.read_test_likes(os.path.join(data, 'f0te.%s.txt' % sc), teids)
    return R.evaluate_scenario(U, V, b, uids, vids, rated, teids, teivt, tests, step, total,
                               canonical=True, return_lists=True)[1]


@pytest.mark.parametrize('g,scs', [('g4', ['im', 'om']), ('g5', ['im', 'om']), ('g6', ['all'])])
def test_g456_cli_and_lists(golden_dir, g, scs):
    d = os.path.join(golden_dir, g)
    exp = json.load(open(os.path.join(d, 'expected.json')))
    data, model = os.path.join(d, 'data'), os.path.join(d, 'model')
    assert R.evaluate_cli(data, model, scenarios=scs) == exp['stdout']
    assert R.evaluate_cli(data, model, scenarios=scs, canonical=False) == exp['stdout']
    if 'stdout_s3_t10' in exp:
        assert R.evaluate_cli(data, model, step=3, total=10, scenarios=['om', 'im']) == exp['stdout_s3_t10']
    for sc in scs:
        assert exp['tie_free'][sc]
        assert _lists(data, model, sc, 30) == exp['lists'][sc]


def test_g7_edge_cases(golden_dir):
    d = os.path.join(golden_dir, 'g7')
    exp = json.load(open(os.path.join(d, 'expected.json')))
    data, model = os.path.join(d, 'data'), os.path.join(d, 'model')
    for run in exp['runs']:
        assert R.evaluate_cli(data, model, step=run['step'], total=run['total'], scenarios=['sm']) == run['stdout']
        assert _lists(data, model, 'sm', run['total'], run['step']) == run['lists']


def test_batches_per_epoch_f7():
    assert R.batches_per_epoch(10**6, 256) == 3906
    assert R.batches_per_epoch(10e5, 256) == 3906
    assert R.batches_per_epoch(512, 256) == 2


def test_g8_legacy_pregenerated_sampler(golden_dir):
    """old/methods/bpr.py:88-99 + :101-105, bit-exact under np.random.seed(321)"""
    from collections import defaultdict
    exp = json.load(open(os.path.join(golden_dir, 'g8', 'expected.json')))
    users = {int(k): v for k, v in exp['users'].items()}
    items = {int(k): v for k, v in exp['items'].items()}
    train_dict = defaultdict(list)
    for uid, iid in exp['data']:
        train_dict[users[uid]].append(items[iid])
    assert {str(k): v for k, v in train_dict.items()} == exp['train_dict']
    assert list(train_dict.keys()) == [int(k) for k in exp['train_dict'].keys()]        # insertion order feeds keys()
    np.random.seed(321)
    su, sp, sn = R.legacy_pregenerated_sampler(train_dict, len(items), 64)
    assert su.tolist() == exp['su'] and sp.tolist() == exp['sp'] and sn.tolist() == exp['sn']
    for u, i, j in zip(su, sp, sn):
        assert i in train_dict[u] and j not in train_dict[u]
    # old/methods/bpr.py:72: while (z+1)*B < n  -- the batch that would end exactly at n is dropped too
    assert [R.legacy_batches(n, 4) for n in (0, 1, 4, 5, 8, 9)] == [0, 0, 0, 1, 1, 2]


def test_g9_utils_evaluate(golden_dir):
    """utils.py get_history_from_file / get_score / evaluate restated, pinned on the G4 data"""
    exp = json.load(open(os.path.join(golden_dir, 'g9', 'expected.json')))
    data, model = os.path.join(golden_dir, 'g4', 'data'), os.path.join(golden_dir, 'g4', 'model')
    uids, vids = R.read_id_list(os.path.join(data, 'uid')), R.read_id_list(os.path.join(data, 'vid'))
    U = R.read_embed_text(os.path.join(model, 'final-U.dat'), uids)
    V = R.read_embed_text(os.path.join(model, 'final-V.dat'), vids)
    rated, counter = R.read_history_counts(os.path.join(data, 'f0tr.txt'))
    assert counter == exp['counter']
    assert R.read_history_counts(os.path.join(data, 'nope.txt')) == ({}, {})
    for run in exp['runs']:
        sc = run['scenario']
        te_iids = R.read_id_list(os.path.join(data, 'f0te.%s.idl' % sc))
        te_ivt = R.read_iv_list(os.path.join(data, 'f0te.%s.idl' % sc))
        assert {v: k for k, v in te_iids.items()} == te_ivt
        likes = {}
        for line in open(os.path.join(data, 'f0te.%s.txt' % sc)):
            terms = line.strip().split(',')
            likes[terms[0]] = set(t.split(':')[0] for t in terms[1:] if t.split(':')[1] == '1')
        score = R.utils_get_score(U, V, vids, te_iids)
        if sc == 'im':
            np.testing.assert_array_equal(score, np.load(os.path.join(golden_dir, 'g9', 'score_im.npy')))
        for canonical in (False, True):                              # G4 is tie-free: both orders agree with the reference
            hits, trrs, count = R.utils_evaluate(score, rated, likes, uids, te_iids, te_ivt, run['step'], run['total'],
                                                 run['total'] // run['step'], canonical=canonical)
            assert hits == run['hits'] and count == run['count']
            assert trrs == run['trrs']                               # same additions in the same order


def test_mfma_chain_scores_is_an_fp32_dot_product():
    """oracle/ref_np.mfma_chain_scores (the summation order the K4 kernels are held to): exact where every partial sum is
    representable, within the fp32 dot-product error bound of float64 otherwise, bias added last, -0.0 canonicalised"""
    rng = np.random.Generator(np.random.PCG64(17))
    for k in (1, 7, 50, 128):
        U = rng.integers(-8, 9, (6, k)).astype(np.float32) / 8
        V = rng.integers(-8, 9, (9, k)).astype(np.float32) / 8
        b = rng.integers(-8, 9, 9).astype(np.float32) / 8
        np.testing.assert_array_equal(R.mfma_chain_scores(U, V, b), U @ V.T + b)
        U = (rng.standard_normal((6, k)) * 0.01).astype(np.float32)
        V = (rng.standard_normal((9, k)) * 0.01).astype(np.float32)
        got = R.mfma_chain_scores(U, V, None)
        s64 = U.astype(np.float64) @ V.astype(np.float64).T
        bound = k * 2.0 ** -24 * (np.abs(U).astype(np.float64) @ np.abs(V).astype(np.float64).T)
        assert got.dtype == np.float32 and np.all(np.abs(got - s64) <= bound + 1e-45)
    z = R.mfma_chain_scores(np.zeros((1, 4), np.float32), -np.ones((1, 4), np.float32), None)
    assert not np.signbit(z).any()
