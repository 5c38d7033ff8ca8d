"""-m gpu: utils.get_score / utils.evaluate through K4 + K6 + K7 against the reference's own outputs
(golden G9) and against the oracle restatement on exact-arithmetic inputs with deliberate ties.
Index work (raw ranks, hit counts): bit-exact.  Reciprocal-rank sums: fp64, |d| <= 1e-12 relative (the reference adds
the same terms in a different order)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import ref_np as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def hip():
    import tkr_hip
    tkr_hip.lib()
    return tkr_hip


def _likes(path):
    likes = {}
    for line in open(path):
        terms = line.strip().split(',')
        likes[terms[0]] = set(t.split(':')[0] for t in terms[1:] if t.split(':')[1] == '1')
    return likes


def test_g9_reference_outputs(hip, golden_dir):
    import utils
    exp = json.load(open(os.path.join(golden_dir, 'g9', 'expected.json')))
    data, model = os.path.join(golden_dir, 'g4', 'data'), os.path.join(golden_dir, 'g4', 'model')
    uids = utils.get_id_dict_from_file(os.path.join(data, 'uid'))
    vids = utils.get_id_dict_from_file(os.path.join(data, 'vid'))
    U = utils.get_embed_from_file(os.path.join(model, 'final-U.dat'), uids)
    V = utils.get_embed_from_file(os.path.join(model, 'final-V.dat'), vids)
    rated, counter = utils.get_history_from_file(os.path.join(data, 'f0tr.txt'))
    assert counter == exp['counter']
    for run in exp['runs']:
        sc = run['scenario']
        te_iids = utils.get_id_dict_from_file(os.path.join(data, 'f0te.%s.idl' % sc))
        te_ivt = utils.get_iv_dict_from_file(os.path.join(data, 'f0te.%s.idl' % sc))
        score = utils.get_score(U, V, vids, te_iids)
        assert score.shape == (len(uids), len(te_iids))
        if sc == 'im':                                                # the dense view, for callers that want it
            np.testing.assert_allclose(np.asarray(score), np.load(os.path.join(golden_dir, 'g9', 'score_im.npy')), rtol=2e-5, atol=1e-7)
        hits, trrs, count = utils.evaluate(score, rated, _likes(os.path.join(data, 'f0te.%s.txt' % sc)), uids, te_iids, te_ivt,
                                           run['step'], run['total'], run['total'] // run['step'])
        assert hits == run['hits'] and count == run['count']
        np.testing.assert_allclose(trrs, run['trrs'], rtol=1e-12)
    with pytest.raises(TypeError):
        utils.evaluate(np.zeros((2, 2), np.float32), {}, {}, {}, {}, {}, 5, 30, 6)


@pytest.mark.parametrize('n_users,n_te,k,step,total', [(150, 90, 8, 5, 30), (64, 400, 16, 7, 64), (40, 700, 4, 50, 300),
                                                        (30, 20, 8, 3, 30)])
def test_exact_arithmetic_with_ties_matches_oracle(hip, n_users, n_te, k, step, total):
    """small-integer factors: every score is exact in fp32 and ties are frequent; canonical order on both sides.
    Covers total > 32 (several K4 launches), total > 256 (several K6 launches) and total > unrated columns."""
    import utils
    rng = np.random.Generator(np.random.PCG64(n_users + n_te))
    U = rng.integers(-3, 4, (n_users, k)).astype(np.float32) / 2
    V = rng.integers(-3, 4, (n_te + 5, k)).astype(np.float32) / 2
    uids = {'u%d' % x: x for x in range(n_users)}
    iids = {'i%d' % x: x for x in range(n_te + 5)}
    order = rng.permutation(n_te + 5)[:n_te]
    te_iids = {'i%d' % x: c for c, x in enumerate(order)}
    te_ivt = {c: t for t, c in te_iids.items()}
    rated, likes = {}, {}
    for uid in uids:
        rated[uid] = set('i%d' % x for x in rng.choice(n_te + 5, int(rng.integers(0, n_te // 2)), replace=False))
        likes[uid] = set('i%d' % x for x in rng.choice(n_te + 5, int(rng.integers(0, 12)), replace=False))
    interval = total // step
    score = utils.get_score(U, V, iids, te_iids)
    hits, trrs, count = utils.evaluate(score, rated, likes, uids, te_iids, te_ivt, step, total, interval)
    ref_score = R.utils_get_score(U, V, iids, te_iids)
    np.testing.assert_array_equal(np.asarray(score), ref_score)       # exact arithmetic
    eh, er, ec = R.utils_evaluate(ref_score, rated, likes, uids, te_iids, te_ivt, step, total, interval, canonical=True)
    assert hits == eh and count == ec
    np.testing.assert_allclose(trrs, er, rtol=1e-12)


def test_raw_ranks_direct(hip):
    """K6 alone: raw rank = position among ALL columns for kept columns, -1 padding untouched, bias honoured"""
    rng = np.random.Generator(np.random.PCG64(3))
    n_rows, n_cols, k, K = 37, 130, 200, 24
    U = rng.integers(-4, 5, (n_rows, k)).astype(np.float32) / 4
    V = rng.integers(-4, 5, (n_cols, k)).astype(np.float32) / 4
    b = rng.integers(-8, 9, n_cols).astype(np.float32) / 8
    s = U @ V.T + b
    rated = [np.sort(rng.choice(n_cols, int(rng.integers(0, 110)), replace=False)) for _ in range(n_rows)]
    ptr = np.zeros(n_rows + 1, np.int64)
    np.cumsum([len(x) for x in rated], out=ptr[1:])
    ids = np.full((n_rows, K), -1, np.int32)
    want = np.full((n_rows, K), -1, np.int32)
    for r in range(n_rows):
        order = np.argsort(s[r], kind='stable')[::-1]
        pos = {int(c): t for t, c in enumerate(order)}
        kept = [int(c) for c in order if c not in set(rated[r].tolist())][:K]
        ids[r, :len(kept)] = kept
        want[r, :len(kept)] = [pos[c] for c in kept]
    dev = torch.device('cuda')
    got = hip.raw_ranks(torch.from_numpy(U).to(dev), torch.from_numpy(V).to(dev), torch.from_numpy(ids).to(dev),
                        torch.from_numpy(ptr).to(dev), torch.from_numpy(np.concatenate(rated).astype(np.int32)).to(dev),
                        bias=torch.from_numpy(b).to(dev))
    np.testing.assert_array_equal(got.cpu().numpy(), want)
