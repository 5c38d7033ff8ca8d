"""Generate the golden fixtures G1-G9 by running the REFERENCE's own importable code.

Run in the build container only (``/root/reference`` does not exist on the GPU box):

    python tests/golden/make_golden.py

What runs from the reference (never copied into this repo, only executed):
  * ``utils.py``            get_id_dict_from_file / get_data_from_file / export_embed_to_file /
                            get_embed_from_file                                    (G1, G3)
  * ``single/bpr.py``       BPR.load_training_data / _uniform_user_sampling with the
                            ``tensorflow`` module stubbed (those methods are pure numpy) (G1, G2)
  * ``evaluate.py``         the CLI itself through a subprocess (stdout is the fixture) and
                            its helper functions get_ids/get_ivt/get_mat/get_history for the
                            per-user list extraction                                (G4-G7)
  * ``old/methods/bpr.py``  _data_to_dict / _uniform_user_sampling with ``theano`` stubbed       (G8)
  * ``utils.py``            get_history_from_file / get_score / evaluate                      (G9)
The TF train step cannot run (TensorFlow 1.15 absent): no fixture pins it.
Inputs are produced by ``top-k-rec_amd/synth.py`` (this repo) and committed next to the
expected outputs, so the tests never need the generator to stay frozen.
"""
import json
import os
import shutil
import subprocess
import sys
from unittest import mock

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, os.path.join(ROOT, 'top-k-rec_amd'))
import synth  # noqa: E402

for name in ('tensorflow', 'tensorflow.compat', 'tensorflow.compat.v1'):
    sys.modules[name] = mock.MagicMock()
sys.path.insert(0, REF)
import utils as ref_utils          # noqa: E402  (reference)
import evaluate as ref_eval        # noqa: E402  (reference)
from single.bpr import BPR as RefBPR   # noqa: E402  (reference, TF stubbed)


def fresh(path):
    shutil.rmtree(path, ignore_errors=True)
    os.makedirs(path)
    return path


def run_cli(data, model, scenarios, step=5, total=30):
    out = subprocess.run([sys.executable, os.path.join(REF, 'evaluate.py'), '-d', data, '-m', model,
                          '-s', str(step), '-t', str(total), '-sl'] + list(scenarios),
                         capture_output=True, text=True, check=True)
    return out.stdout.strip().split('\n')


def ref_lists(data, model, scenario, total, fold=0, with_bias=False):
    """Per-user filtered top lists, built with the reference's helpers and the same
    np.dot / np.argsort(default kind) calls as evaluate.py:78-81,96-105."""
    uids = ref_eval.get_ids(os.path.join(data, 'uid'))
    vids = ref_eval.get_ids(os.path.join(data, 'vid'))
    rated, _ = ref_eval.get_history(os.path.join(data, 'f%dtr.txt' % fold))
    umat = ref_eval.get_mat(os.path.join(model, 'final-U.dat'), uids)
    vmat = ref_eval.get_mat(os.path.join(model, 'final-V.dat'), vids)
    idl = os.path.join(data, 'f%dte.%s.idl' % (fold, scenario))
    teids, teivt = ref_eval.get_ids(idl), ref_eval.get_ivt(idl)
    temat = np.zeros((len(teids), vmat.shape[1]), dtype=np.float32)
    for vid in teids:
        temat[teids[vid], :] = vmat[vids[vid], :]
    scores = np.dot(umat, temat.T)
    if with_bias:
        scores += ref_eval.get_mat(os.path.join(model, 'final-B.dat'), vids).reshape((1, -1))
    rlist = np.argsort(scores, axis=1)
    stable = np.argsort(scores, axis=1, kind='stable')
    lists, tie_free = {}, True
    for line in open(os.path.join(data, 'f%dte.%s.txt' % (fold, scenario))):
        terms = line.strip().split(',')
        uid = terms[0]
        if not any(int(t.split(':')[1]) == 1 for t in terms[1:]):
            continue
        for order, store in ((rlist, True), (stable, False)):
            kept = []
            for t in range(len(teids)):
                c = int(order[uids[uid], len(teids) - 1 - t])
                if teivt[c] not in rated[uid]:
                    kept.append(c)
                if len(kept) == total:
                    break
            if store:
                lists[uid] = kept
            elif kept != lists[uid]:
                tie_free = False
    return lists, tie_free


def write_model(model_dir, U, V, b=None):
    fresh(model_dir)
    ref_utils.export_embed_to_file(os.path.join(model_dir, 'final-U.dat'), U)
    ref_utils.export_embed_to_file(os.path.join(model_dir, 'final-V.dat'), V)
    if b is not None:
        ref_utils.export_embed_to_file(os.path.join(model_dir, 'final-B.dat'), b)


# ---------------------------------------------------------------- G1 loader
def g1():
    d = fresh(os.path.join(HERE, 'g1'))
    open(os.path.join(d, 'uid'), 'w').write('u1\nu2\nu3\nu4\nu2\nu5\n')       # duplicate id line (utils.py:10-16 quirk)
    open(os.path.join(d, 'vid'), 'w').write('a\nb\nc\nd\ne\nf\n')
    open(os.path.join(d, 'tr.txt'), 'w').write(
        'u3,a:1,b:0,c:1,a:1\n'            # duplicate positive kept
        'u1,d:1,zz:1,e:0\n'               # unknown item skipped
        'u9,a:1\n'                        # unknown user skipped
        'u4\n'                            # no items
        'u2,f:1,b:1\n'
        'u5,c:0\n'                        # only dislikes -> not in tr_users
        'u3,e:1\n')                       # second line for the same user appends
    m = RefBPR(k=4)
    m.load_training_data(os.path.join(d, 'uid'), os.path.join(d, 'vid'), os.path.join(d, 'tr.txt'), data_copy=True)
    json.dump(dict(uids=m.uids, iids=m.iids, data=[list(p) for p in m.data], n_users=m.n_users,
                   n_items=m.n_items, epoch_sample_limit=m.epoch_sample_limit,
                   tr_data={str(k): [int(x) for x in v] for k, v in m.tr_data.items()},
                   tr_users=[int(x) for x in m.tr_users]), open(os.path.join(d, 'expected.json'), 'w'), indent=1)
    return m


# ---------------------------------------------------------------- G2 sampler
def g2():
    d = fresh(os.path.join(HERE, 'g2'))
    r = synth.make_ratings(40, 30, 0, seed=5, mu=2.0, sigma=0.5, min_r=3, max_r=12)
    synth.write_dataset(d, r)
    m = RefBPR(k=4)
    m.load_training_data(os.path.join(d, 'uid'), os.path.join(d, 'vid'), os.path.join(d, 'f0tr.txt'))
    np.random.seed(123)
    gen = m._uniform_user_sampling(16)
    ub, ib, jb = [], [], []
    for _ in range(5):
        u, i, j = next(gen)
        ub.append(np.array(u)); ib.append(np.array(i)); jb.append(np.array(j))
    np.savez(os.path.join(d, 'expected.npz'), ub=np.stack(ub), ib=np.stack(ib), jb=np.stack(jb),
             ub_dtype=str(ub[0].dtype), ib_dtype=str(ib[0].dtype))


# ---------------------------------------------------------------- G3 text I/O
def g3():
    d = fresh(os.path.join(HERE, 'g3'))
    rng = np.random.Generator(np.random.PCG64(3))
    mat = (rng.standard_normal((7, 5)) * 0.01).astype(np.float32)
    mat[0, 0] = -0.0; mat[1, 1] = 0.0000004; mat[2, 2] = 0.0000005; mat[3, 3] = -1234.5678; mat[4, 4] = 1e-8
    bias = (rng.standard_normal((7, 1)) * 0.1).astype(np.float32)
    ref_utils.export_embed_to_file(os.path.join(d, 'mat.dat'), mat)
    ref_utils.export_embed_to_file(os.path.join(d, 'bias.dat'), bias)
    ids = {('id%d' % k): k for k in (3, 0, 6, 1, 2, 5, 4)}
    np.savez(os.path.join(d, 'expected.npz'), mat=mat, bias=bias,
             back_all=ref_utils.get_embed_from_file(os.path.join(d, 'mat.dat')),
             back_ids=ref_utils.get_embed_from_file(os.path.join(d, 'mat.dat'), ids),
             back_bias=ref_utils.get_embed_from_file(os.path.join(d, 'bias.dat'), ids))
    json.dump(ids, open(os.path.join(d, 'ids.json'), 'w'))


# ---------------------------------------------------------------- G4 evaluate CLI, medium
def g4():
    d = fresh(os.path.join(HERE, 'g4'))
    data, model = os.path.join(d, 'data'), os.path.join(d, 'model')
    r = synth.make_ratings(160, 96, 24, seed=11, mu=2.9, sigma=0.5, min_r=5, max_r=40, om_per_user=4)
    vid_order = np.random.Generator(np.random.PCG64(1)).permutation(120)      # vid order != idl order
    synth.write_dataset(data, r, vid_order=vid_order)
    rng = np.random.Generator(np.random.PCG64(12))
    # planted factors + noise: accuracy well above chance, no bias file (reference F5 crash)
    U = (0.25 * rng.standard_normal((160, 8))).astype(np.float32)
    V = (0.25 * rng.standard_normal((120, 8))).astype(np.float32)
    write_model(model, U, V)
    out = dict(stdout=run_cli(data, model, ['im', 'om']), lists={}, tie_free={})
    out['stdout_s3_t10'] = run_cli(data, model, ['om', 'im'], step=3, total=10)
    for sc in ('im', 'om'):
        out['lists'][sc], out['tie_free'][sc] = ref_lists(data, model, sc, 30)
    json.dump(out, open(os.path.join(d, 'expected.json'), 'w'))


# ---------------------------------------------------------------- G5 exact arithmetic
def g5():
    d = fresh(os.path.join(HERE, 'g5'))
    data, model = os.path.join(d, 'data'), os.path.join(d, 'model')
    r = synth.make_ratings(96, 80, 16, seed=21, mu=2.7, sigma=0.4, min_r=5, max_r=30, om_per_user=3)
    synth.write_dataset(data, r)
    rng = np.random.Generator(np.random.PCG64(22))
    k = 8
    V = rng.integers(-512, 513, (96, k)).astype(np.float32) / 64.0
    U = rng.integers(-512, 513, (96, k)).astype(np.float32) / 64.0
    for _ in range(200):                               # re-draw user rows that have an exact score tie
        s = U.astype(np.float64) @ V.astype(np.float64).T
        bad = [u for u in range(96) if len(np.unique(s[u])) != 96]
        if not bad:
            break
        U[bad] = rng.integers(-512, 513, (len(bad), k)).astype(np.float32) / 64.0
    assert not bad
    write_model(model, U, V)
    out = dict(stdout=run_cli(data, model, ['im', 'om']), lists={}, tie_free={})
    for sc in ('im', 'om'):
        out['lists'][sc], out['tie_free'][sc] = ref_lists(data, model, sc, 30)
        assert out['tie_free'][sc]
    json.dump(out, open(os.path.join(d, 'expected.json'), 'w'))


# ---------------------------------------------------------------- G6 bias path (idl == vid)
def g6():
    d = fresh(os.path.join(HERE, 'g6'))
    data, model = os.path.join(d, 'data'), os.path.join(d, 'model')
    r = synth.make_ratings(64, 48, 0, seed=31, mu=2.5, sigma=0.4, min_r=5, max_r=20)
    synth.write_dataset(data, r)
    shutil.copy(os.path.join(data, 'vid'), os.path.join(data, 'f0te.all.idl'))
    shutil.copy(os.path.join(data, 'f0te.im.txt'), os.path.join(data, 'f0te.all.txt'))
    rng = np.random.Generator(np.random.PCG64(32))
    U = rng.integers(-256, 257, (64, 8)).astype(np.float32) / 64.0
    V = rng.integers(-256, 257, (48, 8)).astype(np.float32) / 64.0
    b = rng.integers(-4096, 4097, (48, 1)).astype(np.float32) / 64.0 + np.arange(48, dtype=np.float32)[:, None] / 4096.0
    write_model(model, U, V, b)
    out = dict(stdout=run_cli(data, model, ['all']), lists={}, tie_free={})
    out['lists']['all'], out['tie_free']['all'] = ref_lists(data, model, 'all', 30, with_bias=True)
    json.dump(out, open(os.path.join(d, 'expected.json'), 'w'))


# ---------------------------------------------------------------- G7 edge cases
def g7():
    d = fresh(os.path.join(HERE, 'g7'))
    data, model = os.path.join(d, 'data'), os.path.join(d, 'model')
    os.makedirs(data)
    users = ['u%d' % x for x in range(6)]
    items = ['v%02d' % x for x in range(12)]
    open(os.path.join(data, 'uid'), 'w').write(''.join(x + '\n' for x in users))
    open(os.path.join(data, 'vid'), 'w').write(''.join(x + '\n' for x in items))
    open(os.path.join(data, 'f0te.sm.idl'), 'w').write(''.join(x + '\n' for x in items[2:12]))
    open(os.path.join(data, 'f0tr.txt'), 'w').write(
        'u0,v02:1,v03:0,v04:1,v05:1,v06:0,v07:1,v08:1,v09:0\n'   # only 2 unrated test columns left
        'u1,v00:1,v01:1\n'                                        # rated items outside the scenario
        'u2,v05:1,v11:0\n'
        'u3,v02:1\n'
        'u4,v03:0\n'
        'u5,v04:1,v10:1\n')
    open(os.path.join(data, 'f0te.sm.txt'), 'w').write(
        'u0,v10:1,v11:1,v02:1\n'      # v02 liked in test but train-rated: counted in |likes|, never a hit
        'u1,v02:1,v03:0,v09:1\n'
        'u2,v05:1\n'                  # all test likes are train-rated
        'u3,v04:0\n'                  # no likes -> skipped entirely
        'u5,v11:1,v06:1,v07:1\n')
    rng = np.random.Generator(np.random.PCG64(71))
    U = rng.integers(-64, 65, (6, 4)).astype(np.float32) / 64.0
    V = (rng.integers(-64, 65, (12, 4)).astype(np.float32) + np.arange(12, dtype=np.float32)[:, None] / 16.0) / 64.0
    write_model(model, U, V)
    out = dict(runs=[])
    for step, total in ((5, 30), (5, 12), (3, 7), (1, 4)):
        lists, tf = ref_lists(data, model, 'sm', total)
        out['runs'].append(dict(step=step, total=total, stdout=run_cli(data, model, ['sm'], step, total),
                                lists=lists, tie_free=tf))
    json.dump(out, open(os.path.join(d, 'expected.json'), 'w'))


# ---------------------------------------------------------------- G8 legacy pre-generated sampler
def g8():
    """old/methods/bpr.py:88-99 run with ``theano`` stubbed (the sampler and _data_to_dict are pure numpy)."""
    import importlib.util
    for name in ('theano', 'theano.tensor'):
        sys.modules[name] = mock.MagicMock()
    spec = importlib.util.spec_from_file_location('legacy_bpr', os.path.join(REF, 'old', 'methods', 'bpr.py'))
    legacy = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(legacy)
    d = fresh(os.path.join(HERE, 'g8'))
    users = {100 + 7 * x: x for x in range(12)}                  # raw id -> index, as bpr_train.py:23-40 builds them
    items = {5000 + 3 * x: x for x in range(20)}
    rng = np.random.Generator(np.random.PCG64(81))
    data = []
    for uid in users:
        if uid == 100 + 7 * 5:
            continue                                             # a user without positives never enters train_dict
        for iid in rng.choice(list(items), int(rng.integers(1, 9)), replace=False):
            data.append((int(uid), int(iid)))
    m = legacy.BPR(4, users, items)
    m._train_dict = m._data_to_dict(data, users, items)
    np.random.seed(321)
    su, sp, sn = m._uniform_user_sampling(64)
    json.dump(dict(users={str(k): v for k, v in users.items()}, items={str(k): v for k, v in items.items()}, data=data,
                   train_dict={str(k): [int(x) for x in v] for k, v in m._train_dict.items()},
                   su=[int(x) for x in su], sp=[int(x) for x in sp], sn=[int(x) for x in sn]),
              open(os.path.join(d, 'expected.json'), 'w'))


# ---------------------------------------------------------------- G9 utils.get_score / utils.evaluate
def g9():
    """utils.py:92-127 on the G4 data and model: dense score matrix, raw-rank buckets, reciprocal-rank sums."""
    d = fresh(os.path.join(HERE, 'g9'))
    src = os.path.join(HERE, 'g4')
    data, model = os.path.join(src, 'data'), os.path.join(src, 'model')
    uids = ref_utils.get_id_dict_from_file(os.path.join(data, 'uid'))
    vids = ref_utils.get_id_dict_from_file(os.path.join(data, 'vid'))
    U = ref_utils.get_embed_from_file(os.path.join(model, 'final-U.dat'), uids)
    V = ref_utils.get_embed_from_file(os.path.join(model, 'final-V.dat'), vids)
    rated, counter = ref_utils.get_history_from_file(os.path.join(data, 'f0tr.txt'))
    out = dict(counter=counter, runs=[])
    for sc in ('im', 'om'):
        te_iids = ref_utils.get_id_dict_from_file(os.path.join(data, 'f0te.%s.idl' % sc))
        te_ivt = {v: k for k, v in te_iids.items()}
        likes = {}
        for line in open(os.path.join(data, 'f0te.%s.txt' % sc)):
            terms = line.strip().split(',')
            likes[terms[0]] = set(t.split(':')[0] for t in terms[1:] if t.split(':')[1] == '1')
        score = ref_utils.get_score(U, V, vids, te_iids)
        for step, total in ((5, 30), (3, 10)):
            interval = total // step
            hits, trrs, count = ref_utils.evaluate(score, rated, likes, uids, te_iids, te_ivt, step, total, interval)
            out['runs'].append(dict(scenario=sc, step=step, total=total, hits=hits, trrs=trrs, count=count))
        if sc == 'im':
            np.save(os.path.join(d, 'score_im.npy'), score)
    json.dump(out, open(os.path.join(d, 'expected.json'), 'w'))


if __name__ == '__main__':
    todo = sys.argv[1:] or ['g1', 'g2', 'g3', 'g4', 'g5', 'g6', 'g7', 'g8', 'g9']
    for name in todo:
        globals()[name]()
    print('golden fixtures written under', HERE, todo)
