"""CPU tests of the host side: the reference-API mirror (utils / single.REC / single.BPR loaders and
text formats) against the golden vectors, the C-ABI library (loads, exports every symbol declared in
include/tkr.h -- no compute without a GPU), loud failure without a GPU, and the world-size-2 gloo test
of the multi-GPU exchange rule."""
import ctypes
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_utils_match_golden(golden_dir, tmp_path):
    import utils
    d = os.path.join(golden_dir, 'g1')
    exp = json.load(open(os.path.join(d, 'expected.json')))
    uids = utils.get_id_dict_from_file(os.path.join(d, 'uid'))
    iids = utils.get_id_dict_from_file(os.path.join(d, 'vid'))
    assert uids == exp['uids'] and iids == exp['iids']
    assert utils.get_id_dict_from_file(os.path.join(d, 'nope')) == {}
    data = utils.get_data_from_file(os.path.join(d, 'tr.txt'), uids, iids)
    assert [list(p) for p in data] == exp['data']
    g3 = os.path.join(golden_dir, 'g3')
    e3 = np.load(os.path.join(g3, 'expected.npz'))
    ids = json.load(open(os.path.join(g3, 'ids.json')))
    utils.export_embed_to_file(str(tmp_path / 'new' / 'mat.dat'), e3['mat'])
    utils.export_embed_to_file(str(tmp_path / 'new' / 'bias.dat'), e3['bias'])
    for name in ('mat.dat', 'bias.dat'):
        assert open(tmp_path / 'new' / name, 'rb').read() == open(os.path.join(g3, name), 'rb').read()
    np.testing.assert_array_equal(utils.get_embed_from_file(os.path.join(g3, 'mat.dat')), e3['back_all'])
    np.testing.assert_array_equal(utils.get_embed_from_file(os.path.join(g3, 'mat.dat'), ids), e3['back_ids'])
    np.testing.assert_array_equal(utils.get_embed_from_file(os.path.join(g3, 'bias.dat'), ids), e3['back_bias'])
    assert utils.get_embed_from_file(os.path.join(g3, 'missing')) is None


def test_bpr_loader_matches_reference(golden_dir):
    from single import BPR, REC, VBPR
    d = os.path.join(golden_dir, 'g1')
    exp = json.load(open(os.path.join(d, 'expected.json')))
    m = BPR(k=4)
    assert isinstance(m, REC) and (m.lu, m.li, m.lj, m.lb, m.lr, m.mode) == (2.5e-3, 2.5e-3, 2.5e-4, 0, 1.0e-4, 'l2')
    m.load_training_data(os.path.join(d, 'uid'), os.path.join(d, 'vid'), os.path.join(d, 'tr.txt'), data_copy=True)
    assert m.uids == exp['uids'] and m.iids == exp['iids']
    assert [list(p) for p in m.data] == exp['data']
    assert (m.n_users, m.n_items, m.epoch_sample_limit) == (exp['n_users'], exp['n_items'], exp['epoch_sample_limit'])
    assert {str(k): list(v) for k, v in m.tr_data.items()} == exp['tr_data']
    assert m.tr_users == exp['tr_users']
    m2 = BPR(k=4)
    m2.load_training_data(os.path.join(d, 'uid'), os.path.join(d, 'vid'), os.path.join(d, 'tr.txt'))
    assert not hasattr(m2, 'data')                               # bpr.py:67-68
    v = VBPR(k=8, d=5, lambda_e=0.1)
    assert (v.k, v.d, v.le) == (8, 5, 0.1) and isinstance(v, BPR)


def test_out_of_scope_models_raise():
    import single
    for name in ('WMF', 'DPM', 'CER', 'ENCODER', 'MLP'):
        with pytest.raises(NotImplementedError):
            getattr(single, name)(k=4)


def test_train_argument_checks_without_gpu(golden_dir):
    from single import BPR
    import tkr_hip
    d = os.path.join(golden_dir, 'g2')
    m = BPR(k=4)
    m.load_training_data(os.path.join(d, 'uid'), os.path.join(d, 'vid'), os.path.join(d, 'f0tr.txt'))
    with pytest.raises(AssertionError):
        m.train(epochs=1.5)
    with pytest.raises(AssertionError):
        m.train(epoch_sample_limit=10.5)
    with pytest.raises(ValueError):
        m.train(sampling='item uniform')
    with pytest.raises(ValueError):
        m.train(batch_size=10 ** 6)                              # reference loop would never end
    if not torch.cuda.is_available():
        with pytest.raises(tkr_hip.TkrError):                    # loud failure: no CPU fallback
            m.train(epochs=1, batch_size=16, epoch_sample_limit=10e1)


def test_export_import_embeddings_roundtrip(golden_dir, tmp_path):
    from single import BPR
    d = os.path.join(golden_dir, 'g2')
    m = BPR(k=3)
    m.load_training_data(os.path.join(d, 'uid'), os.path.join(d, 'vid'), os.path.join(d, 'f0tr.txt'))
    rng = np.random.Generator(np.random.PCG64(0))
    m.fue = (rng.standard_normal((m.n_users, 3)) * 0.01).astype(np.float32)
    m.fie = (rng.standard_normal((m.n_items, 3)) * 0.01).astype(np.float32)
    m.fib = (rng.standard_normal((m.n_items, 1)) * 0.01).astype(np.float32)
    target = str(tmp_path / 'model')
    m.export_embeddings(target)                                  # creates the directory (rec.py:48-50)
    assert sorted(os.listdir(target)) == ['final-B.dat', 'final-U.dat', 'final-V.dat']   # no engine -> no weights file
    m2 = BPR(k=3)
    m2.load_training_data(os.path.join(d, 'uid'), os.path.join(d, 'vid'), os.path.join(d, 'f0tr.txt'))
    m2.import_embeddings(target)
    for a, b in ((m.fue, m2.fue), (m.fie, m2.fie), (m.fib, m2.fib)):
        assert b.shape == a.shape and np.max(np.abs(a - b)) <= 5.1e-7        # '%f' keeps 6 decimals (+ fp32 rounding)
    from oracle import ref_np as R
    np.testing.assert_array_equal(m2.fue, R.read_embed_text(os.path.join(target, 'final-U.dat'), m.uids))


def test_load_content_data(tmp_path, golden_dir):
    import pickle
    import scipy.sparse as ss
    from single import VBPR
    from oracle import ref_np as R
    d = os.path.join(golden_dir, 'g2')
    m = VBPR(k=4, d=6)
    m.load_training_data(os.path.join(d, 'uid'), os.path.join(d, 'vid'), os.path.join(d, 'f0tr.txt'))
    fids = list(m.iids.keys())[::-1][:20]                        # features for a subset, in another order
    open(tmp_path / 'fid', 'w').write(''.join(x + '\n' for x in fids))
    rng = np.random.Generator(np.random.PCG64(1))
    dense = rng.random((20, 6)).astype(np.float32)
    for blob, name in ((dense, 'dense.pkl'), (ss.csr_matrix(dense * (dense > 0.5)), 'sparse.pkl')):
        pickle.dump(blob, open(tmp_path / name, 'wb'))
        m.load_content_data(str(tmp_path / name), str(tmp_path / 'fid'))
        exp = R.load_content(str(tmp_path / name), str(tmp_path / 'fid'), m.iids, m.n_items, 6)
        np.testing.assert_array_equal(m.feat, exp)
        assert m.feat.dtype == np.float32 and m.feat.shape == (m.n_items, 6)


def test_library_exports_every_declared_symbol():
    import tkr_hip
    header = open(os.path.join(ROOT, 'include', 'tkr.h')).read()
    declared = set(re.findall(r'^int(?:32_t|64_t)? (tkr_\w+)\(', header, flags=re.M))
    exported = set(tkr_hip.EXPORTS) | set(tkr_hip.EXPORTS_I64)
    assert declared == exported, declared ^ exported
    lib = ctypes.CDLL(tkr_hip.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name)
    # ... and the other direction: every tkr_* symbol the library exports is declared (helpers between its translation units are hidden)
    import subprocess
    nm = subprocess.run(['nm', '-D', '--defined-only', tkr_hip.LIB_PATH], capture_output=True, text=True, check=True).stdout
    dynamic = {ln.split()[-1] for ln in nm.splitlines() if ln.split() and ln.split()[-1].startswith('tkr_')}
    extra = {'tkr_k4_prof_read', 'tkr_debug_own_k1_prof'}        # profiling builds only (-DTKR_R2_PROF / -DTKR_PLAN_STAMP)
    assert dynamic - extra <= declared, (dynamic - extra) - declared
    assert lib.tkr_version() == tkr_hip.VERSION and lib.tkr_lab_build() in (0, 1)
    assert lib.tkr_plan_team(256) == 4 and lib.tkr_plan_team(8192) == 8 and lib.tkr_plan_team(65536) == 16
    assert lib.tkr_plan_max_blocks(256) == 192 + 153 and lib.tkr_plan_max_blocks(2048) == 768 + 1228 and lib.tkr_plan_max_blocks(8192) == 3072 + 1445
    # argument validation happens before any device access
    assert lib.tkr_score_topk(None, None, 0, None, None, 0, 0, None, 0, 0, None, None, None, 0, None) == -1


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    import tkr_hip
    monkeypatch.setattr(tkr_hip, '_lib', None)
    monkeypatch.setattr(tkr_hip, 'LIB_PATH', str(tmp_path / 'libtkr_hip.so'))
    with pytest.raises(tkr_hip.TkrError):
        tkr_hip.lib()


_GLOO_WORKER = r'''
import os, sys
sys.path[:0] = [%(root)r, %(pkg)r]
import numpy as np, torch, torch.distributed as dist
import dist as tdist
rank = int(os.environ['RANK']); world = int(os.environ['WORLD_SIZE'])
dist.init_process_group('gloo')
assert tdist.world() == (rank, world)
users = list(range(11))
mine = tdist.shard_users(users, rank, world)
allu = [None] * world
dist.all_gather_object(allu, mine)
assert sorted(sum(allu, [])) == users and len(set(sum(allu, []))) == len(users)
assert tdist.batches_per_rank(3906, world) == 3906 // world            # (epoch_sample_limit // B) // N of north_star

class Eng:                      # the engine surface ItemSync needs (device-agnostic tensors)
    def __init__(self):
        g = torch.Generator().manual_seed(0)
        self.t = {'V': [torch.randn(9, 4, generator=g), torch.ones(9, 4)], 'b': [torch.zeros(9), torch.ones(9)]}
    def get(self, n): return self.t[n][0], self.t[n][1]
    def set_replicated(self, new):
        for n, (p, ms) in new.items(): self.t[n] = [p.clone(), ms.clone()]
e = Eng()
V0 = e.t['V'][0].clone()
sync = tdist.ItemSync(e)
sync.begin()
e.t['V'][0] += (rank + 1) * 0.5          # each replica moves the items by its own delta
e.t['V'][1] *= (rank + 1)
e.t['b'][0][rank] = 1.0                  # disjoint bias updates
sync.end()
assert torch.allclose(e.t['V'][0], V0 + 0.5 * sum(range(1, world + 1)))        # P0 + sum of deltas
assert torch.allclose(e.t['V'][1], torch.full((9, 4), sum(range(1, world + 1)) / world))   # slots: mean
assert torch.allclose(e.t['b'][0][:world], torch.ones(world))
# user rows: every rank holds ONLY the rows of its users (5 users dealt round-robin -> shards of 3 and 2; of 1 and 0 on 8 ranks) and they are gathered once
owned = tdist.shard_users(list(range(5)), rank, world)
rows = torch.full((len(owned), 2), float(rank + 1)); slots = rows * 10
full, full_ms = np.zeros((5, 2), np.float32), np.zeros((5, 2), np.float32)
for ids, r_, m_ in tdist.gather_owned_rows(owned, rows, slots):
    full[ids], full_ms[ids] = r_, m_
exp = np.zeros((5, 2), np.float32)
for r in range(world): exp[r::world] = r + 1
assert np.array_equal(full, exp) and np.array_equal(full_ms, exp * 10)
assert tdist.shared_seed(None if rank else 12345) == 12345 and tdist.shared_seed(7 + rank) == 7      # rank 0's seed wins
dist.barrier(); dist.destroy_process_group()
print('ok', rank)
'''


@pytest.mark.parametrize('world', [2, 8])          # 8: the north_star node (one process per GPU), here on gloo
def test_item_sync_world_size_2_gloo(tmp_path, world):
    script = tmp_path / 'worker.py'
    script.write_text(_GLOO_WORKER % dict(root=ROOT, pkg=os.path.join(ROOT, 'top-k-rec_amd')))
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=%d' % world,
                          '--master-addr', '127.0.0.1', '--master-port', str(29631 + world), str(script)],
                         capture_output=True, text=True, timeout=400, env=dict(os.environ, OMP_NUM_THREADS='1'))
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count('ok') == world


# ---------------------------------------------------------------- native text I/O (csrc/textio.hip)
def _py_ratings(path, users, items):
    """per-field restatement of the three reference parsers' common core"""
    line_user, line_ptr, item, like = [], [0], [], []
    for line in open(path):
        terms = line.strip().split(',')
        line_user.append(users.get(terms[0], -1))
        for t in terms[1:]:
            item.append(items.get(t.split(':')[0], -1))
            like.append(int(t.split(':')[1]))
        line_ptr.append(len(item))
    return line_user, line_ptr, item, like


def test_native_ratings_parser_matches_python(golden_dir, tmp_path):
    import textio
    from oracle import ref_np as R
    cases = [(os.path.join(golden_dir, 'g1', 'tr.txt'), os.path.join(golden_dir, 'g1', 'uid'), os.path.join(golden_dir, 'g1', 'vid')),
             (os.path.join(golden_dir, 'g4', 'data', 'f0tr.txt'), os.path.join(golden_dir, 'g4', 'data', 'uid'),
              os.path.join(golden_dir, 'g4', 'data', 'f0te.om.idl')),
             (os.path.join(golden_dir, 'g7', 'data', 'f0te.sm.txt'), os.path.join(golden_dir, 'g7', 'data', 'uid'),
              os.path.join(golden_dir, 'g7', 'data', 'f0te.sm.idl'))]
    odd = tmp_path / 'odd.txt'                      # CRLF, blank line, spaces, no trailing newline, like with sign, extra ':' part
    odd.write_bytes(b'u1,a:1,b:0\r\n\n  u2,c: 1 ,a:+1:zz,\tq:-3\nu3\nzz,a:1\n u1 ,b:01')
    (tmp_path / 'u').write_text('u1\nu2\nu3\n')
    (tmp_path / 'v').write_text('a\nb\nc\n')
    cases.append((str(odd), str(tmp_path / 'u'), str(tmp_path / 'v')))
    for path, upath, vpath in cases:
        users, items = R.read_id_list(upath), R.read_id_list(vpath)
        got = textio.parse_ratings(path, users, items)
        lu, lp, it, lk = _py_ratings(path, users, items)
        assert got.line_user.tolist() == lu and got.line_ptr.tolist() == lp
        assert got.item.tolist() == it and got.like.tolist() == lk
        assert got.entry_user.tolist() == [lu[l] for l in range(len(lu)) for _ in range(lp[l + 1] - lp[l])]
    # where the reference raises, so does the parser
    bad = tmp_path / 'bad.txt'
    bad.write_text('u1,a\n')
    with pytest.raises(textio.TextFormatError):
        textio.parse_ratings(str(bad), {'u1': 0}, {'a': 0})
    bad.write_text('u1,a:x\n')
    with pytest.raises(textio.TextFormatError):
        textio.parse_ratings(str(bad), {'u1': 0}, {'a': 0})
    with pytest.raises(OSError):
        textio.parse_ratings(str(tmp_path / 'missing.txt'), {'u1': 0}, {'a': 0})
    empty = tmp_path / 'empty.txt'
    empty.write_text('')
    got = textio.parse_ratings(str(empty), {'u1': 0}, {'a': 0})
    assert len(got.line_user) == 0 and got.line_ptr.tolist() == [0]


def test_native_matrix_io_matches_reference_bytes(golden_dir, tmp_path, monkeypatch):
    import textio
    d = os.path.join(golden_dir, 'g3')
    exp = np.load(os.path.join(d, 'expected.npz'))
    for name in ('mat', 'bias'):
        out = tmp_path / (name + '.dat')
        textio.write_matrix(str(out), exp[name])
        assert out.read_bytes() == open(os.path.join(d, name + '.dat'), 'rb').read()      # reference export_embed_to_file
        assert not os.path.exists(str(out) + '.npy')                                      # TKR_NO_CACHE=1 in the test env
    np.testing.assert_array_equal(textio.read_matrix(os.path.join(d, 'mat.dat')), exp['back_all'])
    # random values incl. specials: same bytes as Python's '%f', same values back as np.float32(str)
    rng = np.random.Generator(np.random.PCG64(9))
    m = np.concatenate([rng.standard_normal(4000) * 10.0 ** rng.integers(-8, 6, 4000),
                        [0.0, -0.0, np.inf, -np.inf, np.nan, 3.4e38, -3.4e38, 1e-45, 0.0000005, 0.00000049999]]).astype(np.float32)
    m = m[:4008].reshape(-1, 8)
    path = tmp_path / 'r.dat'
    textio.write_matrix(str(path), m)
    text = ''.join(''.join('%f ' % v for v in row) + '\n' for row in m)
    assert path.read_text() == text
    back = textio.read_matrix(str(path))
    want = np.array([[np.float32(t) for t in line.strip().split(' ')] for line in text.strip('\n').split('\n')], dtype=np.float32)
    np.testing.assert_array_equal(back, want)
    # ragged rows and junk raise like numpy would
    path.write_text('1.0 2.0 \n3.0 \n')
    with pytest.raises(textio.TextFormatError):
        textio.read_matrix(str(path))
    path.write_text('1.0 x \n')
    with pytest.raises(textio.TextFormatError):
        textio.read_matrix(str(path))
    # n1: the binary copy is what the TEXT says (6 decimals); it is used only while the stamp inside it (size + mtime_ns of the text it
    # was parsed from) equals the text file's -- an OLDER text put in its place (cp -p, rsync -t, tar) must not be served from the copy
    monkeypatch.setenv('TKR_NO_CACHE', '0')
    textio.write_matrix(str(path), m[:3])
    cached = np.load(str(path) + '.npy')
    np.testing.assert_array_equal(cached, want[:3])
    np.save(str(path) + '.npy', np.full((3, 8), 7, np.float32))                            # same stamp: the copy is what is read ...
    assert np.all(textio.read_matrix(str(path)) == 7)
    old_stat = os.stat(str(path))
    path.write_text('1.5 2.5 \n')                                                           # ... until the text changes,
    os.utime(str(path), ns=(old_stat.st_atime_ns, old_stat.st_mtime_ns - 10 ** 10))         # even to a file with an OLDER timestamp
    os.utime(str(path) + '.npy', None)                                                      # and a copy that is newer than it
    np.testing.assert_array_equal(textio.read_matrix(str(path)), np.array([[1.5, 2.5]], np.float32))
    np.testing.assert_array_equal(np.load(str(path) + '.npy'), np.array([[1.5, 2.5]], np.float32))
    path.write_text('9.0 2.5 \n')                                                           # same size, same second, different content
    np.testing.assert_array_equal(textio.read_matrix(str(path)), np.array([[9.0, 2.5]], np.float32))


def test_ratings_cache_is_stamped(tmp_path, monkeypatch):
    """n1 for the rating files: parse_ratings keeps its flat arrays beside the text (<file>.csr.npz) and uses them only for the same
    text (size + mtime_ns) AND the same id tables"""
    import textio
    monkeypatch.setenv('TKR_NO_CACHE', '0')
    path = tmp_path / 'f0tr.txt'
    path.write_text('u1,a:1,b:0\nu2,b:1\nu9,a:1\n')
    users, items = {'u1': 0, 'u2': 1}, {'a': 0, 'b': 1}
    first = textio.parse_ratings(str(path), users, items)
    assert os.path.exists(str(path) + '.csr.npz')
    again = textio.parse_ratings(str(path), users, items)
    for name in ('line_user', 'line_ptr', 'item', 'like'):
        np.testing.assert_array_equal(getattr(first, name), getattr(again, name))
    assert first.line_user.tolist() == [0, 1, -1] and first.item.tolist() == [0, 1, 1, 0] and first.like.tolist() == [1, 0, 1, 1]
    swapped = textio.parse_ratings(str(path), users, {'a': 1, 'b': 0})                        # other id table: not served from the copy
    assert swapped.item.tolist() == [1, 0, 0, 1]
    st = os.stat(str(path))
    path.write_text('u1,a:0,b:0\nu2,b:1\nu9,a:1\n')                                          # same size, older timestamp
    os.utime(str(path), ns=(st.st_atime_ns, st.st_mtime_ns - 10 ** 10))
    assert textio.parse_ratings(str(path), users, items).like.tolist() == [0, 0, 1, 1]


def test_utils_history_and_ivt_match_reference(golden_dir):
    """utils.py:18-24, 73-89 mirrors against golden G9's counter and the oracle"""
    import utils
    from oracle import ref_np as R
    exp = json.load(open(os.path.join(golden_dir, 'g9', 'expected.json')))
    path = os.path.join(golden_dir, 'g4', 'data', 'f0tr.txt')
    browsed, counter = utils.get_history_from_file(path)
    assert counter == exp['counter'] and (browsed, counter) == R.read_history_counts(path)
    assert utils.get_history_from_file(path + '.missing') == ({}, {})
    idl = os.path.join(golden_dir, 'g4', 'data', 'f0te.om.idl')
    assert utils.get_iv_dict_from_file(idl) == R.read_iv_list(idl) and utils.get_iv_dict_from_file(idl + '.missing') == {}
    if not torch.cuda.is_available():
        with pytest.raises(Exception, match='MI355X'):
            utils.get_score(np.zeros((2, 2), np.float32), np.zeros((2, 2), np.float32), {'a': 0}, {'a': 0})


def test_native_ratings_parser_fuzz(tmp_path):
    """random well-formed files with hostile spacing / line endings / signs: the native parser and the per-field Python
    restatement of the reference's loops agree entry for entry"""
    import textio
    rng = np.random.Generator(np.random.PCG64(77))
    users = {'u%d' % x: x for x in range(40)}
    items = {'i%d' % x: x for x in range(60)}
    items['odd id'] = 60                                            # ids are arbitrary tokens: inner spaces survive
    for trial in range(25):
        lines = []
        for _ in range(int(rng.integers(0, 30))):
            uid = rng.choice(['u%d' % rng.integers(0, 50), ' u%d' % rng.integers(0, 40), ''])
            fields = []
            for _ in range(int(rng.integers(0, 12))):
                iid = rng.choice(['i%d' % rng.integers(0, 70), 'odd id', 'i%d ' % rng.integers(0, 60)])
                like = rng.choice(['0', '1', '+1', '-1', ' 1', '1 ', '01', '5', '1:extra', '0:1'])
                fields.append('%s:%s' % (iid, like))
            lines.append(','.join([uid] + fields) + rng.choice(['\n', '\r\n', ' \n', '\t\n']))
        text = ''.join(lines)
        if trial % 3 == 0:
            text = text.rstrip('\r\n\t ')                         # no trailing newline
        path = tmp_path / ('f%d.txt' % trial)
        path.write_bytes(text.encode())
        got = textio.parse_ratings(str(path), users, items)
        lu, lp, it, lk = _py_ratings(str(path), users, items)
        assert got.line_user.tolist() == lu and got.line_ptr.tolist() == lp, trial
        assert got.item.tolist() == it and got.like.tolist() == lk, trial


def test_evaluate_shards_partition_the_scenario(golden_dir):
    """evaluate.shard_scenario (SURVEY.md §8e scoring): the ranks' blocks are disjoint, cover every test line in order, and carry
    exactly the like / rated columns of their lines -- so the all-reduced hit counters and like counts equal the single-process ones"""
    import evaluate as E
    data = os.path.join(golden_dir, 'g4', 'data')
    uids = E.read_ids(os.path.join(data, 'uid'))
    full = E.load_scenario(data, 0, 'im', uids)
    for world in (1, 2, 3, 7, len(full.users) + 5):
        parts = [E.shard_scenario(full, r, world) for r in range(world)]
        assert np.array_equal(np.concatenate([p.users for p in parts]), full.users)
        assert sum(p.tcount for p in parts) == full.tcount
        at = 0
        for p in parts:
            for q in range(len(p.users)):
                assert np.array_equal(p.like_cols[p.like_ptr[q]:p.like_ptr[q + 1]], full.like_cols[full.like_ptr[at]:full.like_ptr[at + 1]])
                assert np.array_equal(p.rated_cols[p.rated_ptr[q]:p.rated_ptr[q + 1]], full.rated_cols[full.rated_ptr[at]:full.rated_ptr[at + 1]])
                at += 1
        assert at == len(full.users)
