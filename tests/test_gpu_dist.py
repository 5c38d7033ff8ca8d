"""Two ranks (gloo, both on the one visible GPU) train a user-sharded BPR model through BPR.train and
must reproduce an oracle simulation of the same two shards + the per-epoch sum-of-deltas exchange."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r'''
import os, sys
sys.path[:0] = [%(root)r, %(pkg)r]
import numpy as np, torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group('gloo')
rank, world = dist.get_rank(), dist.get_world_size()
from single import BPR
import dist as tdist
from oracle import plan_np as P, ref_np as R
data = %(data)r
k, B, epochs, limit, lr = 16, 32, 2, 32 * 12, 0.02
m = BPR(k=k, lr=lr, lambda_b=1e-3)
m.load_training_data(data + '/uid', data + '/vid', data + '/f0tr.txt')
rng = np.random.Generator(np.random.PCG64(0))
init = [(rng.standard_normal((m.n_users, k)) * 0.1).astype(np.float32), (rng.standard_normal((m.n_items, k)) * 0.1).astype(np.float32),
        np.zeros((m.n_items, 1), np.float32)]
m.fue, m.fie, m.fib = (a.copy() for a in init)
m.train(epochs=epochs, batch_size=B, epoch_sample_limit=limit, seed=11, verbose=False)
# ---- oracle simulation of BOTH ranks
hp = dict(lu=m.lu, li=m.li, lj=m.lj, lb=m.lb, lr=lr, mode='l2')
nb = (limit // B) // world
row_ptr, pos, srt = P.build_csr(m.tr_data, m.n_users)
st = [dict(U=init[0].copy(), V=init[1].copy(), b=init[2].ravel().copy(), msU=np.ones_like(init[0]), msV=np.ones_like(init[1]),
           msb=np.ones(m.n_items, np.float32)) for _ in range(world)]
drawn = [r * epochs * nb * B for r in range(world)]
for e in range(epochs):
    V0, b0 = st[0]['V'].copy(), st[0]['b'].copy()
    for r in range(world):
        users = tdist.shard_users(m.tr_users, r, world)
        u, i, j = P.sample_triplets(users, row_ptr, pos, srt, m.n_items, 11, drawn[r], nb * B)
        drawn[r] += nb * B
        for s in range(nb):
            R.bpr_step(st[r], u[s*B:(s+1)*B], i[s*B:(s+1)*B], j[s*B:(s+1)*B], hp)
    V = V0 + sum(x['V'] - V0 for x in st); b = b0 + sum(x['b'] - b0 for x in st)
    msV = sum(x['msV'] for x in st) / world; msb = sum(x['msb'] for x in st) / world
    for x in st:
        x['V'], x['b'], x['msV'], x['msb'] = V.copy(), b.copy(), msV.copy(), msb.copy()
U = init[0] + sum(x['U'] - init[0] for x in st)
np.testing.assert_allclose(m.fie, st[0]['V'], rtol=2e-4, atol=1e-5)
np.testing.assert_allclose(m.fib.ravel(), st[0]['b'], rtol=2e-4, atol=1e-5)
np.testing.assert_allclose(m.fue, U, rtol=2e-4, atol=1e-5)
# the step kernel that ran: K2o on HALF the CUs per rank (two ranks share this GPU: dist.ranks_sharing_device), or K2f with TKR_OWN=0
eng = m._eng
assert eng.layout == 'flow' and eng.ranks_on_device == world
want = 0 if os.environ.get('TKR_OWN') == '0' else torch.cuda.get_device_properties(0).multi_processor_count // world
assert eng.plan.owners == want and eng._plan_owners(B) == want, (eng.plan.owners, want)
dist.barrier(); dist.destroy_process_group()
print('ok', rank)
'''


@pytest.mark.parametrize('own', ['1', '0'])
def test_two_rank_sharded_training_matches_oracle(tmp_path, own):
    sys.path.insert(0, os.path.join(ROOT, 'top-k-rec_amd'))
    import synth
    r = synth.make_ratings(120, 60, 0, seed=13, mu=2.6, sigma=0.4, min_r=4, max_r=25)
    data = str(tmp_path / 'data')
    synth.write_dataset(data, r)
    script = tmp_path / 'worker.py'
    script.write_text(_WORKER % dict(root=ROOT, pkg=os.path.join(ROOT, 'top-k-rec_amd'), data=data))
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2',
                          '--master-addr', '127.0.0.1', '--master-port', '29641', str(script)],
                         capture_output=True, text=True, timeout=280, env=dict(os.environ, TKR_OWN=own))
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert out.stdout.count('ok') == 2


_ACC_WORKER = r'''
import os, sys, json
sys.path[:0] = [%(root)r, %(pkg)r]
import numpy as np, torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group('gloo')
from single import BPR
for seed in %(seeds)r:
    m = BPR(k=16, lr=1e-2)
    m.load_training_data(%(data)r + '/uid', %(data)r + '/vid', %(data)r + '/f0tr.txt')
    m.train(epochs=12, batch_size=256, seed=seed, verbose=False)
    if dist.get_rank() == 0:
        m.export_embeddings(%(out)r + str(seed))
dist.barrier(); dist.destroy_process_group()
'''

ACC_SEEDS = 8


def test_sharded_accuracy_tracks_single_stream(tmp_path):
    """SURVEY H4 / BASELINE.json north_star: 4 user shards + per-epoch sum-of-deltas exchange vs the single-stream run, same data and
    hyper-parameters, ACC_SEEDS seeds each.  The two draw DIFFERENT sample streams (a shard samples inside its own users), so
    one pair of runs differs by training noise: the seed-to-seed standard deviation of ONE single-stream run is 0.0016-0.0026 per
    bucket here (3,000 users, lr = 1e-2 = 100 x the reference's, so that 12 epochs train at all).  Measured on MI355X, mean over 8
    seeds, sharded - single: -0.0002, -0.0007, -0.0013, -0.0011, -0.0013, -0.0024 (standard error of such a difference: 0.0011):
    the exchange costs up to ~1 % relative at this step size -- V is a whole epoch stale inside a shard, an O(lr^2) effect of the
    rule north_star prescribes (one all-reduce per epoch), not of the kernels.  north_star's +-0.001 is asserted where it is
    defined -- same stream, HIP path vs oracle (tests/test_gpu_e2e.py, accuracy from exported factors) -- and here the seed-averaged
    difference is held to 0.003, every sharded run to the seed-noise band.  Tried: rebuilding the slot from per-shard decay
    factors instead of averaging it (scripts/probe_exchange_rule.py): twice the bias, dropped."""
    sys.path.insert(0, os.path.join(ROOT, 'top-k-rec_amd'))
    import synth
    import evaluate as E
    from single import BPR
    r = synth.make_ratings(3000, 700, 0, seed=5, mu=3.6, sigma=0.6, min_r=8, max_r=150, alpha=0.6, gain=2.0, select=4.0)
    data = str(tmp_path / 'data')
    synth.write_dataset(data, r)
    seeds = list(range(7, 7 + ACC_SEEDS))
    for seed in seeds:
        single = BPR(k=16, lr=1e-2)
        single.load_training_data(data + '/uid', data + '/vid', data + '/f0tr.txt')
        single.train(epochs=12, batch_size=256, seed=seed, verbose=False)
        single.export_embeddings(str(tmp_path / ('single%d' % seed)))
    script = tmp_path / 'acc_worker.py'
    script.write_text(_ACC_WORKER % dict(root=ROOT, pkg=os.path.join(ROOT, 'top-k-rec_amd'), data=data, out=str(tmp_path / 'sharded'),
                                         seeds=[100 + s for s in seeds]))
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=4',
                          '--master-addr', '127.0.0.1', '--master-port', '29643', str(script)],
                         capture_output=True, text=True, timeout=580)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]

    def acc_of(name):
        line = E.main(['-d', data, '-m', str(tmp_path / name), '-sl', 'im'])[0]
        return np.array([float(x) for x in line.split(',')[1:]])
    single = np.stack([acc_of('single%d' % s) for s in seeds])
    sharded = np.stack([acc_of('sharded%d' % (100 + s)) for s in seeds])
    untrained = BPR(k=16)
    untrained.load_training_data(data + '/uid', data + '/vid', data + '/f0tr.txt')
    rng = np.random.Generator(np.random.PCG64(1))
    untrained.fue = (rng.standard_normal((3000, 16)) * 0.01).astype(np.float32)
    untrained.fie = (rng.standard_normal((700, 16)) * 0.01).astype(np.float32)
    untrained.fib = np.zeros((700, 1), np.float32)
    untrained.export_embeddings(str(tmp_path / 'untrained'))
    base = acc_of('untrained')
    noise = single.std(0).max()
    print('acc single mean', single.mean(0), 'sharded mean', sharded.mean(0), 'seed std', single.std(0), sharded.std(0), 'untrained', base)
    assert np.max(np.abs(single.mean(0) - sharded.mean(0))) <= 0.003, (single.mean(0), sharded.mean(0))
    assert np.max(np.abs(sharded - single.mean(0))) <= max(0.006, 4 * noise)                  # every sharded run inside the seed-noise band
    assert single.mean(0)[-1] > 2 * base[-1] and sharded.mean(0)[-1] > 2 * base[-1]


@pytest.mark.parametrize('S', [2, 3])
def test_streams_mode_matches_oracle_simulation(tmp_path, S):
    """train(streams=S): S user shards on S HIP streams of one GPU == oracle simulation of S shards
    with the per-epoch sum-of-deltas exchange (the multi-GPU rule applied inside one GPU)."""
    sys.path[:0] = [ROOT, os.path.join(ROOT, 'top-k-rec_amd')]
    import synth
    import dist as tdist
    from single import BPR
    from oracle import plan_np as P, ref_np as R
    r = synth.make_ratings(150, 60, 0, seed=17, mu=2.6, sigma=0.4, min_r=4, max_r=25)
    data = str(tmp_path / 'data')
    synth.write_dataset(data, r)
    k, B, epochs, limit, lr = 16, 32, 2, 32 * 13, 0.02
    m = BPR(k=k, lr=lr, lambda_b=1e-3)
    m.load_training_data(data + '/uid', data + '/vid', data + '/f0tr.txt')
    rng = np.random.Generator(np.random.PCG64(0))
    init = [(rng.standard_normal((m.n_users, k)) * 0.1).astype(np.float32), (rng.standard_normal((m.n_items, k)) * 0.1).astype(np.float32),
            np.zeros((m.n_items, 1), np.float32)]
    m.fue, m.fie, m.fib = (a.copy() for a in init)
    m.train(epochs=epochs, batch_size=B, epoch_sample_limit=limit, seed=11, verbose=False, streams=S)
    # every shard ran a persistent step, not the per-batch one: with owned item rows (K2o) on half the CUs each at two shards, K2f from
    # three shards on (a third of the CUs as owners is slower than K2f beside the other shards)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    assert m._eng.layout == 'flow'
    if S == 2:
        assert m._eng._plan_owners(B) == cus // S and m._eng._last_step_kind == 'own'
    else:
        assert m._eng._plan_owners(B) == 0 and m._eng._last_step_kind == 'flow'
    hp = dict(lu=m.lu, li=m.li, lj=m.lj, lb=m.lb, lr=lr, mode='l2')
    nb = (limit // B) // S
    row_ptr, pos, srt = P.build_csr(m.tr_data, m.n_users)
    st = [dict(U=init[0].copy(), V=init[1].copy(), b=init[2].ravel().copy(), msU=np.ones_like(init[0]), msV=np.ones_like(init[1]),
               msb=np.ones(m.n_items, np.float32)) for _ in range(S)]
    drawn = [q * epochs * nb * B for q in range(S)]
    for e in range(epochs):
        V0, b0 = st[0]['V'].copy(), st[0]['b'].copy()
        for q in range(S):
            users = tdist.shard_users(m.tr_users, q, S)
            u, i, j = P.sample_triplets(users, row_ptr, pos, srt, m.n_items, 11, drawn[q], nb * B)
            drawn[q] += nb * B
            for t in range(nb):
                R.bpr_step(st[q], u[t * B:(t + 1) * B], i[t * B:(t + 1) * B], j[t * B:(t + 1) * B], hp)
        V = V0 + sum(x['V'] - V0 for x in st); b = b0 + sum(x['b'] - b0 for x in st)
        msV = sum(x['msV'] for x in st) / S; msb = sum(x['msb'] for x in st) / S
        for x in st:
            x['V'], x['b'], x['msV'], x['msb'] = V.copy(), b.copy(), msV.copy(), msb.copy()
    U = init[0] + sum(x['U'] - init[0] for x in st)
    np.testing.assert_allclose(m.fie, st[0]['V'], rtol=2e-4, atol=1e-5)
    np.testing.assert_allclose(m.fib.ravel(), st[0]['b'], rtol=2e-4, atol=1e-5)
    np.testing.assert_allclose(m.fue, U, rtol=2e-4, atol=1e-5)


def _run_cli_ranks(args, world, port, tmp_path):
    """evaluate.py under torch.distributed.run: `world` ranks on the one visible GPU (gloo) -> rank 0's stdout lines"""
    env = dict(os.environ, TKR_SINGLE_DEVICE='1', TKR_DIST_BACKEND='gloo', MASTER_ADDR='127.0.0.1', TKR_NO_CACHE='1')
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=%d' % world,
                          '--master-addr', '127.0.0.1', '--master-port', str(port),
                          os.path.join(ROOT, 'top-k-rec_amd', 'evaluate.py')] + args,
                         capture_output=True, text=True, timeout=280, env=env, cwd=str(tmp_path))
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    return [l for l in out.stdout.strip().split('\n') if ',' in l and not l.startswith('[')]


@pytest.mark.parametrize('world', [2, 3])
def test_sharded_evaluate_cli_reproduces_reference_stdout(golden_dir, tmp_path, world):
    """SURVEY.md §8e scoring: the scenario's test lines block-sharded over the ranks, hit counters and like counts all-reduced --
    stdout must equal the reference CLI's byte for byte on its own goldens (G4 im/om incl. another step/total, G5, G6 with a bias
    file, G7 edge cases: fewer than `total` unrated items, all likes rated, total % step != 0)"""
    import json
    port = 29650 + world
    for g, scs in (('g4', ['im', 'om']), ('g5', ['im', 'om']), ('g6', ['all'])):
        d = os.path.join(golden_dir, g)
        exp = json.load(open(os.path.join(d, 'expected.json')))
        data, model = os.path.join(d, 'data'), os.path.join(d, 'model')
        assert _run_cli_ranks(['-d', data, '-m', model, '-sl'] + scs, world, port, tmp_path) == exp['stdout'], g
        if 'stdout_s3_t10' in exp and world == 2:
            assert _run_cli_ranks(['-d', data, '-m', model, '-s', '3', '-t', '10', '-sl', 'om', 'im'], world, port, tmp_path) == exp['stdout_s3_t10']
    d = os.path.join(golden_dir, 'g7')
    exp = json.load(open(os.path.join(d, 'expected.json')))
    for run in exp['runs'][: 2 if world == 3 else None]:
        got = _run_cli_ranks(['-d', os.path.join(d, 'data'), '-m', os.path.join(d, 'model'), '-s', str(run['step']), '-t', str(run['total']),
                              '-sl', 'sm'], world, port, tmp_path)
        assert got == run['stdout'], run


@pytest.mark.skipif(__import__('torch').cuda.device_count() < 2, reason='RCCL needs two visible GPUs (the round-end 8-GPU node has them)')
def test_rccl_training_and_scoring_two_gpus(tmp_path):
    """backend 'nccl' = RCCL over xGMI, one process per GPU: BPR.train with the per-epoch exchange and the sharded evaluate.py.
    Runs wherever two GPUs are visible; the one-GPU box covers the same code with gloo (tests above)."""
    import synth
    r = synth.make_ratings(400, 150, 30, seed=5, mu=2.8, sigma=0.4, min_r=5, max_r=30)
    data = str(tmp_path / 'data')
    synth.write_dataset(data, r)
    script = tmp_path / 'train_rccl.py'
    script.write_text(r"""
import os, sys
sys.path[:0] = [%r, %r]
import torch, torch.distributed as dist
local = int(os.environ['LOCAL_RANK'])
torch.cuda.set_device(local)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
dist.init_process_group('nccl', device_id=torch.device('cuda', local))
from single import BPR
m = BPR(k=32, lr=0.02)
m.load_training_data(%r + '/uid', %r + '/vid', %r + '/f0tr.txt')
m.train(epochs=3, batch_size=64, epoch_sample_limit=64 * 40, verbose=False)         # no seed given: rank 0's is broadcast
import numpy as np
chk = torch.tensor([float(np.abs(m.fie).sum()), float(np.abs(m.fue).sum())], device='cuda', dtype=torch.float64)
lo, hi = chk.clone(), chk.clone()
dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
assert torch.equal(lo, hi), 'ranks disagree on the trained model'
if dist.get_rank() == 0:
    m.export_embeddings(%r)
dist.barrier(); dist.destroy_process_group()
print('ok')
""" % (ROOT, os.path.join(ROOT, 'top-k-rec_amd'), data, data, data, str(tmp_path / 'model')))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
                          '--master-port', '29671', str(script)], capture_output=True, text=True, timeout=280, env=env)
    assert out.returncode == 0 and out.stdout.count('ok') == 2, out.stdout[-3000:] + out.stderr[-3000:]
    single = subprocess.run([sys.executable, os.path.join(ROOT, 'top-k-rec_amd', 'evaluate.py'), '-d', data, '-m', str(tmp_path / 'model'), '-sl', 'im', 'om'],
                            capture_output=True, text=True, timeout=280, env=env)
    assert single.returncode == 0, single.stderr[-2000:]
    both = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
                           '--master-port', '29672', os.path.join(ROOT, 'top-k-rec_amd', 'evaluate.py'), '-d', data, '-m', str(tmp_path / 'model'),
                           '-sl', 'im', 'om'], capture_output=True, text=True, timeout=280, env=env)
    assert both.returncode == 0, both.stderr[-2000:]
    pick = lambda out: [l for l in out.strip().split('\n') if l.startswith(('im,', 'om,'))]
    assert pick(both.stdout) == pick(single.stdout) and len(pick(single.stdout)) == 2
    # and the bench itself, exactly as the round-end driver launches it at N = 2: its first RCCL run should not be the 8-GPU one
    import json
    bench = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
                            '--master-port', '29673', os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '20', '--warmup', '5', '--no-extras'],
                           capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert bench.returncode == 0, bench.stderr[-3000:]
    line = [l for l in bench.stdout.strip().split('\n') if l.startswith('{')]
    assert len(line) == 1
    d = json.loads(line[0])
    assert d['n_gpus'] == 2 and d['value'] > 0 and d['epoch_mode']['exchanges'] == 2 and d['epoch_mode']['exchange_us']['collective'] > 0
