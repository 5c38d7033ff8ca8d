"""-m gpu: BASELINE.json north_star's accuracy statement -- "identical accuracy@{5..30} to the reference within +-0.001" -- at
the place where it is defined: the MovieLens-10M shape, the reference's own driver settings (train.py:3-6: BPR(k=50), defaults
lr = 1e-4, lambda = 2.5e-3 / 2.5e-3 / 2.5e-4 / 0, 5 epochs of 10^6 // 256 batches of 256), through the real text path
(export_embeddings -> final-*.dat -> evaluate.py).  VERDICT r3 #4/#5: the only sharded-vs-single accuracy test ran 3,000 users at
lr = 1e-2 (tests/test_gpu_dist.py keeps that study of the O(lr^2) bias of a once-per-epoch exchange).

  1. HIP path vs oracle on the SAME stream: one reference epoch on the device, the oracle (oracle/ref_np.bpr_step) replays the same
     999,936 triplets from the same init; both models go through the text export and the CLI -> the same accuracy@{5..30}.
  2. 8 user shards (8 ranks, gloo, one GPU) with the per-epoch exchange vs the single stream, seed-averaged -> within +-0.001.

Data: synth.ML10M_SIGNAL (the preset on which a trained model beats the popularity ranking; real uid / vid sizes), written in the
reference's text formats."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEEDS = (11, 12, 13)
K, LR, EPOCHS, B, LIMIT = 50, 1e-4, 5, 256, 10 ** 6            # /root/reference/train.py:3-6

_WORKER = r'''
import os, sys
sys.path[:0] = [%(root)r, %(pkg)r]
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group('gloo')
from single import BPR
for seed in %(seeds)r:
    m = BPR(k=%(k)d)
    m.load_training_data(%(data)r + '/uid', %(data)r + '/vid', %(data)r + '/f0tr.txt')
    m.train(epochs=%(epochs)d, batch_size=%(B)d, epoch_sample_limit=%(limit)d, seed=seed, verbose=False)
    if dist.get_rank() == 0:
        m.export_embeddings(%(out)r + str(seed))
dist.barrier(); dist.destroy_process_group()
'''


@pytest.fixture(scope='module')
def dataset(tmp_path_factory):
    sys.path.insert(0, os.path.join(ROOT, 'top-k-rec_amd'))
    import synth
    r = synth.make_ratings(seed=42, **synth.ML10M_SIGNAL)
    data = str(tmp_path_factory.mktemp('ml10m') / 'data')
    synth.write_dataset(data, r)
    return data


def _acc(data, model):
    import evaluate as E
    line = E.main(['-d', data, '-m', model, '-sl', 'im'])[0]
    return np.array([float(x) for x in line.split(',')[1:]])


def test_device_epoch_and_oracle_replay_give_the_same_accuracy(dataset, tmp_path):
    from single import BPR, _engine
    from oracle import plan_np as P, ref_np as R
    from utils import export_embed_to_file
    data = dataset
    m = BPR(k=K)
    m.load_training_data(data + '/uid', data + '/vid', data + '/f0tr.txt')
    m.train(epochs=1, batch_size=B, epoch_sample_limit=LIMIT, seed=SEEDS[0], verbose=False)
    m.export_embeddings(str(tmp_path / 'hip'))
    # the oracle from the same init (the tables before training are reproducible from the seed) on the same stream
    eng0 = _engine.BprEngine(m.n_users, m.n_items, K, m._hyper(), torch.device('cuda'), seed=SEEDS[0])
    ref = dict(U=eng0.get('U')[0].cpu().numpy(), V=eng0.get('V')[0].cpu().numpy(), b=np.zeros(m.n_items, np.float32))
    for n in ('U', 'V', 'b'):
        ref['ms' + n] = np.ones_like(ref[n])
    del eng0
    row_ptr, pos, srt = P.build_csr(m.tr_data, m.n_users)
    nb = LIMIT // B
    u, i, j = P.sample_triplets(m.tr_users, row_ptr, pos, srt, m.n_items, SEEDS[0], 0, nb * B)
    for bb in range(nb):
        sl = slice(bb * B, (bb + 1) * B)
        R.bpr_step(ref, u[sl], i[sl], j[sl], m._hyper())
    np.testing.assert_allclose(m.fie, ref['V'], rtol=2e-4, atol=1e-6)
    os.mkdir(tmp_path / 'oracle')
    export_embed_to_file(str(tmp_path / 'oracle' / 'final-U.dat'), ref['U'])
    export_embed_to_file(str(tmp_path / 'oracle' / 'final-V.dat'), ref['V'])
    export_embed_to_file(str(tmp_path / 'oracle' / 'final-B.dat'), ref['b'].reshape(-1, 1))
    a, b = _acc(data, str(tmp_path / 'hip')), _acc(data, str(tmp_path / 'oracle'))
    print('accuracy@{5..30}: HIP path', a, 'oracle replay', b)
    assert np.max(np.abs(a - b)) <= 1e-4, (a, b)                   # north_star: +-0.001; the two differ by rounding of the '%f' text at most


def test_eight_shards_track_the_single_stream_at_the_reference_settings(dataset, tmp_path):
    from single import BPR
    data = dataset
    for seed in SEEDS:
        m = BPR(k=K)
        m.load_training_data(data + '/uid', data + '/vid', data + '/f0tr.txt')
        m.train(epochs=EPOCHS, batch_size=B, epoch_sample_limit=LIMIT, seed=seed, verbose=False)
        m.export_embeddings(str(tmp_path / ('single%d' % seed)))
    script = tmp_path / 'worker.py'
    script.write_text(_WORKER % dict(root=ROOT, pkg=os.path.join(ROOT, 'top-k-rec_amd'), data=data, out=str(tmp_path / 'sharded'), k=K, epochs=EPOCHS, B=B,
                                     limit=LIMIT, seeds=[100 + s for s in SEEDS]))
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=8', '--master-addr', '127.0.0.1',
                          '--master-port', '29681', str(script)], capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    single = np.stack([_acc(data, str(tmp_path / ('single%d' % s))) for s in SEEDS])
    sharded = np.stack([_acc(data, str(tmp_path / ('sharded%d' % (100 + s)))) for s in SEEDS])
    untrained = BPR(k=K)
    untrained.load_training_data(data + '/uid', data + '/vid', data + '/f0tr.txt')
    rng = np.random.Generator(np.random.PCG64(1))
    untrained.fue = (rng.standard_normal((untrained.n_users, K)) * 0.01).astype(np.float32)
    untrained.fie = (rng.standard_normal((untrained.n_items, K)) * 0.01).astype(np.float32)
    untrained.fib = np.zeros((untrained.n_items, 1), np.float32)
    untrained.export_embeddings(str(tmp_path / 'untrained'))
    base = _acc(data, str(tmp_path / 'untrained'))
    print('accuracy@{5..30}: single mean', single.mean(0), 'std', single.std(0), '| 8 shards mean', sharded.mean(0), 'std', sharded.std(0), '| untrained', base)
    assert np.max(np.abs(single.mean(0) - sharded.mean(0))) <= 1e-3, (single.mean(0), sharded.mean(0))      # north_star's +-0.001
    assert single.mean(0)[-1] > 1.5 * base[-1]                     # and the comparison is not between two untrained models
