"""Which reduction rule for the RMSProp slots of the replicated item tables tracks the single-stream run best?

S user shards are simulated inside one process (one BprEngine per shard, bulk layout, the arithmetic of dist.ItemSync applied
by hand once per epoch) at the configuration of tests/test_gpu_dist.py::test_sharded_accuracy_tracks_single_stream, several seeds.
Rules for the slot of an item element (parameters always: P0 + sum of deltas):
  mean      ms <- mean_r ms_r                                   (rounds 1-2)
  recombine ms_r = a_r * ms0 + c_r with a_r = rho^{n_r} (n_r = updates of the row on shard r: the update counter);
            ms <- a * ms0 + (1 - a) * g2,  a = prod_r a_r,  g2 = mean over shards with n_r > 0 of c_r / (1 - a_r)
            (exact for pure decay and for a constant squared gradient, the two limits of the single stream)
Prints accuracy@{5..30} per rule averaged over the seeds next to the single-stream figure.
    python scripts/probe_exchange_rule.py [S] [seeds]
"""
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'top-k-rec_amd')]
os.environ.setdefault('TKR_NO_CACHE', '1')
os.environ['TKR_FLOW'] = '0'          # every engine here stays in the bulk layout (the arithmetic under test is the exchange, not the step kernel)

import dist as tdist          # noqa: E402
import evaluate as E          # noqa: E402
import synth                  # noqa: E402
from single import BPR, _engine  # noqa: E402

RHO = 0.9


def train_sharded(m, S, epochs, B, seed, rule):
    dev = torch.device('cuda')
    n_batches = m.epoch_sample_limit // B
    nb = max(1, n_batches // S)
    engines = [_engine.BprEngine(m.n_users, m.n_items, m.k, m._hyper(), dev, seed) for _ in range(S)]
    csrs = []
    for i, e in enumerate(engines):
        e.prepare(B, 'bulk')
        if i:
            e.copy_model_from(engines[0])
        csrs.append(m._make_csr(tdist.shard_users(m.tr_users, i, S), dev))
        e.triplets_drawn = i * epochs * nb * B
    lead = engines[0]
    u0, mu0 = (t.clone() for t in lead.get('U'))
    for _ in range(epochs):
        start = {n: tuple(t.clone() for t in lead.get(n)) for n in ('V', 'b')}
        for e, csr in zip(engines, csrs):
            e.run_batches(csr, nb, B, want_loss=False)
        counts = [e.cnt.icnt.clone().float() for e in engines]
        new = {}
        for n in ('V', 'b'):
            p0, ms0 = start[n]
            ps = [e.get(n) for e in engines]
            p = p0 + sum(x - p0 for x, _ in ps)
            if rule == 'mean':
                ms = sum(y for _, y in ps) / S
            else:
                shape = (-1, 1) if p0.dim() == 2 else (-1,)
                a_r = [torch.pow(RHO, c).reshape(shape) for c in counts]
                a = torch.ones_like(a_r[0])
                g2 = torch.zeros_like(ms0)
                live = torch.zeros_like(a_r[0])
                for (x, y), ar in zip(ps, a_r):
                    a = a * ar
                    on = (ar < 1).float()
                    g2 = g2 + on * (y - ar * ms0) / (1 - ar).clamp_min(1e-12)
                    live = live + on
                g2 = g2 / live.clamp_min(1)
                ms = a * ms0 + (1 - a) * g2
            new[n] = (p, ms)
        for e in engines:
            e.set_replicated(new)
    parts = [e.get('U') for e in engines]
    m.fue = (u0 + sum(p - u0 for p, _ in parts)).cpu().numpy()
    m.fie = lead.get('V')[0].cpu().numpy()
    m.fib = lead.get('b')[0].reshape(-1, 1).cpu().numpy()


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    n_seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    tmp = tempfile.mkdtemp()
    r = synth.make_ratings(3000, 700, 0, seed=5, mu=3.6, sigma=0.6, min_r=8, max_r=150, alpha=0.6, gain=2.0, select=4.0)
    data = os.path.join(tmp, 'data')
    synth.write_dataset(data, r)

    def acc_of(m, name):
        out = os.path.join(tmp, name)
        m.export_embeddings(out)
        line = E.main(['-d', data, '-m', out, '-sl', 'im'])[0]
        return np.array([float(x) for x in line.split(',')[1:]])

    def fresh():
        m = BPR(k=16, lr=1e-2)
        m.load_training_data(data + '/uid', data + '/vid', data + '/f0tr.txt')
        return m
    res = {'single': [], 'mean': [], 'recombine': []}
    for s in range(n_seeds):
        m = fresh()
        m.train(epochs=12, batch_size=256, seed=7 + s, verbose=False)
        res['single'].append(acc_of(m, 'single%d' % s))
        for rule in ('mean', 'recombine'):
            m = fresh()
            m._eng = None
            train_sharded(m, S, 12, 256, 107 + s, rule)
            res[rule].append(acc_of(m, '%s%d' % (rule, s)))
    for k, v in res.items():
        v = np.stack(v)
        print('%-10s mean %s  std %s' % (k, np.round(v.mean(0), 5), np.round(v.std(0), 5)))
    for rule in ('mean', 'recombine'):
        print('%-10s - single: %s' % (rule, np.round(np.stack(res[rule]).mean(0) - np.stack(res['single']).mean(0), 5)))


if __name__ == '__main__':
    main()
