"""Legacy BPR with the plain-SGD update, on MI355X.

Mirror of the reference's ``old/methods/bpr.py`` class surface -- ``BPR(K, users, items,
lambda_u, lambda_i, lambda_j, lambda_bias, learning_rate)`` (:17-29), ``train(train_data,
epochs, batch_size)`` (:63-86), the parameters ``W``, ``H``, ``B`` with ``get_value()`` /
``set_value()`` like the Theano shared variables they replace (:36-38) -- on the same kernels
as ``single/bpr.py``: K1 draws and plans the (u, i, j) batches on the device, K2 runs with
``tkr_bpr_state.opt = 1``: ``P <- P - lr * dcost/dP`` (:57-61) on the rows a batch touches,
no RMSProp slot traffic.  The objective (:43-51) is the 'l2' objective of single/bpr.py:92-95.

Stated differences from the reference:
  * the sample stream is the device Philox stream of K1 (same distribution: uniform user with
    positives, uniform positive incl. duplicates, uniform negative outside the user's
    positives), not the pre-generated ``numpy.random`` arrays of :88-99; the restatement of
    that sampler is oracle/ref_np.py ``legacy_pregenerated_sampler`` (golden G8);
  * the number of batches follows :72 exactly (``while (z+1)*batch_size < n_sgd_samples``);
    the per-batch ``\\rProcessed ...`` progress writes are not emitted (they would serialise
    the launch chain), the first and the last stderr lines are;
  * initial ``W``, ``H`` ~ 0.01 * N(0, 1), ``B`` = 0 (:36-38) come from the engine's seeded
    device generator, not from ``numpy.random``.
"""
from __future__ import annotations

import sys
import time
import numpy as np
import torch

from single._engine import BprEngine, TrainingCSR


class _Shared:
    """get_value()/set_value() view of one engine table (stands in for theano.shared)"""

    def __init__(self, model, name):
        self._model, self._name = model, name

    def get_value(self):
        return self._model._engine.get(self._name)[0].cpu().numpy()

    def set_value(self, value):
        eng = self._model._engine
        value = np.ascontiguousarray(value, dtype=np.float32)
        if self._name == 'U':
            eng.set_users(U=value)
        elif self._name == 'V':
            eng.set_items(V=value)
        else:
            eng.set_items(b=value)


class BPR(object):
    """Constructor and ``train`` keep the reference's argument lists (old/methods/bpr.py:17,63); everything behind them is
    this build's: the id maps and hyper-parameters live in two small records, the positives go straight into the device CSR."""

    def __init__(self, K, users, items, lambda_u=0.0025, lambda_i=0.0025, lambda_j=0.00025, lambda_bias=0.0,
                 learning_rate=1.0e-4, *, seed=None, device=None):
        self.ids = {'users': users, 'items': items}                     # external id -> row index, as handed in
        self.shape = (len(users), len(items), K)
        self.hyper = dict(lu=lambda_u, li=lambda_i, lj=lambda_j, lb=lambda_bias, lr=learning_rate, mode='l2', opt='sgd')
        self._engine = BprEngine(self.shape[0], self.shape[1], K, self.hyper, device=device, seed=seed)
        self.W, self.H, self.B = _Shared(self, 'U'), _Shared(self, 'V'), _Shared(self, 'b')
        self.positives = {}                                             # row index of a user -> its item row indices, input order
        self.losses = None

    def _group_positives(self, pairs):
        """(user id, item id) pairs -> {user row: [item rows]} in input order, duplicates kept (they weight the positive draw)"""
        u_of, i_of = self.ids['users'], self.ids['items']
        rows = np.fromiter((u_of[u] for u, _ in pairs), dtype=np.int64, count=len(pairs))
        cols = np.fromiter((i_of[i] for _, i in pairs), dtype=np.int64, count=len(pairs))
        order = np.argsort(rows, kind='stable')
        cuts = np.flatnonzero(np.diff(rows[order])) + 1
        return {int(rows[order[g[0]]]): cols[order[g]].tolist() for g in np.split(np.arange(len(pairs)), cuts) if len(g)}

    def train(self, train_data, epochs=30, batch_size=256):
        n_pairs = len(train_data)
        if n_pairs < batch_size:                                         # message text of old/methods/bpr.py:65
            sys.stderr.write("WARNING: Batch size is greater than number of training samples, switching to a batch size of %s\n" % n_pairs)
            batch_size = n_pairs
        self.positives = self._group_positives(list(train_data))
        budget = n_pairs * epochs                                        # triplets the reference pre-generates (:70, :89)
        sys.stderr.write("Generating %s random training samples\n" % budget)
        n_batches = (budget - 1) // batch_size if budget > 0 else 0      # the reference's loop condition, old/methods/bpr.py:72
        device = self._engine.device
        started = time.time()
        if n_batches > 0:
            csr = TrainingCSR(self.positives, sorted(self.positives), self.shape[0], device)
            self.losses = self._engine.run_batches(csr, n_batches, batch_size, want_loss=True)
        torch.cuda.synchronize(device)
        spent = time.time() - started
        if budget > 0:
            seen = n_batches * batch_size
            sys.stderr.write("Processed %s ( %.2f%% )\n" % (seen, 100.0 * seen / budget))
            sys.stderr.write("\nTotal training time %.2f seconds; %e per sample\n" % (spent, spent / budget))
            sys.stderr.flush()
