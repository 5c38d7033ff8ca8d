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
from collections import defaultdict

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

    def __init__(self, K, users, items, lambda_u=0.0025, lambda_i=0.0025, lambda_j=0.00025, lambda_bias=0.0,
                 learning_rate=1.0e-4, *, seed=None, device=None):
        self._K = K
        self._train_users = users
        self._train_items = items
        self._n_users = len(users)
        self._n_items = len(items)
        self._lambda_u = lambda_u
        self._lambda_i = lambda_i
        self._lambda_j = lambda_j
        self._lambda_bias = lambda_bias
        self._learning_rate = learning_rate
        self._train_dict = {}
        hp = dict(lu=lambda_u, li=lambda_i, lj=lambda_j, lb=lambda_bias, lr=learning_rate, mode='l2', opt='sgd')
        self._engine = BprEngine(self._n_users, self._n_items, K, hp, device=device, seed=seed)
        self.W, self.H, self.B = _Shared(self, 'U'), _Shared(self, 'V'), _Shared(self, 'b')
        self.losses = None

    def train(self, train_data, epochs=30, batch_size=256):
        if len(train_data) < batch_size:
            sys.stderr.write("WARNING: Batch size is greater than number of training samples, switching to a batch size of %s\n" % str(len(train_data)))
            batch_size = len(train_data)
        self._train_dict = self._data_to_dict(train_data, self._train_users, self._train_items)
        n_sgd_samples = len(train_data) * epochs
        sys.stderr.write("Generating %s random training samples\n" % str(n_sgd_samples))
        n_batches = (n_sgd_samples - 1) // batch_size if n_sgd_samples > 0 else 0      # old/methods/bpr.py:72
        csr = TrainingCSR(self._train_dict, sorted(self._train_dict.keys()), self._n_users, self._engine.device)
        t0 = time.time()
        done = n_batches
        if n_batches > 0:
            self.losses = self._engine.run_batches(csr, n_batches, batch_size, want_loss=True)
        torch.cuda.synchronize(self._engine.device)
        t2 = time.time()
        if n_sgd_samples > 0:
            sys.stderr.write("Processed %s ( %.2f%% )\n" % (str(done * batch_size), 100.0 * float(done * batch_size) / n_sgd_samples))
            sys.stderr.write("\nTotal training time %.2f seconds; %e per sample\n" % (t2 - t0, (t2 - t0) / n_sgd_samples))
            sys.stderr.flush()

    def _data_to_dict(self, data, users, items):
        data_dict = defaultdict(list)
        for (user, item) in data:
            data_dict[users[user]].append(items[item])
        return data_dict
