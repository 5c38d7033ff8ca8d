"""BPR matrix factorisation trained on MI355X (class API of the reference's single/bpr.py).

    model = BPR(k=50)
    model.load_training_data('data/uid', 'data/vid', 'data/f0tr.txt')
    model.train(epochs=5, batch_size=256, epoch_sample_limit=10**6)
    model.export_embeddings('embed/bpr')

Constructor, method names, arguments, public attributes and result arrays (``fue``, ``fie``,
``fib``) follow single/bpr.py:19-183.  What differs is where the work runs: the
(u,i,j) draw (bpr.py:155-165), the loss/gradient/RMSProp step (bpr.py:81-100,141) are HIP
kernels (csrc/sampler.hip, csrc/bpr_step.hip) on tables held in HBM.

Deliberate deviations from reference quirks:
  * ``epoch_sample_limit`` may be an integral float (train.py passes 10e5; bpr.py:111 asserts int)
  * an unknown ``sampling`` string raises ValueError (reference: TypeError on a None sampler)
  * the loss is evaluated with a stable softplus (identical wherever log(1+exp(-x)) is finite)
  * the checkpoint written by export_model is a torch file holding parameters + RMSProp slots
    (the reference writes a TF Saver checkpoint under the same name, ``weights``)
Extra keyword-only arguments (``seed``, ``device``, ``verbose``) default to reference behaviour.
"""
from __future__ import annotations

import os
import sys
import time
import warnings
from collections import defaultdict

import numpy as np
import torch

import textio
import tkr_hip
from utils import get_data_from_file, get_id_dict_from_file, tprint
from .rec import REC
from . import _engine


class BPR(REC):
    def __init__(self, k: int, lambda_u: float = 2.5e-3, lambda_i: float = 2.5e-3, lambda_j: float = 2.5e-4,
                 lambda_b: float = 0, lr: float = 1.0e-4, mode: str = 'l2') -> None:
        self.k = k
        self.lu, self.li, self.lj, self.lb = lambda_u, lambda_i, lambda_j, lambda_b
        self.lr = lr
        self.mode = mode
        self.uids = self.iids = self.data = None
        self.epoch_sample_limit = None
        self.n_users = self.n_items = None
        self.tr_data = self.tr_users = None
        self.pred = self.obj = self.solver = None      # TF graph handles in the reference; unused
        self.fue = self.fie = self.fib = None
        self._eng = None        # device tables (the reference's tf.Session + variables)
        self._csr = None        # device CSR of tr_data
        self._owned = None      # multi-GPU: global indices of the users whose rows this rank holds
        self._global_users = None
        self.last_epoch_loss = None

    # ------------------------------------------------------------------ data (bpr.py:51-69)
    def load_training_data(self, uid_file: str, iid_file: str, tr_file: str, data_copy: bool = False) -> None:
        tprint('Load training data from %s' % (tr_file))
        self.uids = get_id_dict_from_file(uid_file)
        self.iids = get_id_dict_from_file(iid_file)
        assert isinstance(self.uids, dict)
        assert isinstance(self.iids, dict)
        self.n_users = len(self.uids)
        assert self.n_users > 0
        self.n_items = len(self.iids)
        assert self.n_items > 0
        # One native pass over the ratings file (textio.parse_ratings) instead of the per-field Python loops of
        # utils.py:58-70 and bpr.py:167-171; the rules are theirs: known user, known item, like == 1, file order.
        if os.path.isfile(tr_file):
            R = textio.parse_ratings(tr_file, self.uids, self.iids)
            eu = R.entry_user
            keep = (eu >= 0) & (R.item >= 0) & (R.like == 1)
            u, it = eu[keep].astype(np.int64), R.item[keep]
        else:
            u, it = np.zeros(0, np.int64), np.zeros(0, np.int32)
        self.epoch_sample_limit = len(u)
        order = np.argsort(u, kind='stable')                       # grouped by user, file order inside a user
        us, its = u[order], it[order]
        uniq, start, counts = np.unique(us, return_index=True, return_counts=True)
        self.tr_data = defaultdict(list)
        for q in np.argsort(order[start], kind='stable'):          # dict key order = first appearance in the file
            self.tr_data[int(uniq[q])] = its[start[q]:start[q] + counts[q]].tolist()
        self.tr_users = list(self.tr_data.keys())
        row_ptr = np.zeros(self.n_users + 1, dtype=np.int64)
        if len(uniq):
            row_ptr[uniq + 1] = counts
        np.cumsum(row_ptr, out=row_ptr)
        self._csr_arrays = (row_ptr, its.astype(np.int32), np.asarray(self.tr_users, dtype=np.int32))
        self._csr = None
        if data_copy:                                              # the (uid, iid) token pairs themselves: utils.py:58-70
            self.data = get_data_from_file(tr_file, self.uids, self.iids)
            assert len(self.data) == self.epoch_sample_limit
        elif hasattr(self, 'data'):
            del self.data                                          # bpr.py:67-68
        tprint('Loading finished!')

    def _make_csr(self, tr_users, device):
        """device CSR of the training positives for the listed users (all of them, or one shard)"""
        arrays = getattr(self, '_csr_arrays', None)
        if arrays is not None and arrays[0][-1] == sum(len(v) for v in self.tr_data.values()):
            return _engine.TrainingCSR.from_arrays(arrays[0], arrays[1], np.asarray(list(tr_users), dtype=np.int32), device)
        return _engine.TrainingCSR(self.tr_data, tr_users, self.n_users, device)     # tr_data assigned by hand

    def _data_to_training_dict(self, data: list, users: dict, items: dict):
        """user index -> item indices in file order, duplicates preserved (bpr.py:167-171)."""
        grouped = defaultdict(list)
        for uid, iid in data:
            grouped[users[uid]].append(items[iid])
        return grouped

    # ------------------------------------------------------------------ model (bpr.py:71-101)
    def _hyper(self):
        return dict(lu=self.lu, li=self.li, lj=self.lj, lb=self.lb, lr=self.lr, mode=self.mode)

    def _make_engine(self, device, seed, n_users=None, user_seed=None):
        return _engine.BprEngine(self.n_users if n_users is None else n_users, self.n_items, self.k, self._hyper(), device, seed, user_seed=user_seed)

    def build_graph(self, *, device=None, seed=None, owned=None):
        """Allocate and initialise the model tables in HBM (user_embed / item_embed ~ N(0, 0.01),
        item_bias = 0, RMSProp slots = 1; bpr.py:77-79,100).  The reference returns the three
        feed placeholders; there is nothing to feed here, so the engine is returned instead.

        ``owned`` (multi-GPU): the global indices of the users this rank trains.  Only THEIR rows are allocated -- local
        row q is user owned[q], the training CSR is re-indexed to match -- while the item-side tables are full replicas."""
        self._owned = None if owned is None else np.asarray(owned, dtype=np.int64)
        if self._owned is None:
            self._eng = self._make_engine(device, seed)
            if self._csr is None or self._csr.row_ptr.device != self._eng.device or getattr(self._csr, 'local', False):
                self._csr = self._make_csr(self.tr_users, self._eng.device)
            return self._eng
        import dist as tdist
        rank, _ = tdist.world()
        self._eng = self._make_engine(device, seed, n_users=len(self._owned), user_seed=(0 if seed is None else seed) * 1000003 + 7919 * (rank + 1))
        self._csr = self._make_local_csr(self._owned, self._eng.device)
        return self._eng

    def _make_local_csr(self, owned, device):
        """training CSR of a user shard in the shard's own row numbering (row q = user owned[q])"""
        row_ptr, pos, _ = self._csr_arrays if getattr(self, '_csr_arrays', None) is not None else (None, None, None)
        if row_ptr is None or row_ptr[-1] != sum(len(v) for v in self.tr_data.values()):      # tr_data assigned by hand
            row_ptr = np.zeros(self.n_users + 1, dtype=np.int64)
            for u, items in self.tr_data.items():
                row_ptr[u + 1] = len(items)
            np.cumsum(row_ptr, out=row_ptr)
            pos = np.fromiter((it for u in sorted(self.tr_data) for it in self.tr_data[u]), dtype=np.int32, count=int(row_ptr[-1]))
        return _engine.TrainingCSR.shard(row_ptr, pos, owned, device)

    def _user_rows(self, table):
        """the rows of a [n_users, ...] host array this rank's engine holds (all of them in a single process)"""
        table = np.asarray(table)
        return table if getattr(self, '_owned', None) is None else np.ascontiguousarray(table[self._owned])

    # ------------------------------------------------------------------ train (bpr.py:103-153)
    def _warm_start(self):
        """bpr.py:127-135: text-imported (or previous) fue/fie/fib override the fresh init."""
        if self.fue is not None:
            tprint('Initialize user embeddings')
            self._eng.set_users(U=self._user_rows(self.fue))
        if self.fie is not None:
            tprint('Initialize item embeddings')
            self._eng.set_items(V=self.fie)
        if self.fib is not None:
            tprint('Initialize item biases')
            self._eng.set_items(b=np.asarray(self.fib).ravel())

    def _collect_users(self, width):
        """fue of the whole model.  One process: the engine's table.  Sharded: every rank contributes the rows it owns
        (parameters and slots, one all-gather each -- no rank ever held or exchanged the full table during training); users
        nobody trains keep their start: the warm-start value, else N(0, 0.01) drawn from the shared seed."""
        p, ms = self._eng.get('U')
        if getattr(self, '_owned', None) is None:
            self._global_users = None
            return p.cpu().numpy()
        import dist as tdist
        if self.fue is not None and np.asarray(self.fue).shape == (self.n_users, width):
            base = np.array(self.fue, dtype=np.float32)
        else:
            base = (np.random.Generator(np.random.PCG64(self._eng.seed)).standard_normal((self.n_users, width)) * 0.01).astype(np.float32)
        base_ms = np.ones_like(base)
        old = getattr(self, '_global_users', None)
        if old is not None and old[1].shape == base_ms.shape:      # slots of users nobody trains: what the checkpoint / previous train() left
            base_ms = old[1].numpy().copy()                # (a table of another shape -- k or the user list changed between train() calls -- is
                                                           # dropped: ADVICE r3, an assert here lost a whole sharded run at its very end)
        elif old is not None:
            warnings.warn('the saved RMSProp slots of the users this run does not train have shape %s, the model now %s: they start from 1.0 again'
                          % (tuple(old[1].shape), base_ms.shape))
        for ids, rows, slots in tdist.gather_owned_rows(self._owned, p, ms):
            base[ids], base_ms[ids] = rows, slots
        self._global_users = (torch.from_numpy(base.copy()), torch.from_numpy(base_ms))       # what export_model writes
        return base

    def _collect(self):
        """bpr.py:151-153"""
        self.fue = self._collect_users(self.k)
        self.fie = self._eng.get('V')[0].cpu().numpy()
        self.fib = self._eng.get('b')[0].reshape(-1, 1).cpu().numpy()

    def train(self, sampling: str = 'user uniform', epochs: int = 5, batch_size: int = 256,
              epoch_sample_limit: int = None, model_path: str = None, *, seed=None, device=None,
              verbose: bool = True, streams: int = 1):
        assert isinstance(sampling, str)
        assert isinstance(epochs, int)
        assert isinstance(batch_size, int)
        if epoch_sample_limit is not None:
            assert float(epoch_sample_limit) == int(epoch_sample_limit), 'epoch_sample_limit must be integral'
            self.epoch_sample_limit = int(epoch_sample_limit)
        if sampling != 'user uniform':
            raise ValueError("unknown sampling %r (only 'user uniform' exists, bpr.py:115-117)" % sampling)
        batch_limit = self.epoch_sample_limit // batch_size + 1
        n_batches = batch_limit - 1                      # bno runs 1 .. batch_limit-1 (bpr.py:138-147)
        if n_batches < 1:
            raise ValueError('epoch_sample_limit < batch_size: the reference loop would never terminate')
        import dist as tdist
        rank, world = tdist.world()
        seed = tdist.shared_seed(seed)                   # one init and one sample-stream key for every rank
        sharded = world > 1 and streams == 1
        if sharded and world > len(self.tr_users):         # the same error on EVERY rank (a rank without users would leave the others
            raise ValueError('%d ranks but only %d users with training positives: every rank needs at least one'      # in the collectives)
                             % (world, len(self.tr_users)))
        self.build_graph(device=device, seed=seed, owned=tdist.shard_users(self.tr_users, rank, world) if sharded else None)
        if model_path is not None:
            assert isinstance(model_path, str)
            tprint("Initialize weights with the previous trained model")
            self.import_embeddings(model_path)
        tprint('Training parameters: lu=%.6f, li=%.6f, lj=%.6f, lb=%.6f' % (self.lu, self.li, self.lj, self.lb))
        tprint('Learning rate is %.6f, regularization mode is %s' % (self.lr, self.mode))
        tprint('Training for %d epochs of %d batches using %s sampler' % (epochs, batch_limit, sampling))
        self._warm_start()
        if world > 1:                                      # ranks packed on one GPU split its CUs between their K2o launches
            self._eng.ranks_on_device = tdist.ranks_sharing_device(self._eng.device)
        self._eng.prepare(batch_size)                      # table layout of this batch size (see _engine.BprEngine)
        # one process per GPU (torch.distributed initialised by the launcher): users sharded, item-side
        # tables replicated and reconciled once per epoch (dist.py; the reference is single-process)
        if streams > 1:
            # opt-in: the multi-GPU layout inside ONE GPU -- `streams` user shards with replicated item tables run
            # concurrently on separate HIP streams, each through the persistent step of its batch size on its share of the
            # CUs, and are reconciled once per epoch by the same rule (dist.LocalShards).  One sequential stream is bound
            # by the hand-offs of its dependency chain and leaves most of the chip idle (bench.py shards_on_one_gpu).  Not
            # the reference's single-stream semantics: off by default.
            assert world == 1, 'streams > 1 and torch.distributed sharding are not combined'
            self._train_streams(epochs, n_batches, batch_size, streams, verbose)
            self._collect()
            return
        if world > 1:
            # each rank owns the rows of its users (build_graph allocated only those and re-indexed the CSR); the item-side
            # tables are replicas, reconciled once per epoch; the user rows are gathered once, after training (_collect)
            n_batches = tdist.batches_per_rank(n_batches, world)
            self._eng.triplets_drawn = rank * epochs * n_batches * batch_size      # disjoint stream positions
            tdist.assert_replicated(self._eng)             # same seed, same warm start: the replicas must start equal
        # A persistent step (K2f / K2o) that cannot get its workgroups resident -- a GPU shared with something that never ends --
        # gives up after a bounded spin and leaves half-updated tables; the engine then steps down one kernel (K2o -> K2f -> K2).
        # The run starts again from the state kept here, on the same counter-based sample stream: it degrades instead of raising.
        for attempt in range(3):
            # (VBPR steps one launch at a time: nothing to give up; TKR_RESTART=0: no copy of the tables, a step that gives up raises)
            start = self._eng.snapshot() if (getattr(self._eng, 'layout', None) == 'flow' and self._eng.cfg.restart) else None
            if self._train_epochs(epochs, n_batches, batch_size, world, verbose):
                break
            if start is None or attempt == 2:
                raise tkr_hip.StepGaveUp('BPR.train: the step gave up on every kernel form')
            self._eng.restore(start)
            self._eng.prepare(batch_size)                  # (the table layout of the kernel the engine stepped down to)
        self._collect()

    def _train_epochs(self, epochs, n_batches, batch_size, world, verbose):
        """the epoch loop of bpr.py:136-150; False = a persistent step gave up on some rank (every rank returns False then)"""
        import dist as tdist
        sync = tdist.ItemSync(self._eng) if world > 1 else None
        for eid in range(epochs):
            t0 = time.time()
            if sync is not None:
                sync.begin()
            # sharded: the exchange follows this call, then an epoch of n_batches more -- its first chunk is planned behind this
            # epoch's last steps (PlanMixin), and the host looks at the loss only after the exchange is queued
            gave_up = mine = False
            try:
                loss = self._run_epoch(n_batches, batch_size, n_batches if (sync is not None and eid + 1 < epochs) else 0, defer=sync is not None)
                if sync is not None:
                    sync.end()
                    loss = self._epoch_loss(loss)
            except tkr_hip.StepGaveUp as e:
                warnings.warn('BPR.train restarts from its initial state: %s' % e)
                gave_up = mine = True
            if sync is not None:
                gave_up = sync.any_gave_up(gave_up)          # the flag rode in the exchange: every rank agrees, no extra collective
            if gave_up:
                if not mine and hasattr(self._eng, 'step_down'):
                    self._eng.step_down()                    # in lockstep with the rank whose step gave up
                return False
            torch.cuda.synchronize(self._eng.device)
            spent = time.time() - t0
            self.last_epoch_loss = loss
            if verbose:
                sys.stderr.write('\rEpoch=%3d, batch=%6d, loss=%8.4f, time=%4.4fs' % (eid + 1, n_batches, loss, spent / n_batches))
                sys.stderr.write(' ... total time collapse %8.4fs' % spent)
                sys.stderr.flush()
                print()
        return True

    def _train_streams(self, epochs, n_batches, batch_size, S, verbose):
        """the multi-GPU layout inside ONE GPU: S user shards with replicated item tables, each a persistent step of its own on
        CUs // S owners (K2o; K2f / K2 where the layout asks for them) on its own HIP stream, reconciled once per epoch by
        dist.LocalShards -- pack, sum, unpack: the exchange of the sharded loop (bpr.py:136-147) without the collective."""
        import dist as tdist
        dev = self._eng.device
        engines = [self._eng] + [self._make_engine(dev, self._eng.seed) for _ in range(S - 1)]
        lead = engines[0]
        for e in engines:
            e.ranks_on_device = S                          # the CUs are split between the shards' launches
            # ... through K2o down to half the CUs per shard; on fewer owners a shard's K2o is slower than K2f beside the other shards
            # (ML-10M shape, aggregate: S = 2: K2o 135 / K2f 139 M triplets/s; S = 3: K2o on 85 owners each 91 M, K2f 145 M)
            e.own_min_owners = torch.cuda.get_device_properties(dev).multi_processor_count // 2
            e.private_side_stream = True                   # ... and every shard plans on a stream of its own --
            # unless that takes the process past HIP's four hardware queues while the step streams alone fit them (S = 2, 3): queues are
            # in-order, two shards' planner streams on one queue made the second shard's launches wait for the first shard's epoch
            # (S = 2: 105 -> 133 M triplets/s with K1 in order on the shard's stream; S = 4 / 8: 127 / 125 -> 123 / 123, kept as they were)
            e.plan_in_order = S + 1 <= 4 < 2 * S + 1
            e.prepare(batch_size)
        for e in engines[1:]:                              # every shard starts from the same (possibly warm-started) model
            e.copy_model_from(lead)
        nb = tdist.batches_per_rank(n_batches, S)
        csrs, hip_streams = [], []
        for i, e in enumerate(engines):
            csrs.append(self._make_csr(tdist.shard_users(self.tr_users, i, S), dev))
            e.triplets_drawn = i * epochs * nb * batch_size                      # disjoint stream positions, one key
            hip_streams.append(torch.cuda.Stream(device=dev))
        users_start, users_ms_start = (t.clone() for t in lead.get('U'))
        shards = tdist.LocalShards(engines, hip_streams)
        self._shards = shards
        torch.cuda.synchronize(dev)
        for eid in range(epochs):
            t0 = time.time()
            shards.begin()
            losses = []
            for e, csr, st in zip(engines, csrs, hip_streams):                  # S persistent launches side by side
                with torch.cuda.stream(st):
                    losses.append(e.run_batches(csr, nb, batch_size, want_loss=True, then_exchange=nb if eid + 1 < epochs else 0)[-1:].clone())
            shards.end()
            torch.cuda.synchronize(dev)
            if shards.any_gave_up():
                raise tkr_hip.StepGaveUp('BPR.train(streams=%d): a persistent step gave up on one of the shards' % S)
            spent = time.time() - t0
            self.last_epoch_loss = float(losses[0][0])
            if verbose:
                sys.stderr.write('\rEpoch=%3d, batch=%6d, loss=%8.4f, time=%4.4fs' % (eid + 1, nb * S, self.last_epoch_loss, spent / (nb * S)))
                sys.stderr.write(' ... total time collapse %8.4fs' % spent)
                sys.stderr.flush()
                print()
        parts = [e.get('U') for e in engines]               # every user row was changed by at most one shard
        lead.set_users(U=users_start + sum(p - users_start for p, _ in parts),
                       msU=users_ms_start + sum(ms - users_ms_start for _, ms in parts))     # slots may come from a checkpoint, not 1

    def _run_epoch(self, n_batches, batch_size, then_exchange=0, defer=False):
        losses = self._eng.run_batches(self._csr, n_batches, batch_size, want_loss=True, then_exchange=then_exchange)
        if defer:
            return losses[-1:].clone()         # read by _epoch_loss once the exchange is queued behind the steps
        return self._epoch_loss(losses[-1:])

    def _epoch_loss(self, last):
        last = float(last[0])
        self._eng.check()
        return last

    # ------------------------------------------------------------------ sampler (bpr.py:155-165)
    def _uniform_user_sampling(self, batch_size: int):
        """Generator of (ub, ib, jb) numpy batches drawn by the device sampler (K1): u uniform over
        tr_users, i uniform over tr_data[u], j uniform over items not in tr_data[u]."""
        device = self._eng.device if self._eng is not None else _engine.default_device()
        if self._csr is None:
            self._csr = self._make_csr(self.tr_users, device)
        import tkr_hip
        plan = _engine.PlanBuffers(1, batch_size, device)
        cnt = _engine.UpdateCounters(self.n_users, self.n_items, device)     # throw-away parities
        seed = int(np.random.SeedSequence().entropy % (2 ** 63)) if self._eng is None else self._eng.seed
        drawn = 0
        while True:
            tkr_hip.sample_plan(self._csr, self.n_users, self.n_items, seed ^ 0x5DEECE66D, drawn, 1, batch_size, cnt, plan)
            drawn += batch_size
            yield plan.u.cpu().numpy().astype(np.int64), plan.i.cpu().numpy(), plan.j.cpu().numpy()

    # ------------------------------------------------------------------ checkpoint (bpr.py:173-183)
    def _checkpoint_tensors(self):
        e = self._eng
        out = {}
        for name in ('U', 'V', 'b'):
            p, ms = e.get(name)
            out[name], out['ms_' + name] = p.cpu(), ms.cpu()
        if getattr(self, '_global_users', None) is not None:           # sharded run: the gathered table, not this rank's rows
            out['U'], out['ms_U'] = self._global_users
        return out

    def _restore_tensors(self, blob):
        e = self._eng
        e.set_users(U=self._user_rows(blob['U']), msU=self._user_rows(blob['ms_U']))
        if getattr(self, '_owned', None) is not None:
            self._global_users = (torch.as_tensor(np.asarray(blob['U'])).clone(), torch.as_tensor(np.asarray(blob['ms_U'])).clone())
        e.set_items(V=blob['V'], b=blob['b'], msV=blob['ms_V'], msb=blob['ms_b'])

    def import_model(self, model_path: str) -> None:
        file_path = os.path.join(model_path, 'weights')
        if os.path.exists(file_path) and self._eng is not None:
            tprint('Restoring model tables from path %s' % (file_path))
            self._restore_tensors(torch.load(file_path, map_location='cpu'))

    def export_model(self, model_path: str) -> None:
        if os.path.exists(model_path) and self._eng is not None:
            file_path = os.path.join(model_path, 'weights')
            tprint('Saving model tables to path %s' % (file_path))
            torch.save(self._checkpoint_tensors(), file_path)
