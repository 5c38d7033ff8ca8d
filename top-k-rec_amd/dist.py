"""Multi-GPU layout of the BPR/VBPR path: one process per GPU, users sharded, item-side state
replicated and reconciled by ONE all-reduce per epoch (RCCL over xGMI; backend 'nccl' on ROCm,
'gloo' in the CPU tests).  The reference is single-process; this is new design.

Reduction rule: parameters  P <- P0 + sum_g (P_g - P0)   (sum of per-replica deltas:
tracks the single-stream trajectory to first order, whereas a plain mean divides the item
displacement by the world size); RMSProp slots  ms <- mean_g ms_g.
Each rank allocates only the rows of the users it owns; they are gathered once, after training.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def ranks_sharing_device(device) -> int:
    """how many ranks of the process group run on the SAME physical GPU as this one (1 without a process group).  Decided from
    the device's identity -- host name + PCI bus id / uuid, all-gathered -- not from device_count(): a launcher that shows every
    rank one GPU (HIP_VISIBLE_DEVICES, SLURM gpus-per-task) and a multi-node job both make world_size > device_count() with one
    rank per GPU.  A collective: every rank calls it at the same point (BPR.train / bench.py do, before the first step)."""
    rank, w = world()
    if w == 1:
        return 1
    import socket
    ident = 'cpu'
    if device is not None and getattr(device, 'type', 'cpu') == 'cuda':
        prop = torch.cuda.get_device_properties(device)
        import os
        ident = os.environ.get('TKR_DEVICE_IDENTITY', '') or str(getattr(prop, 'uuid', '')) or ''
        if not ident or set(ident) <= set('0-'):          # no uuid reported: the PCI address is as good
            if hasattr(prop, 'pci_bus_id'):
                ident = '%s:%s:%s' % (getattr(prop, 'pci_domain_id', 0), prop.pci_bus_id, getattr(prop, 'pci_device_id', 0))
            else:
                # neither a uuid nor a PCI address: every rank behind HIP_VISIBLE_DEVICES would report index 0 and all ranks
                # of a host would count as sharing one GPU.  An unknown identity is NOT shared: rank-unique (TKR_DEVICE_IDENTITY overrides)
                ident = 'unknown-rank-%d' % rank
    mine = (socket.gethostname(), ident)
    everyone = [None] * w
    dist.all_gather_object(everyone, mine)
    return sum(1 for x in everyone if x == mine)


def shared_seed(seed):
    """the seed every rank uses for the model init and as the key of the sample stream: rank 0's (drawn there when the
    caller gave none, like the single-process default) broadcast to all.  Ranks that initialised differently would
    never agree again: the per-epoch exchange moves deltas, not values."""
    rank, w = world()
    if w == 1:
        return seed
    box = [int(seed) if seed is not None else int(np.random.SeedSequence().entropy % (2 ** 63))]
    dist.broadcast_object_list(box, src=0)
    return box[0]


def assert_replicated(engine, names=None):
    """every rank holds bit-identical replicated tables (checked through a min/max all-reduce of two checksums each)"""
    _, w = world()
    if w == 1:
        return
    sums = []
    for n in (names or engine.replicated_names):
        p = engine.get(n)[0].double()
        sums += [p.sum(), (p * p).sum()]
    lo = torch.stack(sums)
    hi = lo.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    if not torch.equal(lo, hi):
        raise RuntimeError('replicated model tables differ between ranks before training (different seed or warm start)')


def shard_users(tr_users, rank: int, world_size: int):
    """deal tr_users round-robin: every shard keeps the uniform-over-users marginal"""
    return list(tr_users)[rank::world_size]


def batches_per_rank(n_batches: int, world_size: int) -> int:
    """equal batch counts per rank (the remainder is dropped, like the reference drops limit % B)"""
    return max(1, n_batches // world_size)


def reduce_deltas(current: torch.Tensor, start: torch.Tensor, group=None) -> torch.Tensor:
    """P0 + sum over ranks of (P_g - P0)"""
    delta = current - start
    dist.all_reduce(delta, op=dist.ReduceOp.SUM, group=group)
    return start + delta


def reduce_mean(t: torch.Tensor, group=None) -> torch.Tensor:
    out = t.clone()
    dist.all_reduce(out, op=dist.ReduceOp.SUM, group=group)
    return out / dist.get_world_size(group)


class ItemSync:
    """Per-epoch exchange of the replicated item-side tables of an engine.

    ``names`` are engine table names updated by every rank (BPR: V, b; VBPR adds the dense
    content tables).  ``begin()`` snapshots, ``end()`` all-reduces and writes back.

    On the GPU the snapshot / pack / unpack are one HIP launch per table (csrc/sync.hip) around ONE collective
    on a flat buffer: xGMI all-reduces of a few MB are latency-bound, and ~25 framework ops per exchange cost
    240 us against a 2.1 ms epoch at 8 GPUs.  Engines without ``replicated_tables`` (the CPU stand-ins of the gloo
    tests) take the same arithmetic through plain tensor ops; the granule tables of the dataflow step (batch <= 512) have their
    own fused kernels (tkr_sync_flow_*: 464 -> ~60 us per exchange at the ML-10M shape, against a 1.25 ms epoch at 8 GPUs)."""

    def __init__(self, engine, names=None):
        self.eng = engine
        self.names = tuple(names or getattr(engine, 'replicated_names', ('V', 'b')))
        self.start = None
        self.tabs = None
        self._bound = None
        self._start_valid = None     # (binding, engine.item_mutations) for which start_flat already holds the epoch's start
        self.timing = None           # a list: end() appends (before pack, after pack, after collective, after unpack) timing events
        self._bind()

    def _bind(self):
        """(re-)attach to the engine's tables: an engine re-allocates them when it changes layout (layout_epoch)"""
        epoch = getattr(self.eng, 'layout_epoch', 0)
        if self._bound == epoch:
            return
        self._bound = epoch
        self.tabs = None
        self.flow = None
        flow = getattr(self.eng, 'flow_sync_tables', None)
        flow = flow() if flow is not None and self.names == ('V', 'b') else None
        if flow is not None and flow[0].is_cuda:                # granule layout of the dataflow step: its own fused kernels
            self.flow = flow
            total = flow[5] * (flow[6] + 1)                     # n_items * k factors + n_items biases
            self.start_flat = torch.empty(total, dtype=torch.float32, device=flow[0].device)
            self.flat = torch.zeros(2 * total + 1, dtype=torch.float32, device=flow[0].device)     # + the gave-up flag (any_gave_up)
            return
        tables = getattr(self.eng, 'replicated_tables', None)
        if tables is not None:
            tabs = [t for t in tables() if t[0] in self.names]
            if tabs and all(t[1].is_cuda for t in tabs):
                self.tabs = tabs
                self.sizes = [int(t[1].numel() // (2 if t[3] is not None else 1)) for t in tabs]
                total = sum(self.sizes)
                dev = tabs[0][1].device
                self.start_flat = torch.empty(total, dtype=torch.float32, device=dev)
                self.flat = torch.zeros(2 * total + 1, dtype=torch.float32, device=dev)

    @staticmethod
    def _shape(P, cnt):
        n = P.shape[1] if cnt is not None else P.shape[0]
        return int(n), int(P.numel() // (2 if cnt is not None else 1) // n)

    def _settle(self, keep_epoch_ahead=False):
        """the counters this object holds must describe the tables: drop planned-but-not-run batches (PlanMixin.settle).
        ``keep_epoch_ahead``: not the chunk an engine planned AHEAD of this exchange for the epoch after it (its item counts live
        on a shadow until after_exchange(); afterwards they describe exactly the freshly re-assigned tables)."""
        settle = getattr(self.eng, 'settle', None)
        if settle is not None:
            if keep_epoch_ahead:
                settle(check=False, keep_epoch_ahead=True)
            else:
                settle(check=False)
            self.eng.check_async()       # a failed persistent step surfaces at the next exchange (or at the final get): no host wait here

    def begin(self):
        # no snapshot launch is needed when the unpack of the exchange before left this epoch's start in start_flat: then nothing here
        # reads the item counters, and a chunk planned ahead of that exchange may stay
        fresh = (self.flow is not None and self._bound == getattr(self.eng, 'layout_epoch', 0) and
                 self._start_valid == (self._bound, getattr(self.eng, 'item_mutations', 0)))
        self._settle(keep_epoch_ahead=fresh)
        self._bind()
        if self.flow is not None:
            import tkr_hip
            V, msV, tail, rd, icnt, n, k, bufs = self.flow
            # the unpack of the exchange before left the new values in start_flat: they ARE this epoch's start, unless somebody wrote
            # the item tables since (set_items, a layout change: the engine counts those)
            if self._start_valid != (self._bound, getattr(self.eng, 'item_mutations', 0)):
                tkr_hip.sync_flow_snapshot(V, tail, icnt, self.start_flat, n, k, bufs)
            self._start_valid = None
            self.start = True
            return
        if self.tabs is None:
            self.start = {n: self.eng.get(n)[0].clone() for n in self.names}
            return
        import tkr_hip
        off = 0
        for (_, P, _, cnt), size in zip(self.tabs, self.sizes):
            n, w = self._shape(P, cnt)
            tkr_hip.sync_snapshot(P, cnt, self.start_flat[off:off + size], n, w)
            off += size
        self.start = True

    def _mark(self, marks):
        if self.timing is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            marks.append(ev)
            if len(marks) == 4:
                self.timing.append(tuple(marks))

    def _flag_status(self, total):
        """the last word of the exchanged vector: did a persistent step of THIS rank give up during the epoch (the engine's device
        status word)?  Summed by the same all-reduce, so that every rank learns it without a collective of its own."""
        ctl = getattr(self.eng, 'ctl', None)
        if ctl is not None and getattr(self.eng, 'layout', None) == 'flow':
            import tkr_hip
            self.flat[2 * total:] = (ctl[tkr_hip.FLOW_CTL_STATUS:tkr_hip.FLOW_CTL_STATUS + 1] != 0).float()
        else:
            self.flat[2 * total:].zero_()
        self._flagged = True

    def any_gave_up(self, mine=False):
        """after end(): True on EVERY rank if a persistent step gave up on ANY rank during the epoch just exchanged (host wait)"""
        if getattr(self, '_flagged', False):
            self._flagged = False
            return bool(mine) or float(self.flat[-1]) > 0.0
        return bool(mine)

    def end(self):
        _, w = world()
        if w == 1:
            return
        marks = []
        self._mark(marks)
        self._pack(w)
        self._mark(marks)
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        self._mark(marks)
        self._unpack(self.flat, w)
        self._mark(marks)

    def _pack(self, w):
        """first third of an exchange: this shard's deltas and its share of the slot means into ``self.flat`` (w = number of shards)"""
        self._settle(keep_epoch_ahead=True)
        if getattr(self.eng, 'layout_epoch', 0) != self._bound:
            # the engine re-allocated its tables after begin() (a different batch size -> a different layout): the snapshot and the
            # counters this object holds belong to tables that no longer exist.  Round 2's bench.py did exactly that in its warm-up
            # (begin() before the first run_batches) and zeroed the live update counters through the stale binding.
            raise RuntimeError('ItemSync.end(): the engine changed its table layout since begin(); call engine.prepare(batch_size) '
                               'before the first begin()')
        if self.flow is not None:
            import tkr_hip
            V, msV, tail, rd, icnt, n, k, bufs = self.flow
            total = n * (k + 1)
            tkr_hip.sync_flow_pack(V, msV, tail, icnt, self.start_flat, self.flat[:total], self.flat[total:], n, k, 1.0 / w, bufs)
            self._flag_status(total)
            return
        if self.tabs is None:
            self._cur = {n: self.eng.get(n) for n in self.names}
            parts = []
            for n in self.names:
                p, ms = self._cur[n]
                parts.append((p - self.start[n]).reshape(-1))
                parts.append((ms / w).reshape(-1))
            self.flat = torch.cat(parts)
            return
        import tkr_hip
        total, off = sum(self.sizes), 0
        for (_, P, ms, cnt), size in zip(self.tabs, self.sizes):
            n, wd = self._shape(P, cnt)
            tkr_hip.sync_pack(P, ms, cnt, self.start_flat[off:off + size], self.flat[off:off + size],
                              self.flat[total + off:total + off + size], n, wd, 1.0 / w)
            off += size
        self._flag_status(total)

    def _unpack(self, flat, w):
        """last third: the summed vector (``flat``: this object's own buffer after an all-reduce, or the sum LocalShards formed)
        back into the tables"""
        if self.flow is not None:
            import tkr_hip
            V, msV, tail, rd, icnt, n, k, bufs = self.flow
            total = n * (k + 1)
            tkr_hip.sync_flow_unpack(V, msV, tail, rd, icnt, self.start_flat, flat[:total], flat[total:2 * total], n, k, bufs)
            self._start_valid = (self._bound, getattr(self.eng, 'item_mutations', 0))
            after = getattr(self.eng, 'after_exchange', None)
            if after is not None:
                after()                  # a chunk planned ahead of this exchange becomes runnable
            return
        if self.tabs is None:
            new, off = {}, 0
            for n in self.names:
                p, ms = self._cur[n]
                m = p.numel()
                new[n] = (self.start[n] + flat[off:off + m].view_as(p), flat[off + m:off + 2 * m].view_as(ms))
                off += 2 * m
            self.eng.set_replicated(new)
            return
        import tkr_hip
        total, off = sum(self.sizes), 0
        for (_, P, ms, cnt), size in zip(self.tabs, self.sizes):
            n, wd = self._shape(P, cnt)
            tkr_hip.sync_unpack(P, ms, self.start_flat[off:off + size], flat[off:off + size],
                                flat[total + off:total + off + size], n, wd)
            off += size
        for cnt in {id(t[3]): t[3] for t in self.tabs if t[3] is not None}.values():
            cnt.zero_()                                       # buffer 0 is current again for every row


class LocalShards:
    """The per-epoch exchange of S user shards that live in ONE process on ONE GPU (BPR.train(streams=S), bench.py's
    shards_on_one_gpu leg): the same rule as between ranks -- P <- P0 + sum of deltas, slots <- mean -- with the collective
    replaced by a sum of the shards' packed vectors.  Every shard packs and unpacks on its own HIP stream; the sum runs on the
    calling stream between two joins.  A rehearsal of the 8-GPU run on one device: pack + sum + unpack is everything an epoch
    boundary costs apart from the collective itself."""

    def __init__(self, engines, streams):
        self.syncs = [ItemSync(e) for e in engines]
        self.streams = list(streams)
        self.acc = None
        self.timing = None               # a list: end() appends (start, packed, summed, unpacked) events of the calling stream

    def begin(self):
        for s, st in zip(self.syncs, self.streams):
            with torch.cuda.stream(st):
                s.begin()

    def end(self):
        S = len(self.syncs)
        main = torch.cuda.current_stream()
        ev = []

        def mark():
            if self.timing is not None:
                e = torch.cuda.Event(enable_timing=True)
                e.record(main)
                ev.append(e)
        for st in self.streams:          # (the timing brackets the exchange alone: the epochs' steps are done)
            main.wait_stream(st)
        mark()
        for s, st in zip(self.syncs, self.streams):
            with torch.cuda.stream(st):
                st.wait_stream(main)
                s._pack(S)
        for st in self.streams:
            main.wait_stream(st)
        mark()
        flats = [s.flat for s in self.syncs]
        if self.acc is None or self.acc.shape != flats[0].shape:
            self.acc = torch.empty_like(flats[0])
        torch.add(flats[0], flats[1], out=self.acc) if S > 1 else self.acc.copy_(flats[0])
        for f in flats[2:]:
            self.acc.add_(f)
        mark()
        for s, st in zip(self.syncs, self.streams):
            with torch.cuda.stream(st):
                st.wait_stream(main)
                s._unpack(self.acc, S)
        for st in self.streams:
            main.wait_stream(st)
        mark()
        if self.timing is not None:
            self.timing.append(tuple(ev))

    def any_gave_up(self, mine=False):
        return bool(mine) or (self.acc is not None and float(self.acc[-1]) > 0.0)


def gather_owned_rows(owned, rows: torch.Tensor, slots: torch.Tensor):
    """After training: every rank contributes the user rows it owns.  -> [(global ids, rows, slots) per rank] as numpy arrays
    on every rank.  One all-gather of the ids and one of [rows | slots]; shards differ in length by at most one user, the
    shorter ones are padded.  (Round 1 kept the FULL user table on every rank and all-reduced it: 246 MB at the Netflix shape.)"""
    _, w = world()
    n, width = int(rows.shape[0]), int(rows.shape[1])
    on_dev = dist.get_backend() == 'nccl'                            # RCCL moves device tensors, gloo host tensors
    dev = rows.device if on_dev else torch.device('cpu')
    sizes = torch.zeros(w, dtype=torch.int64, device=dev)
    mine = torch.tensor([n], dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(sizes, mine)
    cap = int(sizes.max())
    ids = torch.full((cap,), -1, dtype=torch.int64, device=dev)
    ids[:n] = torch.as_tensor(np.asarray(owned, dtype=np.int64)).to(dev)
    body = torch.zeros((cap, 2 * width), dtype=torch.float32, device=dev)
    body[:n, :width] = rows.to(dev)
    body[:n, width:] = slots.to(dev)
    all_ids = torch.empty((w * cap,), dtype=torch.int64, device=dev)
    all_body = torch.empty((w * cap, 2 * width), dtype=torch.float32, device=dev)
    dist.all_gather_into_tensor(all_ids, ids)
    dist.all_gather_into_tensor(all_body, body)
    all_ids, all_body, sizes = all_ids.cpu().numpy(), all_body.cpu().numpy(), sizes.cpu().numpy()
    return [(all_ids[r * cap:r * cap + sizes[r]], all_body[r * cap:r * cap + sizes[r], :width], all_body[r * cap:r * cap + sizes[r], width:])
            for r in range(w)]
