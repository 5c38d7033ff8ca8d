"""Device-side state and step loop behind BPR.train (the work single/bpr.py:107-153 hands to a
TensorFlow session in the reference).  PyTorch-ROCm owns memory and streams; all numerics are
HIP kernels reached through the C ABI (tkr_hip).  Nothing here computes on the CPU.
"""
from __future__ import annotations

import numpy as np
import torch

import tkr_hip
from ._config import Tuning

RHO, EPS = 0.9, 1e-10          # tf.train.RMSPropOptimizer defaults (decay, epsilon)
MAX_PLAN_BATCHES = tkr_hip.PLAN_MAX_BATCHES   # batches planned per K1 call


def default_device():
    if not torch.cuda.is_available():
        raise tkr_hip.TkrError('no MI355X visible to PyTorch-ROCm: BPR/VBPR training runs only on the '
                               'HIP path (there is no CPU fallback)')
    return torch.device('cuda', torch.cuda.current_device())


class TrainingCSR:
    """tr_data / tr_users (single/bpr.py:63-65) as device CSR for K1."""

    def __init__(self, tr_data: dict, tr_users, n_users: int, device):
        deg = np.zeros(n_users, dtype=np.int64)
        for u, items in tr_data.items():
            deg[u] = len(items)
        row_ptr = np.zeros(n_users + 1, dtype=np.int64)
        np.cumsum(deg, out=row_ptr[1:])
        order = sorted(tr_data.keys())
        pos = np.fromiter((it for u in order for it in tr_data[u]), dtype=np.int32, count=int(row_ptr[-1])) \
            if len(order) else np.zeros(0, np.int32)
        self._finish(row_ptr, pos, np.asarray(list(tr_users), dtype=np.int32), device)

    @classmethod
    def from_arrays(cls, row_ptr, pos_cols, tr_users, device):
        self = cls.__new__(cls)
        self._finish(np.asarray(row_ptr, dtype=np.int64), np.asarray(pos_cols, dtype=np.int32),
                     np.asarray(tr_users, dtype=np.int32), device)
        return self

    @classmethod
    def shard(cls, row_ptr, pos_cols, owned, device):
        """CSR of the users `owned` (global indices) in the shard's own numbering: row q = user owned[q].  The engine of a
        rank allocates only these rows (multi-GPU: SURVEY.md §8e)."""
        row_ptr = np.asarray(row_ptr, dtype=np.int64)
        owned = np.asarray(owned, dtype=np.int64)
        deg = row_ptr[owned + 1] - row_ptr[owned]
        local_ptr = np.zeros(len(owned) + 1, dtype=np.int64)
        np.cumsum(deg, out=local_ptr[1:])
        take = np.repeat(row_ptr[owned] - local_ptr[:-1], deg) + np.arange(int(local_ptr[-1]), dtype=np.int64)
        self = cls.from_arrays(local_ptr, np.asarray(pos_cols)[take], np.flatnonzero(deg > 0).astype(np.int32), device)
        self.local = True
        return self

    def _finish(self, row_ptr, pos, tr_users, device):
        assert row_ptr[-1] < 2 ** 31
        owner = np.repeat(np.arange(len(row_ptr) - 1, dtype=np.int64), np.diff(row_ptr))
        big = int(pos.max()) + 1 if len(pos) else 1
        srt = (np.sort(owner * big + pos) % big).astype(np.int32)        # per-row ascending copy
        self.row_ptr = torch.from_numpy(row_ptr.astype(np.int32)).to(device)
        self.pos_cols = torch.from_numpy(pos).to(device)
        self.cols_sorted = torch.from_numpy(srt).to(device)
        self.tr_users = torch.from_numpy(tr_users).to(device)
        self.nnz = int(row_ptr[-1])


class PlanBuffers:
    """Output of K1 for up to ``cap`` batches of ``B`` triplets.  ``flow``: the dataflow form for the persistent step
    kernel K2f (128-byte task records + versioned occurrences) instead of per-launch wave records."""

    def __init__(self, cap: int, B: int, device, flow: bool = False, cols=None, owners: int = 0):
        """``cols`` = (d, row_cap): also room for the column plan of the VBPR step (tkr_vbpr_colplan); ``owners`` > 0 (with flow):
        the owner-ordered form for K2o (csrc/bpr_own.hip) -- item records of a batch in (row % owners, row) order + ``ohdr``"""
        self.cap, self.B, self.flow, self.cols, self.owners = cap, B, flow, cols, (owners if flow else 0)
        i32 = dict(dtype=torch.int32, device=device)
        if cols is not None:
            d, row_cap = cols
            self.colh = torch.empty(cap * d * 8, **i32)
            self.cent = torch.empty(cap * B * 2 * row_cap * 2, **i32)
            self.tcnt = torch.empty(cap * B, **i32)
            self.tent = torch.empty(cap * B * 2 * row_cap * 2, **i32)
        self.u = torch.empty(cap * B, **i32)
        self.i = torch.empty(cap * B, **i32)
        self.j = torch.empty(cap * B, **i32)
        self.task = torch.empty(cap * 3 * B * 4, **i32)
        self.occ = torch.empty(cap * 3 * B * 2, **i32)
        self.occt = torch.empty(cap * 3 * B, **i32)
        if flow:
            self.prec = torch.empty(cap * 3 * B * 32, **i32)
            self.pocc = torch.empty(cap * 3 * B * 4, **i32)
            if self.owners:
                self.ohdr = torch.zeros(self.owners * cap, **i32)
                self.xch = torch.zeros(cap * B * 2 * 2, **i32)       # K2o's scalar slots: a granule {value, epoch} per (triplet, role)
                self.epoch = 0                                       # launches on these slots so far
        else:
            self.rec_stride = tkr_hip.plan_max_blocks(B) * tkr_hip.plan_team(B) * 16
            self.rec = torch.empty(cap * self.rec_stride, **i32)
            self.hdr = torch.empty(cap * 4, **i32)
            self.tpar = torch.empty(cap * B, **i32)          # per-triplet row parities (K3 sparse view)
        self.loss = torch.zeros(cap, dtype=torch.float32, device=device)
        need = tkr_hip.plan_workspace_bytes(B, cap)                       # B > 8192: scratch of the grid-wide planner
        self.ws = torch.empty(need, dtype=torch.uint8, device=device) if need else None


class UpdateCounters:
    """Per-row update counts (parity = buffer holding the row) + K1's scratch bitmaps."""

    def __init__(self, n_users, n_items, device):
        i32 = dict(dtype=torch.int32, device=device)
        self.ucnt = torch.zeros(n_users, **i32)
        self.icnt = torch.zeros(n_items, **i32)
        self.touch_u = torch.zeros(n_users * 16, **i32)
        self.touch_i = torch.zeros(n_items * 16, **i32)


class PlanPipeline:
    """Two plan buffers and the stream discipline around them.  K1 for the chunk after the current one may run on
    a side stream while the main stream runs the step kernels of the current chunk (the planner only needs the
    sample stream position and the update counters, never the model tables).  A side-stream plan starts after
    everything the main stream holds at that moment (the steps that last read the buffer, an in-order K1 that owns
    the counters); the steps of a chunk wait for the event of its plan."""

    _side_streams = {}                   # one planner stream per device and process: creating a HIP stream costs milliseconds

    def __init__(self, device, private_side=False):
        self.device = device
        self._private = private_side     # a planner stream of its own (shards that run concurrently in one process: BPR.train(streams=S))
        self._side = None                # taken on first overlapped use: an idle second queue is not free
        self.bufs = [None, None]
        self.planned = [None, None]      # event: plan in bufs[i] complete (side stream)

    @property
    def side(self):
        if self._side is None and self._private:
            self._side = torch.cuda.Stream(device=self.device)
        if self._side is None:
            key = (self.device.type, self.device.index)
            if key not in PlanPipeline._side_streams:
                PlanPipeline._side_streams[key] = torch.cuda.Stream(device=self.device)
            self._side = PlanPipeline._side_streams[key]
        return self._side

    def ensure(self, cap, B, flow=False, cols=None, owners=0):
        for i in range(2):
            b = self.bufs[i]
            if b is None or b.B != B or b.cap < cap or b.flow != flow or b.cols != cols or b.owners != (owners if flow else 0):
                self.bufs[i] = b = None               # release before allocating the replacement
                self.bufs[i] = PlanBuffers(cap, B, self.device, flow, cols, owners)
                self.planned[i] = None

    def plan(self, i, fn, overlap=True):
        """run fn(plan_buffer) on the side stream (or simply in order on the current stream when overlap is off)"""
        main = torch.cuda.current_stream(self.device)
        if not overlap:
            self.drain()                              # an earlier side-stream K1 owns the counters until it is done
            fn(self.bufs[i])
            self.planned[i] = None
            return
        with torch.cuda.stream(self.side):
            self.side.wait_stream(main)
            fn(self.bufs[i])
            ev = torch.cuda.Event()
            ev.record(self.side)
            self.planned[i] = ev
        self._side_dirty = True                       # something runs on the side stream that the main stream has not waited for

    def acquire(self, i):
        if self.planned[i] is not None:
            torch.cuda.current_stream(self.device).wait_event(self.planned[i])
            self.planned[i] = None
        return self.bufs[i]

    def drain(self):
        # only when a side-stream plan was queued since the last drain: an event create + record + wait + destroy is ~15 us of host
        # time in front of the first launch of a short call (the driver's 20-step run)
        if self._side is not None and getattr(self, '_side_dirty', False):
            torch.cuda.current_stream(self.device).wait_stream(self._side)
            self._side_dirty = False


# Tuning switches (TKR_* variables): single/_config.py, parsed once per engine into `self.cfg`.  overlap_min_batch: below it the planner is
# < 5 % of the time and a second active queue slows the dependent launch cadence of the step kernels (measured: 5.8 -> 6.8 us at B=256)


def _chunk_cap(B):
    """batches planned per K1 call: <= 512 (bitmap words), <= 1M triplets of plan resident per buffer (2M above batch 8192,
    where a batch's wave records alone are 0.4 KB per triplet: two batches of 2^20)"""
    if B > 8192:
        return max(2, (1 << 21) // B)
    return min(MAX_PLAN_BATCHES, max(8, (1 << 20) // B))


class _Chunk:
    """a planned chunk: batches [used, nb) of plan buffer `idx` have not run yet; batch 0 starts at triplet `first`.
    ``shadow``: the chunk was planned AHEAD OF AN EXCHANGE against a zeroed copy of the item counters (this tensor); it only
    becomes runnable once the exchange has zeroed the real ones (PlanMixin.after_exchange)."""
    __slots__ = ('idx', 'nb', 'used', 'B', 'first', 'csr', 'shadow', 'epoch_ahead', 'pending', 'loss_zeroed', 'fused')

    def __init__(self, idx, nb, B, first, csr, shadow=None):
        self.idx, self.nb, self.used, self.B, self.first, self.csr = idx, nb, 0, B, first, csr
        self.shadow, self.epoch_ahead = shadow, shadow is not None
        self.loss_zeroed = False        # the planner zeroed this chunk's loss words beside K1 (not a launch in front of the step)
        self.pending = None             # a tkr_hip.PlanCall: K1 of this chunk has NOT been launched yet -- the step's call launches it (short calls)
        self.fused = False              # ... and that is how this chunk was planned (BprEngine._run_again)


class _ShadowCounters:
    """the engine's update counters with the item counts replaced (K1 of the epoch after an exchange: users go on, items restart at 0)"""

    def __init__(self, cnt, icnt):
        self.ucnt, self.icnt, self.touch_u, self.touch_i = cnt.ucnt, icnt, cnt.touch_u, cnt.touch_i


def _event_pair():
    return torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def _recorded_event_pair():
    """a pair that exists on the device (a torch event is created by its first record())"""
    pair = _event_pair()
    for e in pair:
        e.record()
    return pair


class PlanMixin:
    """Sample stream position + plans of an engine.

    K1 plans a chunk from the current stream position: a FULL chunk (512 batches at B = 256) while the call still has that
    many batches to run, otherwise exactly what the call has left -- every run_batches call samples and plans its own
    batches, like the loop of single/bpr.py:138-147 does (rounds 1-2 always planned a full chunk, so a short call consumed
    batches that an earlier call had sampled: bench.py's 20 timed steps contained no K1 -- VERDICT r2).  A chunk that a
    caller leaves unfinished all the same (an exception, plan_ahead's users) is continued by the next call, and whenever
    something needs the counters of the batches that really RAN (get / set of parameters, the per-epoch exchange, a
    different batch size or sample position) settle() takes the rest of the plan out again (tkr_plan_rollback): K1
    advances the update counters for every PLANNED batch.  The stream is counter-based, so re-planning from the same
    position reproduces the same triplets."""

    def _plan_flow(self):
        """does the step of this engine read the dataflow form of the plan?"""
        return False

    def _plan_owners(self, B):
        """owners of item rows when the step is K2o (the owner-ordered dataflow form), else 0"""
        return 0

    def prepare(self, B, layout=None):
        """engines with more than one table layout pick the one batch size B runs on"""

    def check(self):
        """raise if the device reported a failed step (engines with a persistent kernel); synchronises"""

    def check_async(self):
        """queue a copy of the device's status word; the NEXT check_async / check / settle raises if it was set (no host wait
        now: the per-epoch exchange of a sharded run calls this instead of check)"""

    def _cap(self, B):
        """batches planned per K1 call"""
        return _chunk_cap(B)

    def _plan_overlap(self, B):
        """plan the chunk after the running one on the side stream?"""
        if getattr(self, 'plan_in_order', False):      # one of several shards on one GPU: no planner stream (see BPR._train_streams)
            return False
        return B >= self.cfg.overlap_min_batch or self._plan_flow()

    def _plan_cols(self, B):
        """(d, row_cap) when the step of this engine wants the column plan of its batches beside K1's (VBPR), else None"""
        return None

    def _plan_extra(self, buf, nb, B):
        """more planner-side work for the nb batches K1 just wrote into buf (same stream, right behind K1)"""

    def _init_plans(self, tuning=None):
        self.cfg = tuning if tuning is not None else Tuning.from_env()
        self._cnt = UpdateCounters(self.n_users, self.n_items, self.device)
        self._drawn = 0                 # position in the counter-based sample stream = triplets that RAN
        self._cur = None                # chunk being consumed
        self._ahead = None              # chunk planned on the side stream while _cur runs (large batches only)
        self.plan = None                # the plan buffer of the last chunk that ran
        self.pipe = None
        self.step_events = None         # list of (start, end, n_launches) when a bench wants kernel time
        self._event_pool = []           # event pairs that already exist on the device (reserve_events)
        self._fused_calls = {}          # plan buffer index -> (the C struct of its fused K1 + step call, the csr it names)
        self._again_key = None          # what the last call was, when the same call again can skip the planning decisions (BprEngine)
        self._again_buf = 0             # ... and the plan buffer it used

    @property
    def cnt(self):
        self.settle()
        return self._cnt

    @property
    def triplets_drawn(self):
        return self._drawn

    @triplets_drawn.setter
    def triplets_drawn(self, value):
        self.settle()
        self._drawn = int(value)

    def reserve_events(self, n):
        """n event pairs for step_events, created NOW: a torch event is only created on the device by its first record(),
        tens of microseconds that would otherwise land between the launches of a short timed run"""
        while len(self._event_pool) < n:
            self._event_pool.append(_recorded_event_pair())

    def settle(self, check=True, keep_epoch_ahead=False):
        """drop what is planned but has not run; afterwards the counters describe the tables.  Everything that takes results
        out of an engine (get / set, the exchange, the counters) comes through here, so this is also where a failed
        persistent step is reported (only BPR._run_epoch looked at the status word).
        ``keep_epoch_ahead`` (the exchange itself): a chunk planned ahead of the exchange stays -- the item counters it was
        planned against are its own (shadow) or, once adopted, describe tables that are exactly as the plan assumes."""
        if check:
            self.check()
        if self._cur is None and self._ahead is None:
            return
        ah, cur = self._ahead, self._cur
        if keep_epoch_ahead and ah is not None and ah.epoch_ahead and ah.used == 0 and (cur is None or cur.used == cur.nb):
            return
        self.pipe.drain()
        for ch in (self._ahead, self._cur):          # newest first, like unwinding
            if ch is not None and ch.used < ch.nb and ch.pending is None:
                buf = self.pipe.bufs[ch.idx]
                cnt = self._cnt if ch.shadow is None else _ShadowCounters(self._cnt, ch.shadow)
                tkr_hip.plan_rollback(buf, ch.B, ch.used, ch.nb - ch.used, cnt)
        self._cur = self._ahead = None

    def after_exchange(self):
        """dist.ItemSync calls this right after it re-assigned the item tables (every row at version 0, item counters zero):
        what a chunk planned ahead of the exchange counted on its shadow becomes the real counters."""
        ah = self._ahead
        if ah is None or ah.shadow is None:
            return
        ev = self.pipe.planned[ah.idx]
        if ev is not None:
            torch.cuda.current_stream(self.device).wait_event(ev)
        self._cnt.icnt.copy_(ah.shadow)
        ah.shadow = None

    def _plan_chunk(self, idx, csr, B, first, overlap, want, shadow=None, fuse=False):
        nb = max(1, min(self._cap(B), want))
        cnt = self._cnt if shadow is None else _ShadowCounters(self._cnt, shadow)
        if fuse:            # K1 rides in the step's C call (tkr_bpr_own_plan_run): nothing is launched here
            self.pipe.drain()
            buf = self.pipe.bufs[idx]
            key = ('fused', id(csr), id(cnt.ucnt), id(cnt.icnt), self.seed)
            callers = buf.__dict__.setdefault('_callers', {})
            call = callers.get(key)
            if call is None:
                if len(callers) > 4:
                    callers.clear()
                call = callers[key] = tkr_hip.plan_call(csr, self.n_users, self.n_items, self.seed, B, cnt, buf)
            call.first_triplet, call.n_batches = first, nb
            self.pipe.planned[idx] = None
            ch = _Chunk(idx, nb, B, first, csr, shadow)
            ch.pending, ch.fused = call, shadow is None
            self._fused_calls[idx] = (call, csr, buf)
            return ch

        def fn(buf):
            if shadow is not None:
                shadow.zero_()
            key = (id(csr), id(cnt.ucnt), id(cnt.icnt), self.seed)          # the closure holds csr and cnt: their ids stay theirs
            callers = buf.__dict__.setdefault('_callers', {})
            call = callers.get(key)
            if call is None:
                if len(callers) > 4:
                    callers.clear()
                call = callers[key] = tkr_hip.plan_caller(csr, self.n_users, self.n_items, self.seed, B, cnt, buf)
            call(first, nb)
            self._plan_extra(buf, nb, B)
            buf.loss[:nb].zero_()                     # where K1 runs (the side stream when it is planned ahead): the step starts without a fill in front of it
        self.pipe.plan(idx, fn, overlap)
        ch = _Chunk(idx, nb, B, first, csr, shadow)
        ch.loss_zeroed = True
        return ch

    def _epoch_ahead_ok(self, B):
        """may the first chunk of the epoch AFTER an exchange be planned before it?  Only where the exchange leaves the tables in
        a state K1 can name in advance without touching them (the granule layout: every item at version 0, dist.ItemSync's
        fused unpack) and takes no snapshot at the next begin()."""
        return self._plan_flow() and self.cfg.epoch_ahead and not getattr(self, 'plan_in_order', False)

    def _next_chunk(self, csr, B, want, then_exchange=0):
        """the chunk that holds the batch at the current stream position; `want` = batches the running call still has to run;
        `then_exchange` = batches of the NEXT epoch when an exchange of the item tables follows this call (0: none)"""
        cur = self._cur
        if cur is not None and (cur.B != B or cur.csr is not csr):
            self.settle(check=False)
            cur = None
        if cur is not None and cur.used < cur.nb:
            return cur
        if self._ahead is not None and self._ahead.shadow is not None:      # planned ahead of an exchange that never came
            self.settle(check=False)
            cur = None
        if self.pipe is None:
            self.pipe = PlanPipeline(self.device, getattr(self, 'private_side_stream', False))
        self.pipe.ensure(self._cap(B), B, self._plan_flow(), self._plan_cols(B), self._plan_owners(B))
        # K1 of the chunk after this one on the side stream: behind the per-batch launches of a large batch, or behind the ONE
        # persistent launch of a dataflow chunk (100 us of planner per 512 batches that otherwise sit in front of every 1.4 ms launch)
        overlap = self._plan_overlap(B)
        if overlap:
            self.pipe.side                            # exists from the first call on (a warm-up then pays for its creation, not a later call)
        if cur is not None and self._ahead is not None:
            nxt, self._ahead = self._ahead, None
        else:
            # a short call (this chunk holds all that is left of it, nothing is planned beside it): K1 goes out in the step's call
            fuse = (self.cfg.fuse_short and then_exchange == 0 and want <= self._cap(B) and
                    getattr(getattr(self, '_step', None), 'plan_and_run', None) is not None and self._plan_owners(B) > 0)
            nxt = self._plan_chunk(0 if cur is None else cur.idx ^ 1, csr, B, self._drawn, False, want, fuse=fuse)
        if overlap and want > nxt.nb:                 # the chunk after it, behind the steps of this one
            self._ahead = self._plan_chunk(nxt.idx ^ 1, csr, B, nxt.first + nxt.nb * B, True, want - nxt.nb)
        elif overlap and then_exchange == 0 and not self._plan_flow() and self.cfg.call_ahead and want == nxt.nb and want > 8:
            # the LAST chunk of a call on the plain layout (batch > 512: a launch per batch, K1 6-10 us per batch beside 19 us of step):
            # callers come back for more of the same -- BPR.train runs an epoch per call, at batch 8192 an epoch of 122 batches is
            # less than one chunk and NOTHING was ever planned beside the steps -- so the next call's first chunk is planned now, of
            # this chunk's size, behind this chunk's steps.  A caller that does something else first finds it rolled back (settle()).
            self._ahead = self._plan_chunk(nxt.idx ^ 1, csr, B, nxt.first + nxt.nb * B, True, nxt.nb)
        elif then_exchange > 0 and want <= nxt.nb and self._epoch_ahead_ok(B):
            # the LAST chunk of an epoch: after the exchange every item row is at version 0 again and the users go on where this
            # chunk leaves them -- everything K1 needs to plan the first chunk of the next epoch, so it runs now, on the side stream
            # behind this chunk's steps and the exchange, instead of ~120 us in front of the next epoch's first step
            if getattr(self, '_shadow_icnt', None) is None:
                self._shadow_icnt = torch.zeros_like(self._cnt.icnt)
            self.pipe.side
            self._ahead = self._plan_chunk(nxt.idx ^ 1, csr, B, nxt.first + nxt.nb * B, True, then_exchange, shadow=self._shadow_icnt)
        self._cur = nxt
        return nxt

    def _run(self, csr, n_batches, B, want_loss, step_fn, then_exchange=0):
        """n_batches consecutive batches from the current stream position; step_fn(plan, first_batch, nb, loss)"""
        loss, left = None, n_batches
        while left > 0:
            ch = self._next_chunk(csr, B, left, then_exchange)
            plan = self.pipe.acquire(ch.idx)
            m = min(left, ch.nb - ch.used)
            lo = ch.used
            if want_loss and not ch.loss_zeroed and not getattr(step_fn, 'assigns_loss', False):
                plan.loss[lo:lo + m].zero_()
            if ch.pending is not None:                # K1 of this chunk and the step on its first m batches: one C call
                call, ch.pending = ch.pending, None
                ev = None
                if self.step_events is not None:
                    ev = self._event_pool.pop() if self._event_pool else _recorded_event_pair()
                    self.step_events.append((ev[0], ev[1], m))
                step_fn.plan_and_run(plan, call, lo, m, plan.loss if want_loss else None, ev)
            elif self.step_events is not None:        # bench: HIP events around the step launches
                e0, e1 = self._event_pool.pop() if self._event_pool else _recorded_event_pair()
                if getattr(step_fn, 'takes_events', False):         # recorded in C, right around the launch
                    step_fn(plan, lo, m, plan.loss if want_loss else None, (e0, e1))
                else:
                    e0.record()
                    step_fn(plan, lo, m, plan.loss if want_loss else None)
                    e1.record()
                self.step_events.append((e0, e1, m))
            else:
                step_fn(plan, lo, m, plan.loss if want_loss else None)
            ch.used += m
            left -= m
            self._drawn += m * B
            loss = plan.loss[lo:lo + m] if want_loss else None
            self.plan = plan
        return loss


def plan_ahead(eng, csr, n_batches, B):
    """K1 for n_batches batches into freshly held buffers (no stepping): -> [(PlanBuffers, nb), ...].
    Used by the multi-stream mode, where planner launches would disturb the other streams' step chains."""
    eng.settle()
    cap = eng._cap(B)                 # an engine may plan fewer batches per call than the bitmap allows (VBPR: its column plans)
    cols = eng._plan_cols(B)          # ... and want the column plan of its batches beside K1's (VBPR train(streams > 1) crashed without)
    planned, left = [], n_batches
    pool = getattr(eng, '_plan_pool', [])
    idx = 0
    while left:
        nb = min(cap, left)
        if idx >= len(pool) or pool[idx].B != B or pool[idx].cap < nb or pool[idx].cols != cols or pool[idx].flow:
            buf = PlanBuffers(cap, B, eng.device, cols=cols)
            if idx < len(pool):
                pool[idx] = buf
            else:
                pool.append(buf)
        buf = pool[idx]
        tkr_hip.sample_plan(csr, eng.n_users, eng.n_items, eng.seed, eng._drawn, nb, B, eng._cnt, buf)
        eng._plan_extra(buf, nb, B)
        eng._drawn += nb * B
        planned.append((buf, nb))
        left -= nb
        idx += 1
    eng._plan_pool = pool
    return planned


def run_planned(eng, planned, B, want_loss, step_fn):
    """step kernels of batches planned by plan_ahead, on the current stream"""
    loss = None
    for plan, nb in planned:
        if want_loss:
            plan.loss[:nb].zero_()
        step_fn(plan, 0, nb, plan.loss if want_loss else None)
        loss = plan.loss[:nb] if want_loss else None
        eng.plan = plan
    return loss


class DoubleTable:
    """A parameter table + its RMSProp slot, double-buffered for K2 ([2][n][k], see csrc/bpr_step.hip).
    Which buffer holds row r is the parity of the row's update counter (UpdateCounters), shared by the
    tables that are updated together (V and b; ire and irb)."""

    def __init__(self, n, k, device, init=None, gen=None):
        shape = (2, n, k) if k else (2, n)
        self.p = torch.zeros(shape, dtype=torch.float32, device=device)
        self.ms = torch.ones(shape, dtype=torch.float32, device=device)
        if init is not None:
            self.p[0].normal_(0.0, init, generator=gen)

    def current(self, cnt):
        sel = (cnt & 1).long()
        idx = torch.arange(self.p.shape[1], device=self.p.device)
        return self.p[sel, idx], self.ms[sel, idx]

    def assign(self, values, ms=None):
        """write values into buffer 0 (the caller zeroes the update counters so that buffer 0 is current)"""
        self.mutations = getattr(self, 'mutations', 0) + 1
        self.p[0].copy_(values)
        if ms is not None:
            self.ms[0].copy_(ms)


def _generators(device, seed, user_seed):
    """(generator of the replicated tables, generator of the user rows).  One process: the same object, users first, as
    ever.  A user shard: two generators, so that every rank draws the same item tables whatever its number of users."""
    gen = torch.Generator(device=device)
    if user_seed is None:
        gen.manual_seed(seed & 0x7FFFFFFFFFFFFFFF)
        return gen, gen
    gen.manual_seed((seed ^ 0x5BD1E9955BD1E995) & 0x7FFFFFFFFFFFFFFF)
    gen_u = torch.Generator(device=device)
    gen_u.manual_seed(int(user_seed) & 0x7FFFFFFFFFFFFFFF)
    return gen, gen_u


def _tags(t):
    """the version tags of a granule tensor [..., 2] (float32 storage; [..., 0] value bits, [..., 1] tag bits)"""
    return t.view(torch.int32)[..., 1]


class FlowTable:
    """A parameter table + its RMSProp slot in the layout of the dataflow step (csrc/bpr_flow.hip): [bufs][n][kp] granules
    {fp32 value, uint32 version tag}, kp = k rounded up to 128 (tkr_flow_row_granules); version v of a row lives in
    buffer v & (bufs - 1); bufs = 2, or 4 for the item tables (tkr_flow_state.item_bufs).  Same interface as DoubleTable."""

    def __init__(self, n, k, device, bufs=2):
        assert bufs in (2, 4)
        self.n, self.k, self.kp, self.bufs = n, k, tkr_hip.flow_row_granules(k), bufs
        self.p = torch.zeros((bufs, n, self.kp, 2), dtype=torch.float32, device=device)
        self.ms = torch.zeros((bufs, n, self.kp, 2), dtype=torch.float32, device=device)

    def current(self, cnt):
        sel = (cnt & (self.bufs - 1)).long()
        idx = torch.arange(self.n, device=self.p.device)
        return self.p[sel, idx, :self.k, 0], self.ms[sel, idx, :self.k, 0]

    def assign(self, values, ms):
        """version 0 of every row in buffer 0 (the caller zeroes the update counters); the other buffers hold no version"""
        self.mutations = getattr(self, 'mutations', 0) + 1
        for t, v, pad in ((self.p, values, 0.0), (self.ms, ms, 1.0)):
            t.zero_()
            t[0, :, :, 0] = pad
            t[0, :, :self.k, 0] = v
            _tags(t)[1:] = -1


class FlowTail:
    """[bufs][n][2 * bufs] granules per row: {item bias, its RMSProp slot, expect[0 .. bufs-1] (+ two of padding with four
    buffers)} + the bufs rd words of every row"""

    def __init__(self, n, device, bufs=2):
        assert bufs in (2, 4)
        self.n, self.bufs = n, bufs
        self.t = torch.zeros((bufs, n, 2 * bufs, 2), dtype=torch.float32, device=device)
        self.rd = torch.zeros(bufs * n, dtype=torch.int32, device=device)
        _tags(self.t)[1:] = -1

    def current(self, cnt):
        sel = (cnt & (self.bufs - 1)).long()
        idx = torch.arange(self.n, device=self.t.device)
        return self.t[sel, idx, 0, 0], self.t[sel, idx, 1, 0]

    def assign(self, b=None, msb=None):
        self.mutations = getattr(self, 'mutations', 0) + 1
        self.t.zero_()
        if b is not None:
            self.t[0, :, 0, 0] = b
            self.t[0, :, 1, 0] = msb
        _tags(self.t)[1:] = -1
        self.rd.zero_()


class BprEngine(PlanMixin):
    """Tables + sampler + step loop of one BPR model on one GPU.

    Two layouts of the same model.  ``bulk``: plain double-buffered tables, one launch of K2 per batch (large batches:
    bandwidth-bound).  ``flow``: granule tables with in-band versions, ONE persistent launch of K2f per chunk (batch sizes
    up to cfg.flow_max_batch, where a launch per batch is latency-bound).  run_batches picks by batch size and converts the
    tables when it changes; get / set work on either."""

    def __init__(self, n_users, n_items, k, hp, device=None, seed=None, user_seed=None, tuning=None):
        """``user_seed``: this engine holds ONE SHARD of the users (multi-GPU: n_users = rows owned by the rank): the user
        rows are drawn from their own generator (a different one per rank), the replicated item tables from ``seed`` alone
        (identical on every rank)."""
        self.device = device or default_device()
        self.n_users, self.n_items, self.k = n_users, n_items, k
        self.hp = hp
        self.seed = int(seed if seed is not None else np.random.SeedSequence().entropy % (2 ** 63))
        gen, gen_u = _generators(self.device, self.seed, user_seed)
        # single/bpr.py:77-79: U, V ~ N(0, 0.01); b = 0.  RMSProp `rms` slots start at one.
        self.layout = 'bulk'
        self.layout_epoch = 0           # bumped whenever the tables are re-allocated (dist.ItemSync re-binds)
        self.U = DoubleTable(n_users, k, self.device, 0.01, gen_u)
        self.V = DoubleTable(n_items, k, self.device, 0.01, gen)
        self.b = DoubleTable(n_items, 0, self.device)
        self.tailU = self.tailV = self.ctl = None
        self._item_mutations = 0        # writes to the item tables from outside the step kernels (dist.ItemSync: is its snapshot still the truth?)
        self._flow_ran = False          # a persistent launch ran since the status word was last looked at
        self._status_host = self._status_event = None
        self._init_plans(tuning)

    @property
    def item_mutations(self):
        """every write to the item tables from outside the step kernels: the engine's own (set_items, prepare) AND direct
        assign() calls on the table objects (those left dist.ItemSync's cached epoch start stale, silently)"""
        return self._item_mutations + sum(getattr(t, 'mutations', 0) for t in (self.V, self.b, self.tailV) if t is not None)

    @item_mutations.setter
    def item_mutations(self, value):
        self._item_mutations = value - sum(getattr(t, 'mutations', 0) for t in (self.V, self.b, self.tailV) if t is not None)

    # ---- layout ----------------------------------------------------------------------------------
    def _plan_flow(self):
        return self.layout == 'flow'

    def _plan_owners(self, B):
        """workgroups that own item rows when batch size B steps with K2o, else 0.  K2o wants its workgroups -- 12 waves each, one
        per CU -- all resident at once; ranks that share a GPU (the multi-rank tests of a one-GPU box, a launcher that packs ranks)
        split the CUs: `ranks_on_device` (dist.ranks_sharing_device, set by whoever initialises the process group) ranks run
        CUs // ranks owners each, as long as the rows of an owner still fit its LDS."""
        if self.layout != 'flow' or B > min(self.cfg.own_max_batch, 1024) or self.cfg.own == '0' or getattr(self, '_own_failed', False):
            return 0
        share = max(1, int(getattr(self, 'ranks_on_device', 1)))
        cache = self.__dict__.setdefault('_owners_by_share', {})
        if share not in cache:                         # a property of the device, the table shape and the ranks beside us
            cache[share] = tkr_hip.bpr_own_owners(self.n_items, self.k, self.device, share)
            if cache[share] == 0 and share > 1:
                import warnings
                cus = torch.cuda.get_device_properties(self.device).multi_processor_count if self.device.type == 'cuda' else 0
                warnings.warn('K2o is off: %d ranks share this GPU and %d item rows of width %d do not fit %d owners\' LDS; the persistent '
                              'step without owned rows (K2f) runs instead' % (share, self.n_items, self.k, cus // share))
        if cache[share] < getattr(self, 'own_min_owners', 0):          # shards of one process (BPR._train_streams): K2f below that many owners
            return 0
        return cache[share]

    def wants_flow(self, B):
        """the granule layout + persistent step for this batch size?  Not when the granule tables would not fit: a granule row
        is k rounded up to 128 elements of 8 bytes, in two buffers, parameter + slot = 4 KB per row WHATEVER k is (k = 16: 16x
        the plain layout), and prepare() holds a copy of the old tables while it builds the new ones."""
        if B > self.cfg.flow_max_batch or not self.cfg.flow or self.k > tkr_hip.FLOW_MAX_K:
            return False
        if getattr(self, '_flow_disabled', False):     # a bounded spin of the persistent kernel ran out in this process: K2 from here on
            return False
        if self.layout == 'flow':
            return True
        # decided ONCE per engine (asked on every call, the answer -- and with it the summation order of the step --
        # followed whatever memory happened to be free, could differ between ranks and flip in mid-training)
        fits = getattr(self, '_flow_fits', None)
        if fits is None:
            rows = self.n_users + self.n_items
            need = (rows + (self.cfg.flow_item_bufs // 2 - 1) * self.n_items) * (tkr_hip.flow_row_granules(self.k) * 32 + 96 + 8 * self.k)
            free = torch.cuda.mem_get_info(self.device)[0] if self.device.type == 'cuda' else need
            fits = self._flow_fits = need <= 0.7 * free
        return fits

    def prepare(self, B, layout=None):
        """put the tables into the layout that batch size B runs on (or the one named)"""
        layout = layout or ('flow' if self.wants_flow(B) else 'bulk')
        if layout == self.layout:
            return
        self.settle()
        (pu, mu), (pv, mv), (pb, mb) = (tuple(t.clone() for t in self.get(n)) for n in ('U', 'V', 'b'))
        self.U = self.V = self.b = self.tailU = self.tailV = None
        if layout == 'flow':
            self.U, self.V = FlowTable(self.n_users, self.k, self.device), FlowTable(self.n_items, self.k, self.device, self.cfg.flow_item_bufs)
            self.tailU, self.tailV = FlowTail(self.n_users, self.device), FlowTail(self.n_items, self.device, self.cfg.flow_item_bufs)
            if self.ctl is None:
                self.ctl = torch.zeros(tkr_hip.flow_ctl_words(), dtype=torch.int32, device=self.device)
        else:
            self.U, self.V = DoubleTable(self.n_users, self.k, self.device), DoubleTable(self.n_items, self.k, self.device)
            self.b = DoubleTable(self.n_items, 0, self.device)
        self.layout = layout
        self.layout_epoch += 1
        self.item_mutations += 1
        self.U.assign(pu, mu)
        self.V.assign(pv, mv)
        if layout == 'flow':
            self.tailU.assign()
            self.tailV.assign(pb, mb)
        else:
            self.b.assign(pb, mb)
        self._cnt.ucnt.zero_()
        self._cnt.icnt.zero_()

    def _failed(self, code):
        post = self.ctl[tkr_hip.FLOW_CTL_DEBUG:tkr_hip.FLOW_CTL_DEBUG + 16].cpu().tolist()      # what the first wave that gave up waited for
        self.ctl.zero_()
        self._flow_ran = False
        # The persistent kernel needs one resident wave per ticket queue; where that cannot be had (a GPU shared with a kernel that
        # never ends) every later launch would time out the same way.  The tables of THIS run are lost -- a launch is not
        # transactional -- but the engine stays usable: from here on it steps with K2 (one launch per batch, no co-residency
        # assumption), e.g. after the caller re-imports a checkpoint.
        # K2o asks for more than K2f -- a 12-wave workgroup resident on EVERY CU: where THAT is what failed (the GPU is shared), the
        # next thing to try is K2f (4 waves per CU); a K2f launch that gives up goes straight to the per-batch step (the
        # step-down used to follow whether K2o was configured, not which kernel had run: a K2f failure at batch 512 of an engine
        # that had run K2o at batch 256 "stepped down" to K2f again).
        if getattr(self, '_last_step_kind', None) == 'own':
            self._own_failed = True
            nxt = 'the persistent step without owned rows (K2f)'
        else:
            self._flow_disabled = True
            nxt = 'the per-batch step (K2)'
        self.tables_invalid = True       # until somebody puts tables back (set_users + set_items: BPR.train's retry does)
        self._step_key = None
        raise tkr_hip.StepGaveUp('persistent BPR step gave up waiting for a row version (status %d): tables are invalid; this engine uses %s '
                               'from now on; post-mortem words (csrc/bpr_flow.hip kCtlDebug) %r' % (code, nxt, post))

    def step_down(self):
        """what _failed() does to the choice of kernel, without a failure of this engine's own: a sharded run restarts every rank
        together when ANY rank's persistent step gave up, and the ranks that did not fail themselves step down with it -- otherwise
        they retry the same kernel while the failed rank is one form further, every rank burns an attempt per try, and ranks on
        different layouts would meet in one all-reduce"""
        if getattr(self, '_last_step_kind', None) == 'own':
            self._own_failed = True
        else:
            self._flow_disabled = True
        self._step_key = None
        self._again_key = None

    def _raise_pending(self):
        ev, self._status_event = self._status_event, None
        if ev is not None:
            ev.synchronize()
            if int(self._status_host[0]) != 0:
                self._failed(int(self._status_host[0]))

    def check(self):
        """raise if a bounded spin of the persistent kernel ran out since the last check (synchronises; free when no
        persistent launch happened in between).  The status word is sticky on the device until it is zeroed here."""
        self._raise_pending()
        if self.ctl is None or not self._flow_ran:
            return
        self._flow_ran = False
        code = int(self.ctl[tkr_hip.FLOW_CTL_STATUS])
        if code != 0:
            self._failed(code)

    def check_async(self):
        """the same without a host wait: the status word is copied to pinned host memory behind the launches queued so far
        and looked at by the NEXT check_async / check (dist.ItemSync: once per epoch, a host synchronisation per exchange
        would drain the launch queue of every rank)"""
        self._raise_pending()
        if self.ctl is None or not self._flow_ran:
            return
        self._flow_ran = False
        if self._status_host is None:
            self._status_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self._status_host.copy_(self.ctl[tkr_hip.FLOW_CTL_STATUS:tkr_hip.FLOW_CTL_STATUS + 1], non_blocking=True)
        self._status_event = torch.cuda.Event()
        self._status_event.record()

    # ---- C-ABI state structs ------------------------------------------------------------
    def _hyper_into(self, st):
        hp = self.hp
        st.n_users, st.n_items, st.k = self.n_users, self.n_items, self.k
        st.mode = 0 if hp['mode'] == 'l2' else 1
        st.lu, st.li, st.lj, st.lb, st.lr = hp['lu'], hp['li'], hp['lj'], hp['lb'], hp['lr']
        st.rho, st.eps = RHO, EPS
        st.opt = 1 if hp.get('opt', 'rmsprop') == 'sgd' else 0       # 'sgd': old/methods/bpr.py:57-61
        return st

    def state(self):
        if self.layout == 'flow':
            st = tkr_hip.FlowState()
            st.U, st.msU, st.tailU, st.rdU = self.U.p.data_ptr(), self.U.ms.data_ptr(), self.tailU.t.data_ptr(), self.tailU.rd.data_ptr()
            st.V, st.msV, st.tailV, st.rdV = self.V.p.data_ptr(), self.V.ms.data_ptr(), self.tailV.t.data_ptr(), self.tailV.rd.data_ptr()
            st.item_bufs = self.V.bufs
            return self._hyper_into(st)
        st = tkr_hip.BprState()
        st.U, st.msU = self.U.p.data_ptr(), self.U.ms.data_ptr()
        st.V, st.msV = self.V.p.data_ptr(), self.V.ms.data_ptr()
        st.b, st.msb = self.b.p.data_ptr(), self.b.ms.data_ptr()
        return self._hyper_into(st)

    # ---- parameter access (host <-> current buffers) -----------------------------------------
    def get(self, name):
        if name == 'U':
            return self.U.current(self.cnt.ucnt)
        if name == 'V':
            return self.V.current(self.cnt.icnt)
        return (self.tailV if self.layout == 'flow' else self.b).current(self.cnt.icnt)

    def set_users(self, U=None, msU=None):
        cur, ms = self.U.current(self.cnt.ucnt)
        self.U.assign(cur.clone() if U is None else self._dev(U), ms.clone() if msU is None else self._dev(msU))
        if self.layout == 'flow':
            self.tailU.assign()
        self._cnt.ucnt.zero_()

    def set_items(self, V=None, b=None, msV=None, msb=None):
        self.item_mutations += 1
        cv, mv = self.V.current(self.cnt.icnt)
        cb, mb = self.get('b')
        nb = cb.clone() if b is None else self._dev(b).reshape(-1)
        nmb = mb.clone() if msb is None else self._dev(msb).reshape(-1)
        self.V.assign(cv.clone() if V is None else self._dev(V), mv.clone() if msV is None else self._dev(msV))
        if self.layout == 'flow':
            self.tailV.assign(nb, nmb)
        else:
            self.b.assign(nb, nmb)
        self._cnt.icnt.zero_()

    replicated_names = ('V', 'b')

    def replicated_tables(self):
        """(name, P, ms, update counter or None) of every table all ranks update: dist.ItemSync packs them with
        csrc/sync.hip.  The granule layout has its own fused kernels (flow_sync_tables): empty list here."""
        if self.layout == 'flow':
            return []
        return [('V', self.V.p, self.V.ms, self.cnt.icnt), ('b', self.b.p, self.b.ms, self.cnt.icnt)]

    def flow_sync_tables(self):
        """the item-side granule tables for dist.ItemSync's fused exchange (csrc/sync.hip tkr_sync_flow_*), or None in the plain layout"""
        if self.layout != 'flow':
            return None
        return self.V.p, self.V.ms, self.tailV.t, self.tailV.rd, self._cnt.icnt, self.n_items, self.k, self.V.bufs

    def snapshot(self):
        """parameters, slots and the sample stream position as they are now (device copies): what restore() puts back.  A launch of a
        persistent step is not transactional -- a bounded spin that runs out leaves the tables half-updated -- so BPR.train keeps the
        state it started from and re-runs from there on the next kernel down."""
        snap = {n: tuple(t.clone() for t in self.get(n)) for n in ('U', 'V', 'b')}
        snap['drawn'] = self._drawn
        return snap

    def restore(self, snap):
        self.settle(check=False)
        # K1's per-row touch bits of a call are cleared by ITS commit (plan_commit_first_touch / commit_kernel); a launch whose planner
        # prologue gave up never got there, and stale bits would give the next plan wrong versions and buffer parities
        self._cnt.touch_u.zero_()
        self._cnt.touch_i.zero_()
        self.set_users(U=snap['U'][0], msU=snap['U'][1])
        self.set_items(V=snap['V'][0], b=snap['b'][0], msV=snap['V'][1], msb=snap['b'][1])
        self._drawn = int(snap['drawn'])
        self.tables_invalid = False

    def copy_model_from(self, other):
        """start from another engine's current parameters and slots (shards of one model)"""
        p, ms = other.get('U')
        self.set_users(U=p, msU=ms)
        self.set_replicated({n: other.get(n) for n in self.replicated_names})

    def set_replicated(self, new):
        """write back all-reduced item-side tables: {'V': (p, ms), 'b': (p, ms)} (dist.ItemSync)"""
        self.set_items(V=new['V'][0], b=new['b'][0], msV=new['V'][1], msb=new['b'][1])

    def _dev(self, a):
        if isinstance(a, torch.Tensor):
            return a.to(self.device, torch.float32)
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(self.device)

    # ---- the loop ------------------------------------------------------------------------------
    def run_batches(self, csr: TrainingCSR, n_batches: int, B: int, want_loss=True, then_exchange=0):
        """sample + plan + step for n_batches consecutive batches; returns the per-batch
        losses of the LAST chunk as a device tensor (or None).  ``then_exchange`` = m > 0: the caller runs dist.ItemSync.end()
        right after this call and then goes on with an epoch of m batches -- its first chunk is planned ahead (PlanMixin)."""
        if (self.k > 512 or (self.k > 256 and B > 1024)) and not getattr(self, '_warned_wide', False):
            self._warned_wide = True
            import warnings
            warnings.warn('BPR: k = %d at batch_size = %d is wider than a wave can hold a row in registers beside its partners (k <= 512, '
                          '<= 256 above batch 1024): the generic row form steps (csrc/bpr_step.hip bpr_wide_kernel: every occurrence in two '
                          'passes over k, the gradient sum through memory)' % (self.k, B))
        cfg = self.cfg
        again = (id(csr), 0, B, self.layout_epoch, cfg.own, cfg.own_waves, cfg.fuse_short, cfg.fuse_plan, cfg.own_max_batch, cfg.flow, cfg.flow_max_batch, getattr(self, '_flow_disabled', False),
                 getattr(self, '_own_failed', False), getattr(self, 'ranks_on_device', 1), self.seed)
        if again == getattr(self, '_again_key', None) and then_exchange == 0 and self.step_events is None:
            loss = self._run_again(csr, n_batches, B, want_loss)
            if loss is not False:
                return loss
        self.prepare(B)
        again = again[:3] + (self.layout_epoch,) + again[4:]
        key = (self.layout_epoch, B, self.cfg.flow_waves_per_cu, self._plan_owners(B), self.cfg.own_waves, self.cfg.fuse_plan)
        if getattr(self, '_step_key', None) != key:       # the C struct and the closure are built once per layout, not per call
            self._step_key, self._step = key, self.step_fn(B)
        self._flow_ran = self._flow_ran or self.layout == 'flow'
        if self.layout == 'flow':
            self._last_step_kind = 'own' if self._plan_owners(B) else 'flow'
        loss = self._run(csr, n_batches, B, want_loss, self._step, then_exchange)
        # a short call that left in one C call (K1 in the step's launch): the same call again needs none of the decisions above
        cur = self._cur
        fused = cur is not None and cur.fused and cur.used == cur.nb and self._ahead is None
        self._again_key, self._again_buf = (again, cur.idx) if fused else (None, 0)
        return loss

    def _run_again(self, csr, n_batches, B, want_loss):
        """another short call like the last one (same training data, batch size, layout and switches; a driver's timed 20-batch call
        after its warm-up, an epoch of a small data set): the plan buffers and their C structs exist, so this is the stream position,
        one C call and the bookkeeping -- 12 us less Python in front of the one launch (scripts/probe_short_host.py: 124 -> 111 us
        per 20-batch call).  False: take the long way."""
        cur, pipe = self._cur, self.pipe
        if ((cur is not None and cur.used != cur.nb) or self._ahead is not None or getattr(pipe, '_side_dirty', False)       # (a settle() in between leaves no chunk)
                or getattr(self, 'tables_invalid', False)):
            return False
        if n_batches > self._cap(B):                  # more than one chunk: K1 of the second belongs on the side stream
            return False
        if n_batches <= 0 or (want_loss and not getattr(self._step, 'assigns_loss', False)):
            return False                              # nothing to run / a step form that ADDS to the loss words: _run zeroes them
        idx = self._again_buf ^ 1                     # the other plan buffer, as _next_chunk alternates them
        call = self._fused_calls.get(idx)
        if call is None or call[1] is not csr or call[2] is not pipe.bufs[idx]:       # planned for other data / a replaced buffer
            return False
        call, buf = call[0], call[2]
        call.first_triplet, call.n_batches = self._drawn, n_batches
        pipe.planned[idx] = None
        ch = _Chunk(idx, n_batches, B, self._drawn, csr)
        ch.fused = True
        self._step.plan_and_run(buf, call, 0, n_batches, buf.loss if want_loss else None, None)
        ch.used = n_batches
        self._cur, self.plan, self._again_buf = ch, buf, idx
        self._drawn += n_batches * B
        self._flow_ran, self._last_step_kind = True, 'own'
        return buf.loss[:n_batches] if want_loss else None

    def step_fn(self, B):
        state = self.state()
        if self.layout == 'flow':
            if self._plan_owners(B):
                return tkr_hip.own_stepper(state, B, self.ctl, self.cfg.own_waves | (0 if self.cfg.fuse_plan else 0x1000))
            return tkr_hip.flow_stepper(state, B, self.ctl, self.cfg.flow_waves_per_cu)
        return lambda plan, lo, nb, loss: tkr_hip.bpr_run(state, plan, B, nb, loss, first=lo)


class VbprEngine(PlanMixin):
    """Tables + sampler + step loop of one VBPR model on one GPU (single/vbpr.py:29-74).

    User rows hold [ure | uce] (width 2*kh, the layout of the exported ``fue``); item rows hold ire;
    cem / icb are dense and single-buffered (their update is its own launch, after every read)."""

    SPARSE_DENSITY = 0.25      # below this fraction of nonzeros the step uses the CSR/CSC view of feat (csrc/vbpr_step.hip S1/S3)
    NARROW_D = 512             # ... and below this width whatever the density
    COLS_MAX_BATCH = 1024      # the column-plan step (three launches per batch): pair sums are O(B^2) work of B waves
    PLAN_BYTES = 768 << 20     # column plans held per plan buffer (two buffers): bounds the batches planned per K1 call

    def __init__(self, n_users, n_items, k, d, feat, hp, device=None, seed=None, sparse=None, user_seed=None, tuning=None):
        self.device = device or default_device()
        self.n_users, self.n_items, self.k, self.kh, self.d = n_users, n_items, k, k // 2, d
        self.hp = hp
        self.seed = int(seed if seed is not None else np.random.SeedSequence().entropy % (2 ** 63))
        gen, gen_u = _generators(self.device, self.seed, user_seed)
        kh = self.kh
        self.U = DoubleTable(n_users, 2 * kh, self.device, 0.01, gen_u)        # vbpr.py:37-40
        self.I = DoubleTable(n_items, kh, self.device, 0.01, gen)              # vbpr.py:41
        self.irb = DoubleTable(n_items, 0, self.device)                        # vbpr.py:43
        f32 = dict(dtype=torch.float32, device=self.device)
        self.cem = torch.full((d, kh), 2.0 / (d * k), **f32)                   # vbpr.py:45-46
        self.mscem = torch.ones((d, kh), **f32)
        self.icb = torch.zeros(d, **f32)                                       # vbpr.py:47-48
        self.msicb = torch.ones(d, **f32)
        self.feat = feat if isinstance(feat, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(feat, dtype=np.float32))
        self.feat = self.feat.to(self.device).contiguous()
        assert self.feat.shape == (n_items, d)
        self._init_plans(tuning)
        self.ws = None
        self.sparse = None
        nnz = int(torch.count_nonzero(self.feat))
        # the gather view: sparse features (tf-idf-like, ~0.5 % dense at d = 20,000), and narrow dense ones (BASELINE.json's literal
        # "d = 128"): the fp32-MFMA kernels tile d by 128 / 64 columns and run on 1-2 workgroups there
        # (k // 2 > 128: only the column-plan step has a generic form, and it gathers: the CSR view whatever the density)
        if sparse or (sparse is None and (nnz <= self.SPARSE_DENSITY * n_items * d or d <= self.NARROW_D or self.kh > 128)):
            self.sparse = self._sparse_view(self.feat)
            per_row = (self.sparse['f_ptr'][1:] - self.sparse['f_ptr'][:-1])
            self.max_row_nnz, self.avg_row_nnz = int(per_row.max()), float(per_row.float().mean())

    @staticmethod
    def _sparse_view(feat):
        """CSR over items (ascending columns) and CSC over feature columns (ascending items) of the nonzeros of feat,
        plus the zeroed per-item tag scratch the kernels own (include/tkr.h tkr_vbpr_state)"""
        n_items, d = feat.shape
        i32 = torch.int32
        rc = (feat != 0).nonzero()                                   # row-major: rows ascending, columns ascending inside
        f_ptr = torch.zeros(n_items + 1, dtype=torch.int64, device=feat.device)
        f_ptr[1:] = torch.cumsum(torch.bincount(rc[:, 0], minlength=n_items), 0)
        cr = (feat.t() != 0).nonzero()                               # column-major
        c_ptr = torch.zeros(d + 1, dtype=torch.int64, device=feat.device)
        c_ptr[1:] = torch.cumsum(torch.bincount(cr[:, 0], minlength=d), 0)
        assert int(f_ptr[-1]) < 2 ** 31
        return dict(f_ptr=f_ptr.to(i32), f_col=rc[:, 1].to(i32).contiguous(), f_val=feat[rc[:, 0], rc[:, 1]].contiguous(),
                    c_ptr=c_ptr.to(i32), c_item=cr[:, 1].to(i32).contiguous(), c_val=feat[cr[:, 1], cr[:, 0]].contiguous(),
                    item_tag=torch.zeros(n_items + 8, dtype=torch.int64, device=feat.device))     # slots [n] int32 | two byte maps [ceil(n/4)*4] | counter: 8n + 16 bytes suffice, also for n <= 2

    # ---- column-plan step (csrc/vbpr_step.hip, "column-plan path") --------------------------------
    def wants_cols(self, B):
        if self.sparse is None or not self.cfg.vbpr_cols:
            return False
        # (kh % 4 == 0 and kh <= 128: the register form; kh > 128: the generic form of csrc/vbpr_wide.hip on the same plan)
        return (((self.kh % 4 == 0 and self.kh <= 128) or self.kh > 128) and B <= self.COLS_MAX_BATCH and 0 < self.max_row_nnz <= 1024 and
                tkr_hip.vbpr_colplan_lds_bytes(B, self.d) <= 160 * 1024)

    def _plan_cols(self, B):
        return (self.d, self.max_row_nnz) if self.wants_cols(B) else None

    def _cap(self, B):
        cap = _chunk_cap(B)
        if self.wants_cols(B):
            per_batch = 32 * self.d + 4 * B + 2 * 16 * B * self.max_row_nnz        # colh + tcnt + (cent, tent)
            cap = max(4, min(cap, self.PLAN_BYTES // per_batch))
        return cap

    def _plan_overlap(self, B):
        # the column plan costs ~1.5 us per batch: behind the steps of the chunk before, not in front of its own
        return B >= self.cfg.overlap_min_batch or (self.wants_cols(B) and self.cfg.vbpr_overlap)

    def _plan_extra(self, buf, nb, B):
        if buf.cols is not None:
            tkr_hip.vbpr_colplan(self.sparse, self.d, buf, B, nb, buf.cols[1])

    def cols_per_block(self, B):
        """feature columns per workgroup of the dense update: all the groups of a block own one while runs are short (sparse
        features: 2B * nnz_row / d entries per column), fewer when every column meets most triplets (narrow dense feat)"""
        lpc = 4 if self.kh <= 16 else 8 if self.kh <= 32 else 16 if self.kh <= 64 else 32
        groups = 256 // lpc
        run = 2.0 * B * self.avg_row_nnz / self.d
        return int(max(1, min(groups, groups * 8 // max(int(run), 1))))

    def state(self):
        hp = self.hp
        st = tkr_hip.VbprState()
        st.U, st.msU, st.I, st.msI = self.U.p.data_ptr(), self.U.ms.data_ptr(), self.I.p.data_ptr(), self.I.ms.data_ptr()
        st.irb, st.msirb = self.irb.p.data_ptr(), self.irb.ms.data_ptr()
        st.cem, st.mscem, st.icb, st.msicb = self.cem.data_ptr(), self.mscem.data_ptr(), self.icb.data_ptr(), self.msicb.data_ptr()
        st.feat = self.feat.data_ptr()
        st.n_users, st.n_items, st.kh, st.d = self.n_users, self.n_items, self.kh, self.d
        st.mode = 0 if hp['mode'] == 'l2' else 1
        st.lu, st.li, st.lj, st.lb, st.le, st.lr = hp['lu'], hp['li'], hp['lj'], hp['lb'], hp['le'], hp['lr']
        st.rho, st.eps = RHO, EPS
        if self.sparse is not None:
            for name, t in self.sparse.items():
                setattr(st, name, t.data_ptr())
        return st

    def get(self, name):
        if name == 'U':
            return self.U.current(self.cnt.ucnt)
        if name == 'I':
            return self.I.current(self.cnt.icnt)
        if name == 'irb':
            return self.irb.current(self.cnt.icnt)
        return {'cem': (self.cem, self.mscem), 'icb': (self.icb, self.msicb)}[name]

    _dev = BprEngine._dev

    def set_users(self, U=None, msU=None):
        cur, ms = self.U.current(self.cnt.ucnt)
        self.U.assign(cur if U is None else self._dev(U), ms if msU is None else self._dev(msU))
        self.cnt.ucnt.zero_()

    def set_items(self, I=None, irb=None, msI=None, msirb=None):
        ci, mi = self.I.current(self.cnt.icnt)
        cb, mb = self.irb.current(self.cnt.icnt)
        self.I.assign(ci if I is None else self._dev(I), mi if msI is None else self._dev(msI))
        self.irb.assign(cb if irb is None else self._dev(irb).reshape(-1), mb if msirb is None else self._dev(msirb).reshape(-1))
        self.cnt.icnt.zero_()

    def set_dense(self, cem=None, icb=None, mscem=None, msicb=None):
        for dst, src in ((self.cem, cem), (self.icb, icb), (self.mscem, mscem), (self.msicb, msicb)):
            if src is not None:
                dst.copy_(self._dev(src).reshape(dst.shape))

    def set_replicated(self, new):
        self.set_items(I=new['I'][0], irb=new['irb'][0], msI=new['I'][1], msirb=new['irb'][1])
        self.set_dense(cem=new['cem'][0], icb=new['icb'][0], mscem=new['cem'][1], msicb=new['icb'][1])

    replicated_names = ('I', 'irb', 'cem', 'icb')

    def replicated_tables(self):
        return [('I', self.I.p, self.I.ms, self.cnt.icnt), ('irb', self.irb.p, self.irb.ms, self.cnt.icnt),
                ('cem', self.cem, self.mscem, None), ('icb', self.icb, self.msicb, None)]
    copy_model_from = BprEngine.copy_model_from

    def run_batches(self, csr: TrainingCSR, n_batches: int, B: int, want_loss=True, then_exchange=0):
        if B > 65536 or (self.kh > 128 and not self.wants_cols(B)):
            raise ValueError('VBPR on the HIP path: batch_size <= 65536, and k // 2 > 128 only through the column-plan step (batch_size <= %d, '
                             'feature rows of at most 1024 nonzeros; got k = %d, batch_size = %d)' % (self.COLS_MAX_BATCH, self.k, B))
        if self.kh > 128 and not getattr(self, '_warned_wide', False):
            self._warned_wide = True
            import warnings
            warnings.warn('VBPR: k // 2 = %d is wider than the step\'s kernels hold a row in registers (128): the generic form steps '
                          '(csrc/vbpr_wide.hip: every kernel walks the factors in strides of its threads)' % self.kh)
        if getattr(self, '_step_key', None) != B:           # the C struct and the closure are built once per batch size, not per call
            self._step_key, self._step = B, self.step_fn(B)
        return self._run(csr, n_batches, B, want_loss, self._step)

    def step_fn(self, B):
        need = tkr_hip.vbpr_workspace_floats(B, self.kh, self.d)
        if self.ws is None or self.ws.numel() < need:
            self.ws = torch.empty(need, dtype=torch.float32, device=self.device)
        state = self.state()
        if self.wants_cols(B):
            d, row_cap, cpb = self.d, self.max_row_nnz, self.cols_per_block(B)
            return lambda plan, lo, nb, loss: tkr_hip.vbpr_run_cols(state, plan, B, nb, self.ws, d, row_cap, cpb, loss, first=lo)
        return lambda plan, lo, nb, loss: tkr_hip.vbpr_run(state, plan, B, nb, self.ws, loss, first=lo)
