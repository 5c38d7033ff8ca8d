"""Multi-GPU layout of the BPR/VBPR path: one process per GPU, users sharded, item-side state
replicated and reconciled by ONE all-reduce per epoch (RCCL over xGMI; backend 'nccl' on ROCm,
'gloo' in the CPU tests).  The reference is single-process; this is new design.

Reduction rule: parameters  P <- P0 + sum_g (P_g - P0)   (sum of per-replica deltas:
tracks the single-stream trajectory to first order, whereas a plain mean divides the item
displacement by the world size); RMSProp slots  ms <- mean_g ms_g.
User rows are only ever updated by their owning rank; they are combined once, after training.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


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
    content tables).  ``begin()`` snapshots, ``end()`` all-reduces and writes back."""

    def __init__(self, engine, names=None):
        self.eng = engine
        self.names = names or getattr(engine, 'replicated_names', ('V', 'b'))
        self.start = None

    def begin(self):
        self.start = {n: self.eng.get(n)[0].clone() for n in self.names}

    def end(self):
        """ONE collective per exchange: the parameter deltas and the slots (pre-divided by the world size) of every
        replicated table travel in one flat buffer -- xGMI all-reduces of a few MB are latency-bound, so four
        separate calls would cost four ring set-ups per epoch."""
        _, w = world()
        if w == 1:
            return
        cur = {n: self.eng.get(n) for n in self.names}
        parts = []
        for n in self.names:
            p, ms = cur[n]
            parts.append((p - self.start[n]).reshape(-1))
            parts.append((ms / w).reshape(-1))
        flat = torch.cat(parts)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        new, off = {}, 0
        for n in self.names:
            p, ms = cur[n]
            m = p.numel()
            new[n] = (self.start[n] + flat[off:off + m].view_as(p), flat[off + m:off + 2 * m].view_as(ms))
            off += 2 * m
        self.eng.set_replicated(new)


def combine_user_rows(current: torch.Tensor, start: torch.Tensor) -> torch.Tensor:
    """after training: every user row was changed by at most one rank -> sum of deltas is exact"""
    _, w = world()
    if w == 1:
        return current
    return reduce_deltas(current, start)
