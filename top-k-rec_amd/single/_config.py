"""Tuning switches of the device engines, parsed ONCE per engine from the environment into one validated object.

The reference has no such knobs (its step is `sess.run`, single/bpr.py:141); these choose between the HIP step kernels that replace
it and exist for A/B measurements and for the tests that pin every kernel form.  Defaults are what `BPR.train` runs.  An unknown
variable value raises ValueError naming the variable (twenty inline `os.environ` reads accepted anything)."""
from __future__ import annotations

import os
from dataclasses import dataclass, fields


def _flag(name, raw):
    if raw in ('0', '1'):
        return raw == '1'
    raise ValueError('%s=%r: expected 0 or 1' % (name, raw))


def _int_in(lo, hi):
    def parse(name, raw):
        try:
            v = int(raw, 0)
        except ValueError:
            raise ValueError('%s=%r: expected an integer in [%d, %d]' % (name, raw, lo, hi)) from None
        if not lo <= v <= hi:
            raise ValueError('%s=%d: expected an integer in [%d, %d]' % (name, v, lo, hi))
        return v
    return parse


def _choice(*allowed):
    def parse(name, raw):
        if raw not in allowed:
            raise ValueError('%s=%r: expected one of %s' % (name, raw, ', '.join(allowed)))
        return raw
    return parse


@dataclass
class Tuning:
    flow: bool = True                # TKR_FLOW: granule layout + persistent step (K2f / K2o) for small batches; 0: K2 always
    flow_max_batch: int = 512        # TKR_FLOW_MAX_BATCH: batch sizes up to this take the persistent step (measured per batch, K2f vs K2:
                                     #   64: 1.3 vs 3.9 us, 256: 2.6 vs 4.5, 512: 5.1 vs 5.2, 1024: 8.8 vs 6.8)
    flow_waves_per_cu: int = 0       # TKR_FLOW_WAVES_PER_CU: K2f waves per CU, 0 = the library default
    flow_item_bufs: int = 4          # TKR_FLOW_ITEM_BUFS: buffers per item row of the granule tables (2 or 4; include/tkr.h)
    own: str = '1'                   # TKR_OWN: 0 = never K2o, 1 = K2o where the item rows fit the owners' LDS, 2 = as 1 (kept for scripts)
    own_max_batch: int = 256         # TKR_OWN_MAX_BATCH: batch sizes up to this take K2o (K2o vs K2f per batch at the ML-10M shape, round 5:
                                     #   64: 0.79 vs 1.45 us, 128: 1.16 vs 2.01, 256: 1.95 vs 2.73, 384: 3.44 vs 3.28, 512: 5.47 vs 4.06)
    own_waves: int = 0               # TKR_OWN_WAVES: owner waves per workgroup | experiment bits 8..15, 0 = the library default
    fuse_short: bool = True          # TKR_FUSE_SHORT: K1 and the step of a short call leave in ONE C call (tkr_bpr_own_plan_run)
    fuse_plan: bool = True           # TKR_FUSE_PLAN: ... and K1 runs INSIDE the step's launch (the planner prologue of csrc/bpr_own.hip); 0: its own launches
    overlap_min_batch: int = 2048    # TKR_OVERLAP_MIN_BATCH: from this batch size on K1 of the next chunk runs on the side stream
    epoch_ahead: bool = True         # TKR_EPOCH_AHEAD: plan the first chunk after an exchange ahead of it
    call_ahead: bool = False         # TKR_CALL_AHEAD: plain layout (batch > 512): plan the NEXT call's first chunk behind the last steps of this one
                                     #   (epoch-sized calls at batch 8192: 25.5 -> 24.7 us per batch; a call nobody follows up pays for the plan: 24.1 -> 24.7)
    restart: bool = True             # TKR_RESTART: BPR.train keeps a copy of the state it starts from (371 MB at the ML-10M shape, 2.2 GB at the
                                     #   Netflix shape) so that a persistent step that gives up restarts one kernel down; 0: no copy, such a run raises
    vbpr_cols: bool = True           # TKR_VBPR_COLS: the column-plan form of the VBPR step
    vbpr_overlap: bool = True        # TKR_VBPR_OVERLAP: its column plan on the side stream


    @classmethod
    def from_env(cls, environ=None):
        """a Tuning from TKR_* variables; unset variables keep the defaults, anything unparsable raises ValueError"""
        env = os.environ if environ is None else environ
        parsers = {
            'flow': _flag, 'flow_max_batch': _int_in(0, 1 << 20), 'flow_waves_per_cu': _int_in(0, 32),
            'flow_item_bufs': lambda n, r: int(_choice('2', '4')(n, r)), 'own': _choice('0', '1', '2'),
            'own_max_batch': _int_in(0, 1024), 'own_waves': _int_in(0, 0xffff), 'fuse_short': _flag, 'fuse_plan': _flag,
            'overlap_min_batch': _int_in(1, 1 << 30), 'epoch_ahead': _flag, 'call_ahead': _flag, 'restart': _flag, 'vbpr_cols': _flag, 'vbpr_overlap': _flag,
        }
        out = cls()
        for f in fields(cls):
            if f.name.startswith('_'):
                continue
            var = 'TKR_' + f.name.upper()
            raw = env.get(var)
            if raw is not None and raw != '':
                setattr(out, f.name, parsers[f.name](var, raw))
        return out
