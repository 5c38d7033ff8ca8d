"""ctypes binding of libtkr_hip.so (C ABI in include/tkr.h).

PyTorch is plumbing here: it owns device memory and the stream; every compute call goes
through the C ABI with raw device pointers.  There is NO fallback: if the library is
missing or a call fails, this module raises.
"""
from __future__ import annotations

import ctypes as C
import warnings
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('TKR_HIP_LIB') or os.path.join(_HERE, 'libtkr_hip.so')      # the override is for A/B builds of the kernels (scripts/)

_lib = None
VERSION = 118          # TKR_VERSION of include/tkr.h this binding was written against


class TkrError(RuntimeError):
    pass


class StepGaveUp(TkrError):
    """a bounded spin of a persistent step kernel ran out: the tables of the run are invalid, the engine has stepped down to the
    next kernel (K2o -> K2f -> K2) and works again once valid tables are put back (BPR.train does: single/bpr.py _run_epoch)"""


class BprState(C.Structure):
    """mirror of tkr_bpr_state (include/tkr.h)"""
    _fields_ = [('U', C.c_void_p), ('msU', C.c_void_p),
                ('V', C.c_void_p), ('msV', C.c_void_p), ('b', C.c_void_p), ('msb', C.c_void_p),
                ('n_users', C.c_int32), ('n_items', C.c_int32), ('k', C.c_int32), ('mode', C.c_int32),
                ('lu', C.c_float), ('li', C.c_float), ('lj', C.c_float), ('lb', C.c_float),
                ('lr', C.c_float), ('rho', C.c_float), ('eps', C.c_float), ('opt', C.c_int32)]


class PlanCall(C.Structure):
    """tkr_plan_call of include/tkr.h: the arguments of tkr_sample_plan_owned"""
    _fields_ = [(n, C.c_void_p) for n in ('tr_users', 'row_ptr', 'pos_cols', 'cols_sorted', 'ucnt', 'icnt', 'touch_u', 'touch_i', 'out_u', 'out_i',
                                           'out_j', 'task', 'occ', 'occt', 'prec', 'pocc', 'ohdr')] + \
               [('seed', C.c_uint64), ('first_triplet', C.c_uint64)] + \
               [(n, C.c_int32) for n in ('n_tr', 'n_users', 'n_items', 'n_batches', 'batch_size', 'n_owner', 'ohdr_stride', 'reserved')]


def plan_call(csr, n_users, n_items, seed, B, cnt, plan):
    """-> PlanCall for K1 into `plan` (an owner-ordered dataflow plan buffer); first_triplet / n_batches are the caller's to set"""
    assert plan.owners > 0
    for t in (csr.tr_users, csr.row_ptr, csr.pos_cols, csr.cols_sorted, cnt.ucnt, cnt.icnt, cnt.touch_u, cnt.touch_i, plan.u, plan.task):
        assert t.is_cuda and t.is_contiguous()
    assert cnt.ucnt.numel() == n_users and cnt.icnt.numel() == n_items
    pc = PlanCall()
    for name, t in (('tr_users', csr.tr_users), ('row_ptr', csr.row_ptr), ('pos_cols', csr.pos_cols), ('cols_sorted', csr.cols_sorted),
                    ('ucnt', cnt.ucnt), ('icnt', cnt.icnt), ('touch_u', cnt.touch_u), ('touch_i', cnt.touch_i), ('out_u', plan.u),
                    ('out_i', plan.i), ('out_j', plan.j), ('task', plan.task), ('occ', plan.occ), ('occt', plan.occt), ('prec', plan.prec),
                    ('pocc', plan.pocc), ('ohdr', plan.ohdr)):
        setattr(pc, name, t.data_ptr())
    pc.seed, pc.n_tr, pc.n_users, pc.n_items, pc.batch_size = seed, int(csr.tr_users.numel()), n_users, n_items, B
    pc.n_owner, pc.ohdr_stride = plan.owners, plan.cap
    pc.keep = (csr, cnt)                  # the tensors behind the pointers live as long as the struct
    return pc


class FlowState(C.Structure):
    """mirror of tkr_flow_state (include/tkr.h): granule tables of the persistent dataflow step"""
    _fields_ = [(n, C.c_void_p) for n in ('U', 'msU', 'tailU', 'rdU', 'V', 'msV', 'tailV', 'rdV')] + \
               [(n, C.c_int32) for n in ('n_users', 'n_items', 'k', 'mode')] + \
               [(n, C.c_float) for n in ('lu', 'li', 'lj', 'lb', 'lr', 'rho', 'eps')] + [('opt', C.c_int32), ('item_bufs', C.c_int32)]


class VbprState(C.Structure):
    """mirror of tkr_vbpr_state (include/tkr.h)"""
    _fields_ = [(n, C.c_void_p) for n in ('U', 'msU', 'I', 'msI', 'irb', 'msirb', 'cem', 'mscem', 'icb', 'msicb', 'feat')] + \
               [(n, C.c_int32) for n in ('n_users', 'n_items', 'kh', 'd', 'mode')] + \
               [(n, C.c_float) for n in ('lu', 'li', 'lj', 'lb', 'le', 'lr', 'rho', 'eps')] + \
               [(n, C.c_void_p) for n in ('f_ptr', 'f_col', 'f_val', 'c_ptr', 'c_item', 'c_val', 'item_tag')]


EXPORTS = ('tkr_version', 'tkr_plan_team', 'tkr_plan_max_blocks', 'tkr_sample_plan', 'tkr_sample_plan_owned', 'tkr_plan_rollback', 'tkr_bpr_run', 'tkr_bpr_flow_run', 'tkr_bpr_own_run', 'tkr_bpr_own_run_between', 'tkr_bpr_own_plan_run', 'tkr_bpr_own_owners', 'tkr_bpr_own_owners_shared', 'tkr_flow_row_granules', 'tkr_flow_ctl_words',
           'tkr_vbpr_run', 'tkr_vbpr_colplan', 'tkr_vbpr_run_cols', 'tkr_build_rated_mask', 'tkr_score_topk', 'tkr_count_hits', 'tkr_calib_rowcopy',
           'tkr_idmap_create', 'tkr_idmap_destroy', 'tkr_ratings_parse', 'tkr_ratings_sizes', 'tkr_ratings_copy',
           'tkr_ratings_destroy', 'tkr_matrix_read', 'tkr_matrix_sizes', 'tkr_matrix_copy', 'tkr_matrix_destroy',
           'tkr_matrix_write', 'tkr_raw_ranks', 'tkr_count_hits_rr', 'tkr_topk_set_math', 'tkr_vbpr_set_pairs', 'tkr_lab_build',
           'tkr_sync_snapshot', 'tkr_sync_pack', 'tkr_sync_unpack', 'tkr_sync_flow_snapshot', 'tkr_sync_flow_pack',
           'tkr_sync_flow_unpack')
EXPORTS_I64 = ('tkr_vbpr_workspace_floats', 'tkr_vbpr_colplan_lds_bytes', 'tkr_topk_workspace_bytes_for', 'tkr_topk_workspace_bytes', 'tkr_plan_workspace_bytes')


def lib():
    """Load libtkr_hip.so once; raise if it has not been built (python __graft_entry__.py build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TkrError('%s not found: build it with `make -C top-k-rec_amd/csrc` '
                           '(or __graft_entry__.build()); there is no CPU fallback' % LIB_PATH)
        _lib = C.CDLL(LIB_PATH)
        for name in EXPORTS:
            getattr(_lib, name).restype = C.c_int
        for name in EXPORTS_I64:
            getattr(_lib, name).restype = C.c_int64
        if _lib.tkr_version() != VERSION:
            raise TkrError('%s is version %d, this binding expects %d: rebuild it (make -C top-k-rec_amd/csrc)'
                           % (LIB_PATH, _lib.tkr_version(), VERSION))
    return _lib


def _check(rc, what):
    if rc != 0:
        kind = 'hipError_t' if rc > 0 else 'tkr error'
        raise TkrError('%s failed: %s %d' % (what, kind, rc))


def _p(t):
    if t is None:
        return C.c_void_p(0)
    assert t.is_cuda and t.is_contiguous(), 'device-resident contiguous tensor required'
    return C.c_void_p(t.data_ptr())


def _call(name, anchor, *args):
    """lib().<name>(*args, stream) on the device that owns tensor `anchor` and on torch's current stream OF THAT
    DEVICE: a kernel launched while another device is current would run there against this device's pointers"""
    dev = anchor.device.index
    prev = torch.cuda.current_device()
    if prev != dev:
        torch.cuda.set_device(dev)
    try:
        _check(getattr(lib(), name)(*args, C.c_void_p(torch.cuda.current_stream(anchor.device).cuda_stream)), name)
    finally:
        if prev != dev:
            torch.cuda.set_device(prev)


def version():
    return lib().tkr_version()


PLAN_MAX_BATCHES = 512      # per tkr_sample_plan call (16 bitmap words per row)


def plan_team(B):
    return lib().tkr_plan_team(C.c_int32(B))


def plan_workspace_bytes(B, n_batches):
    return int(lib().tkr_plan_workspace_bytes(C.c_int32(B), C.c_int32(n_batches)))


def plan_max_blocks(B):
    return lib().tkr_plan_max_blocks(C.c_int32(B))


def sample_plan(csr, n_users, n_items, seed, first_triplet, n_batches, B, cnt, plan, ctl=None):
    """csr: tensors tr_users,row_ptr,pos_cols,cols_sorted; cnt: ucnt,icnt,touch_u,touch_i;
    plan: u,i,j,task,occ,occt + either rec,hdr,tpar (one launch per batch, K2/K3) or prec,pocc (dataflow form, K2f; with
    plan.owners > 0 and plan.ohdr the owner-ordered form of K2o); all int32 device tensors."""
    assert n_batches <= PLAN_MAX_BATCHES
    assert plan.u.numel() >= n_batches * B and plan.task.numel() >= n_batches * 3 * B * 4
    prec, pocc = getattr(plan, 'prec', None), getattr(plan, 'pocc', None)
    if getattr(plan, 'owners', 0) > 0:
        assert ctl is None and prec.numel() >= n_batches * 3 * B * 32 and pocc.numel() >= n_batches * 3 * B * 4
        assert plan.ohdr.numel() >= plan.owners * plan.cap and plan.cap >= n_batches
        _call('tkr_sample_plan_owned', plan.u, _p(csr.tr_users), C.c_int32(csr.tr_users.numel()), _p(csr.row_ptr), _p(csr.pos_cols),
              _p(csr.cols_sorted), C.c_int32(n_users), C.c_int32(n_items), C.c_uint64(seed), C.c_uint64(first_triplet),
              C.c_int32(n_batches), C.c_int32(B), _p(cnt.ucnt), _p(cnt.icnt), _p(cnt.touch_u), _p(cnt.touch_i), _p(plan.u), _p(plan.i),
              _p(plan.j), _p(plan.task), _p(plan.occ), _p(plan.occt), _p(prec), _p(pocc), C.c_int32(plan.owners), _p(plan.ohdr),
              C.c_int32(plan.cap))
        return
    ws = getattr(plan, 'ws', None)                      # device scratch of the grid-wide planner (B > 8192)
    assert ws is None or ws.numel() >= plan_workspace_bytes(B, n_batches)
    assert B <= 8192 or ws is not None
    if prec is not None:
        assert prec.numel() >= n_batches * 3 * B * 32 and pocc.numel() >= n_batches * 3 * B * 4
    else:
        assert plan.rec.numel() >= n_batches * plan_max_blocks(B) * plan_team(B) * 16 and plan.hdr.numel() >= n_batches * 4
    assert cnt.ucnt.numel() == n_users and cnt.touch_u.numel() == n_users * 16
    assert cnt.icnt.numel() == n_items and cnt.touch_i.numel() == n_items * 16
    _call('tkr_sample_plan', plan.u, _p(csr.tr_users), C.c_int32(csr.tr_users.numel()), _p(csr.row_ptr), _p(csr.pos_cols),
                                 _p(csr.cols_sorted), C.c_int32(n_users), C.c_int32(n_items), C.c_uint64(seed),
                                 C.c_uint64(first_triplet), _p(ctl), C.c_int32(n_batches), C.c_int32(B),
                                 _p(cnt.ucnt), _p(cnt.icnt), _p(cnt.touch_u), _p(cnt.touch_i),
                                 _p(plan.u), _p(plan.i), _p(plan.j), _p(plan.task), _p(plan.occ), _p(getattr(plan, 'rec', None)),
                                 _p(getattr(plan, 'hdr', None)), _p(plan.occt), _p(getattr(plan, 'tpar', None)), _p(prec), _p(pocc),
                                 _p(ws), C.c_int64(ws.numel() if ws is not None else 0))


_PLAN_ARGTYPES = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_uint64, C.c_uint64, C.c_void_p,
                  C.c_int32, C.c_int32] + [C.c_void_p] * 15 + [C.c_void_p, C.c_int64, C.c_void_p]


def plan_caller(csr, n_users, n_items, seed, B, cnt, plan):
    """-> call(first_triplet, n_batches): sample_plan with everything that does not change between calls checked and marshalled
    ONCE (a short run_batches call is ~150 us of device time; building ~30 ctypes arguments and their assertions per call sat in
    front of its first launch with the GPU idle)."""
    sample_plan.__doc__                                  # same contract
    prec, pocc = getattr(plan, 'prec', None), getattr(plan, 'pocc', None)
    ws = getattr(plan, 'ws', None)
    assert B <= 8192 or ws is not None
    assert cnt.ucnt.numel() == n_users and cnt.touch_u.numel() == n_users * 16
    assert cnt.icnt.numel() == n_items and cnt.touch_i.numel() == n_items * 16
    cap = plan.u.numel() // B
    fn = lib().tkr_sample_plan
    fn.argtypes = _PLAN_ARGTYPES
    ptr = lambda t: None if t is None else t.data_ptr()
    for t in (csr.tr_users, csr.row_ptr, csr.pos_cols, csr.cols_sorted, cnt.ucnt, cnt.icnt, cnt.touch_u, cnt.touch_i, plan.u, plan.task):
        assert t.is_cuda and t.is_contiguous()
    fixed_a = (ptr(csr.tr_users), int(csr.tr_users.numel()), ptr(csr.row_ptr), ptr(csr.pos_cols), ptr(csr.cols_sorted), n_users, n_items, seed)
    fixed_b = (ptr(cnt.ucnt), ptr(cnt.icnt), ptr(cnt.touch_u), ptr(cnt.touch_i), ptr(plan.u), ptr(plan.i), ptr(plan.j), ptr(plan.task), ptr(plan.occ),
               ptr(getattr(plan, 'rec', None)), ptr(getattr(plan, 'hdr', None)), ptr(plan.occt), ptr(getattr(plan, 'tpar', None)), ptr(prec), ptr(pocc),
               ptr(ws), int(ws.numel()) if ws is not None else 0)
    device = plan.u.device
    keep = (csr, cnt)                                    # these live as long as the closure; the closure itself is kept ON the plan buffer
                                                         # (no reference back to it: a cycle would delay the release of a replaced buffer)
    owners = getattr(plan, 'owners', 0)
    if owners > 0:                                       # the owner-ordered dataflow form (K2o)
        fo = lib().tkr_sample_plan_owned
        fo.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_uint64, C.c_uint64, C.c_int32,
                       C.c_int32] + [C.c_void_p] * 12 + [C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]
        tail = (ptr(cnt.ucnt), ptr(cnt.icnt), ptr(cnt.touch_u), ptr(cnt.touch_i), ptr(plan.u), ptr(plan.i), ptr(plan.j), ptr(plan.task),
                ptr(plan.occ), ptr(plan.occt), ptr(prec), ptr(pocc), owners, ptr(plan.ohdr), plan.cap)

        def call_owned(first_triplet, n_batches):
            assert 0 < n_batches <= min(cap, PLAN_MAX_BATCHES) and keep
            prev = torch.cuda.current_device()
            if prev != device.index:
                torch.cuda.set_device(device.index)
            try:
                rc = fo(*fixed_a, first_triplet, n_batches, B, *tail, torch.cuda.current_stream(device).cuda_stream)
            finally:
                if prev != device.index:
                    torch.cuda.set_device(prev)
            if rc:
                _check(rc, 'tkr_sample_plan_owned')
        return call_owned

    def call(first_triplet, n_batches):
        assert 0 < n_batches <= min(cap, PLAN_MAX_BATCHES) and keep
        if torch.cuda.current_device() != device.index:
            prev = torch.cuda.current_device()
            torch.cuda.set_device(device.index)
            try:
                rc = fn(*fixed_a, first_triplet, None, n_batches, B, *fixed_b, torch.cuda.current_stream(device).cuda_stream)
            finally:
                torch.cuda.set_device(prev)
            if rc:
                _check(rc, 'tkr_sample_plan')
            return
        rc = fn(*fixed_a, first_triplet, None, n_batches, B, *fixed_b, torch.cuda.current_stream(device).cuda_stream)
        if rc:
            _check(rc, 'tkr_sample_plan')
    return call


def plan_rollback(plan, B, first_batch, n_batches, cnt):
    """take batches [first_batch, first_batch + n_batches) of a plan out of the update counters again"""
    _call('tkr_plan_rollback', plan.task, _p(plan.task), C.c_int32(B), C.c_int32(first_batch), C.c_int32(n_batches),
                                   _p(cnt.ucnt), _p(cnt.icnt))


def _at(t, offset):
    """device pointer `offset` elements into tensor t (None -> NULL)"""
    if t is None:
        return C.c_void_p(0)
    assert t.is_cuda and t.is_contiguous() and 0 <= offset <= t.numel()
    return C.c_void_p(t.data_ptr() + offset * t.element_size())


def bpr_run(state, plan, B, n_batches, loss_out=None, first=0):
    """batches [first, first + n_batches) of a plan"""
    rs = plan_max_blocks(B) * plan_team(B) * 16
    _call('tkr_bpr_run', plan.rec, C.byref(state), _at(plan.rec, first * rs), _at(plan.occ, first * 6 * B), _at(plan.hdr, first * 4),
                             C.c_int32(B), C.c_int32(n_batches), _at(loss_out, first))


FLOW_MAX_K = 256         # K2f holds a row in 2 x (k / 128) registers per lane: csrc/bpr_flow.hip


def flow_row_granules(k):
    return int(lib().tkr_flow_row_granules(C.c_int32(k)))


def flow_ctl_words():
    return int(lib().tkr_flow_ctl_words())


FLOW_CTL_ARRIVE, FLOW_CTL_STATUS, FLOW_CTL_SPINS = 1024, 1026, 1027      # TKR_FLOW_CTL_* of include/tkr.h (32 ticket counters, 32 words apart, come first)
FLOW_CTL_DEBUG, FLOW_CTL_PROF = 1032, 1056       # post-mortem of a timed-out wait (16 words); TKR_FLOW_PROFILE=1 cycle sums (8 x uint64)


def bpr_flow_run(state, plan, B, n_batches, ctl, loss_out=None, first=0, waves_per_cu=0):
    """batches [first, first + n_batches) of a dataflow plan in ONE persistent launch"""
    _call('tkr_bpr_flow_run', plan.prec, C.byref(state), _at(plan.prec, first * 3 * B * 32), _p(plan.pocc), C.c_int32(B),
          C.c_int32(n_batches), _p(ctl), _p(loss_out), C.c_int32(waves_per_cu))


def flow_stepper(state, B, ctl, waves_per_cu=0):
    """-> step(plan, first, n_batches, loss_out): bpr_flow_run with everything that does not change between calls bound once
    (a 20-batch call is ~70 us of device time: the argument marshalling of the general path is a tenth of that)"""
    fn = lib().tkr_bpr_flow_run
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    device, st = ctl.device, C.addressof(state)
    ctl_ptr, rec_bytes = ctl.data_ptr(), 3 * B * 32 * 4

    def step(plan, first, n_batches, loss_out):
        if torch.cuda.current_device() != device.index:            # the guarded path of _call
            return bpr_flow_run(state, plan, B, n_batches, ctl, loss_out, first, waves_per_cu)
        rc = fn(st, plan.prec.data_ptr() + first * rec_bytes, plan.pocc.data_ptr(), B, n_batches, ctl_ptr,
                None if loss_out is None else loss_out.data_ptr(), waves_per_cu, torch.cuda.current_stream(device).cuda_stream)
        if rc:
            _check(rc, 'tkr_bpr_flow_run')
    step.state = state                                              # the struct lives as long as the closure
    return step


def bpr_own_owners(n_items, k, device=None, share=1):
    """workgroups (= owners of item rows) of the persistent step K2o on the current device when `share` processes split its CUs,
    0 = the rows do not fit the owners' LDS"""
    if device is not None and torch.cuda.current_device() != device.index:
        with torch.cuda.device(device):
            return int(lib().tkr_bpr_own_owners_shared(C.c_int32(n_items), C.c_int32(k), C.c_int32(share)))
    return int(lib().tkr_bpr_own_owners_shared(C.c_int32(n_items), C.c_int32(k), C.c_int32(share)))


def bpr_own_run(state, plan, B, n_batches, ctl, loss_out=None, first=0, owner_waves=0):
    """batches [first, first + n_batches) of an owner-ordered dataflow plan in ONE persistent launch of K2o"""
    plan.epoch += 1                       # no two launches on one plan buffer's scalar slots share an epoch
    _call('tkr_bpr_own_run', plan.prec, C.byref(state), _p(plan.prec), _p(plan.pocc), _p(plan.occt), _p(plan.ohdr), C.c_int32(plan.cap), C.c_int32(plan.owners),
          C.c_int32(B), C.c_int32(first), C.c_int32(n_batches), _p(ctl), _p(loss_out), C.c_int32(owner_waves), _p(plan.xch),
          C.c_uint32(plan.epoch & 0xffffffff or 1))


def own_stepper(state, B, ctl, owner_waves=0):
    """-> step(plan, first, n_batches, loss_out, events=None): bpr_own_run with the fixed arguments bound once (as flow_stepper);
    events = (before, after): two torch events that have been recorded once (so they exist), recorded around the launch in C"""
    fn = lib().tkr_bpr_own_run_between
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                   C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_uint32, C.c_void_p]
    device, st, ctl_ptr = ctl.device, C.addressof(state), ctl.data_ptr()

    def step(plan, first, n_batches, loss_out, events=None):
        if torch.cuda.current_device() != device.index:
            assert events is None
            return bpr_own_run(state, plan, B, n_batches, ctl, loss_out, first, owner_waves)
        plan.epoch += 1
        rc = fn(None if events is None else events[0].cuda_event, None if events is None else events[1].cuda_event,
                st, plan.prec.data_ptr(), plan.pocc.data_ptr(), plan.occt.data_ptr(), plan.ohdr.data_ptr(), plan.cap, plan.owners, B, first, n_batches,
                ctl_ptr, None if loss_out is None else loss_out.data_ptr(), owner_waves, plan.xch.data_ptr(), plan.epoch & 0xffffffff or 1,
                torch.cuda.current_stream(device).cuda_stream)
        if rc:
            _check(rc, 'tkr_bpr_own_run_between')
    fused = lib().tkr_bpr_own_plan_run
    fused.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_uint32, C.c_void_p,
                      C.c_void_p, C.c_void_p]

    raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)          # the stream's handle without a Stream object around it (~1 us)
    dev_index = device.index

    def plan_and_run(plan, call, first, n_batches, loss_out, events=None):
        """K1 of the chunk described by `call` (a PlanCall into `plan`) and the step on its batches [first, first + n_batches): one C call"""
        assert torch.cuda.current_device() == dev_index and 0 < call.n_batches <= min(plan.cap, PLAN_MAX_BATCHES)
        plan.epoch += 1
        stream = raw_stream(dev_index) if raw_stream is not None else torch.cuda.current_stream(device).cuda_stream
        rc = fused(C.addressof(call), st, first, n_batches, ctl_ptr, None if loss_out is None else loss_out.data_ptr(), owner_waves,
                   plan.xch.data_ptr(), plan.epoch & 0xffffffff or 1, None if events is None else events[0].cuda_event,
                   None if events is None else events[1].cuda_event, stream)
        if rc:
            _check(rc, 'tkr_bpr_own_plan_run')
    step.state = state
    step.takes_events = True
    step.plan_and_run = plan_and_run
    step.assigns_loss = not ((owner_waves >> 8) & 0x80)      # the row-read form writes loss_out[b] (csrc/bpr_own.hip own_loss_kernel); the scalar form adds
    return step


def set_vbpr_pairs(mode):
    """where tkr_vbpr_run_cols forms a batch's pair sums: 0 own launch, 1 every task for itself, 2 the first blocks of the update launch (include/tkr.h)"""
    _check(lib().tkr_vbpr_set_pairs(C.c_int32(int(mode))), 'tkr_vbpr_set_pairs')


def vbpr_workspace_floats(B, kh, d):
    return int(lib().tkr_vbpr_workspace_floats(C.c_int32(B), C.c_int32(kh), C.c_int32(d)))


def vbpr_run(state, plan, B, n_batches, workspace, loss_out=None, first=0):
    rs = plan_max_blocks(B) * plan_team(B) * 16
    _call('tkr_vbpr_run', plan.rec, C.byref(state), _at(plan.i, first * B), _at(plan.j, first * B), _at(plan.rec, first * rs),
                              _at(plan.occ, first * 6 * B), _at(plan.hdr, first * 4), _at(plan.occt, first * 3 * B),
                              _at(plan.u, first * B), _at(getattr(plan, 'tpar', None), first * B), C.c_int32(B),
                              C.c_int32(n_batches), _p(workspace), _at(loss_out, first))


def vbpr_colplan_lds_bytes(B, d):
    return int(lib().tkr_vbpr_colplan_lds_bytes(C.c_int32(B), C.c_int32(d)))


def vbpr_colplan(sparse, d, plan, B, n_batches, row_cap):
    """K1-side preparation of the column-plan VBPR step: plan.colh / cent / tcnt / tent from plan.i / plan.j and the CSR of feat"""
    tcap = 2 * row_cap
    assert plan.colh.numel() >= n_batches * d * 8 and plan.cent.numel() >= n_batches * B * tcap * 2
    assert plan.tcnt.numel() >= n_batches * B and plan.tent.numel() >= n_batches * B * tcap * 2
    _call('tkr_vbpr_colplan', plan.i, _p(sparse['f_ptr']), _p(sparse['f_col']), _p(sparse['f_val']), C.c_int32(d), _p(plan.i), _p(plan.j),
          C.c_int32(B), C.c_int32(n_batches), C.c_int32(row_cap), _p(plan.colh), _p(plan.cent), _p(plan.tcnt), _p(plan.tent))


def vbpr_run_cols(state, plan, B, n_batches, workspace, d, row_cap, cols_per_block=0, loss_out=None, first=0):
    rs = plan_max_blocks(B) * plan_team(B) * 16
    ent = B * 2 * row_cap * 2
    _call('tkr_vbpr_run_cols', plan.rec, C.byref(state), _at(plan.i, first * B), _at(plan.j, first * B), _at(plan.rec, first * rs),
          _at(plan.occ, first * 6 * B), _at(plan.hdr, first * 4), _at(plan.occt, first * 3 * B), _at(plan.u, first * B),
          _at(plan.tpar, first * B), _at(plan.colh, first * d * 8), _at(plan.cent, first * ent), _at(plan.tcnt, first * B),
          _at(plan.tent, first * ent), C.c_int32(row_cap), C.c_int32(cols_per_block), C.c_int32(B), C.c_int32(n_batches), _p(workspace),
          _at(loss_out, first))


# ---- K4 / K5 -------------------------------------------------------------------------------------
def build_rated_mask(rated_ptr, rated_cols, n_rows, n_cols):
    """CSR (int64 ptr, int32 ascending cols; both on device) -> bitmask [ceil(n_cols/32)][pitch] uint32"""
    assert rated_ptr.dtype == torch.int64 and rated_cols.dtype == torch.int32
    pitch = (n_rows + 31) // 32 * 32
    mask = torch.zeros(((n_cols + 31) // 32) * pitch, dtype=torch.int32, device=rated_ptr.device)
    if rated_cols.numel() == 0:
        rated_cols = torch.zeros(1, dtype=torch.int32, device=rated_ptr.device)
    _call('tkr_build_rated_mask', mask, _p(rated_ptr), _p(rated_cols), C.c_int32(n_rows), C.c_int32(n_cols), _p(mask), C.c_int32(pitch))
    return mask, pitch


_topk_ws = {}


def _topk_workspace(n_rows, K, device, n_cols=0, k=0):
    """cached scratch for the item-range split and the pre-converted item factors (grows on demand, per device)"""
    need = int(lib().tkr_topk_workspace_bytes_for(C.c_int32(n_rows), C.c_int32(n_cols), C.c_int32(k), C.c_int32(K)))
    ws = _topk_ws.get(device)
    if ws is None or ws.numel() < need:
        ws = _topk_ws[device] = torch.empty(need, dtype=torch.uint8, device=device)
    return ws


TOPK_MAX_K = 32      # one launch of K4 ranks at most this many columns per row


def _score_topk_once(U, Vt, K, bias, user_idx, mask, mask_pitch, want_scores, split):
    n_rows = int(user_idx.numel()) if user_idx is not None else int(U.shape[0])
    ids = torch.empty((n_rows, K), dtype=torch.int32, device=U.device)
    scores = torch.empty((n_rows, K), dtype=torch.float32, device=U.device) if want_scores else None
    ws = _topk_workspace(n_rows, K, U.device, int(Vt.shape[0]), int(U.shape[1])) if split else None
    _call('tkr_score_topk', U, _p(U), _p(user_idx), C.c_int32(n_rows), _p(Vt), _p(bias), C.c_int32(Vt.shape[0]),
                                C.c_int32(U.shape[1]), _p(mask), C.c_int32(mask_pitch), C.c_int32(K), _p(ids),
                                _p(scores), _p(ws), C.c_int64(ws.numel() if ws is not None else 0))
    return ids, scores


TOPK_MATH_DEFAULT = 'refine'


def set_topk_math(mode):
    """'refine' (default: one scaled fp16 pass with a rigorous error bound picks the candidates, the arithmetic of 'fp32' ranks
    them -- same lists and score bits as 'fp32'; k <= 128), 'bf16x3' (split products on the dense matrix pipe, k <= 128) or
    'fp32' (fp32 MFMA) -- see include/tkr.h"""
    _check(lib().tkr_topk_set_math(C.c_int32({'bf16x3': 0, 'fp32': 1, 'refine': 2}[mode])), 'tkr_topk_set_math')


def lab():
    """True when the loaded library was built with `make LAB=1` (csrc/Makefile): it then also holds the kernel forms that were measured
    and dropped (K2o scalar exchange / scout / 16 waves / loader ring, K4 bf16x3, the VBPR pair-sum placements 1 and 2)"""
    return bool(lib().tkr_lab_build())


def score_topk(U, Vt, K, bias=None, user_idx=None, mask=None, mask_pitch=0, want_scores=False, split=True):
    """-> ids int32 [n_rows, K] (and scores fp32 [n_rows, K]).

    K > 32 (evaluate.py -t above 32) is served exactly by several launches: the best 32, then the best 32
    of what is left (the columns already found are added to a copy of the rated mask), and so on."""
    assert U.dtype == torch.float32 and Vt.dtype == torch.float32 and U.shape[1] == Vt.shape[1]
    if U.shape[1] > 768 and not getattr(score_topk, '_warned_wide', False):
        score_topk._warned_wide = True
        warnings.warn('K4: factor width %d is above 768, where a workgroup\'s users no longer stay resident as MFMA operands: the generic '
                      'form runs (one wave per row, every lane gathers its own item row: csrc/topk.hip score_topk_wide_kernel)' % U.shape[1])
    if K <= TOPK_MAX_K:
        ids, scores = _score_topk_once(U, Vt, K, bias, user_idx, mask, mask_pitch, want_scores, split)
        return (ids, scores) if want_scores else ids
    n_rows = int(user_idx.numel()) if user_idx is not None else int(U.shape[0])
    n_cols = int(Vt.shape[0])
    pitch = mask_pitch if mask is not None else (n_rows + 31) // 32 * 32
    work = mask.clone() if mask is not None else torch.zeros(((n_cols + 31) // 32) * pitch, dtype=torch.int32, device=U.device)
    ptr = torch.arange(0, (n_rows + 1) * TOPK_MAX_K, TOPK_MAX_K, dtype=torch.int64, device=U.device)
    parts_i, parts_s, left = [], [], K
    while left > 0:
        ids, scores = _score_topk_once(U, Vt, TOPK_MAX_K, bias, user_idx, work, pitch, want_scores, split)
        take = min(left, TOPK_MAX_K)
        parts_i.append(ids[:, :take])
        if want_scores:
            parts_s.append(scores[:, :take])
        left -= take
        if left > 0:      # found columns join the mask (negative ids = padding are skipped by the kernel)
            _call('tkr_build_rated_mask', work, _p(ptr), _p(ids.reshape(-1)), C.c_int32(n_rows), C.c_int32(n_cols), _p(work), C.c_int32(pitch))
    ids = torch.cat(parts_i, dim=1).contiguous()
    return (ids, torch.cat(parts_s, dim=1).contiguous()) if want_scores else ids


def count_hits(ids, like_ptr, like_cols, step, interval):
    """-> int64[interval]: hits per bucket as evaluate.py accumulates them (cumulative over buckets)"""
    assert ids.dtype == torch.int32 and like_ptr.dtype == torch.int64 and like_cols.dtype == torch.int32
    first = torch.zeros(max(interval, 1), dtype=torch.int64, device=ids.device)
    if like_cols.numel() == 0:
        like_cols = torch.zeros(1, dtype=torch.int32, device=ids.device)
    _call('tkr_count_hits', ids, _p(ids), C.c_int32(ids.shape[0]), C.c_int32(ids.shape[1]), _p(like_ptr), _p(like_cols),
                                C.c_int32(step), C.c_int32(interval), _p(first))
    return torch.cumsum(first[:interval], 0)


def raw_ranks(U, Vt, ids, rated_ptr, rated_cols, bias=None, user_idx=None):
    """K6 -> int32 [n_rows, K]: rank of every kept column among ALL columns of its row (utils.py:113)"""
    assert ids.dtype == torch.int32 and rated_ptr.dtype == torch.int64 and rated_cols.dtype == torch.int32
    n_rows, K = int(ids.shape[0]), int(ids.shape[1])
    out = torch.empty((n_rows, K), dtype=torch.int32, device=ids.device)
    if rated_cols.numel() == 0:
        rated_cols = torch.zeros(1, dtype=torch.int32, device=ids.device)
    _call('tkr_raw_ranks', ids, _p(U), _p(user_idx), C.c_int32(n_rows), _p(Vt), _p(bias), C.c_int32(U.shape[1]), _p(rated_ptr),
                               _p(rated_cols), _p(ids), C.c_int32(K), _p(out))
    return out


def count_hits_rr(ids, raw_rank, like_ptr, like_cols, step, interval):
    """K7 -> (hits int64[interval], trrs float64[interval]) accumulated over buckets like utils.py:115-117"""
    n_rows = int(ids.shape[0])
    hit = torch.zeros((n_rows, max(interval, 1)), dtype=torch.int32, device=ids.device)
    rr = torch.zeros((n_rows, max(interval, 1)), dtype=torch.float64, device=ids.device)
    if like_cols.numel() == 0:
        like_cols = torch.zeros(1, dtype=torch.int32, device=ids.device)
    if interval > 0:
        _call('tkr_count_hits_rr', ids, _p(ids), _p(raw_rank), C.c_int32(n_rows), C.c_int32(ids.shape[1]), _p(like_ptr),
                                       _p(like_cols), C.c_int32(step), C.c_int32(interval), _p(hit), _p(rr))
    return torch.cumsum(hit.sum(0, dtype=torch.int64)[:interval], 0), torch.cumsum(rr.sum(0)[:interval], 0)


# ---- per-epoch exchange of replicated tables (csrc/sync.hip) ----------------------------------------
def sync_snapshot(P, cnt, start, n, w):
    _call('tkr_sync_snapshot', P, _p(P), _p(cnt), _p(start), C.c_int64(n), C.c_int32(w))


def sync_pack(P, ms, cnt, start, flat_delta, flat_ms, n, w, inv_world):
    _call('tkr_sync_pack', P, _p(P), _p(ms), _p(cnt), _p(start), _p(flat_delta), _p(flat_ms), C.c_int64(n), C.c_int32(w),
                               C.c_float(inv_world))


def sync_flow_snapshot(V, tailV, icnt, start, n, k, item_bufs=2):
    _call('tkr_sync_flow_snapshot', V, _p(V), _p(tailV), _p(icnt), _p(start), C.c_int32(n), C.c_int32(k), C.c_int32(item_bufs))


def sync_flow_pack(V, msV, tailV, icnt, start, flat_delta, flat_ms, n, k, inv_world, item_bufs=2):
    _call('tkr_sync_flow_pack', V, _p(V), _p(msV), _p(tailV), _p(icnt), _p(start), _p(flat_delta), _p(flat_ms), C.c_int32(n),
          C.c_int32(k), C.c_float(inv_world), C.c_int32(item_bufs))


def sync_flow_unpack(V, msV, tailV, rdV, icnt, start, flat_delta, flat_ms, n, k, item_bufs=2):
    _call('tkr_sync_flow_unpack', V, _p(V), _p(msV), _p(tailV), _p(rdV), _p(icnt), _p(start), _p(flat_delta), _p(flat_ms),
          C.c_int32(n), C.c_int32(k), C.c_int32(item_bufs))


def sync_unpack(P, ms, start, flat_delta, flat_ms, n, w):
    _call('tkr_sync_unpack', P, _p(P), _p(ms), _p(start), _p(flat_delta), _p(flat_ms), C.c_int64(n), C.c_int32(w))
