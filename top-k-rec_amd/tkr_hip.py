"""ctypes binding of libtkr_hip.so (C ABI in include/tkr.h).

PyTorch is plumbing here: it owns device memory and the stream; every compute call goes
through the C ABI with raw device pointers.  There is NO fallback: if the library is
missing or a call fails, this module raises.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libtkr_hip.so')

_lib = None


class TkrError(RuntimeError):
    pass


class BprState(C.Structure):
    """mirror of tkr_bpr_state (include/tkr.h)"""
    _fields_ = [('U', C.c_void_p), ('msU', C.c_void_p), ('ustamp', C.c_void_p),
                ('V', C.c_void_p), ('msV', C.c_void_p), ('b', C.c_void_p), ('msb', C.c_void_p),
                ('istamp', C.c_void_p),
                ('n_users', C.c_int32), ('n_items', C.c_int32), ('k', C.c_int32), ('mode', C.c_int32),
                ('lu', C.c_float), ('li', C.c_float), ('lj', C.c_float), ('lb', C.c_float),
                ('lr', C.c_float), ('rho', C.c_float), ('eps', C.c_float)]


class VbprState(C.Structure):
    """mirror of tkr_vbpr_state (include/tkr.h)"""
    _fields_ = [('ure', C.c_void_p), ('ms_ure', C.c_void_p), ('uce', C.c_void_p), ('ms_uce', C.c_void_p),
                ('ustamp', C.c_void_p),
                ('ire', C.c_void_p), ('ms_ire', C.c_void_p), ('irb', C.c_void_p), ('ms_irb', C.c_void_p),
                ('istamp', C.c_void_p),
                ('cem', C.c_void_p), ('ms_cem', C.c_void_p), ('icb', C.c_void_p), ('ms_icb', C.c_void_p),
                ('feat', C.c_void_p),
                ('n_users', C.c_int32), ('n_items', C.c_int32), ('kh', C.c_int32), ('d', C.c_int32),
                ('mode', C.c_int32),
                ('lu', C.c_float), ('li', C.c_float), ('lj', C.c_float), ('lb', C.c_float), ('le', C.c_float),
                ('lr', C.c_float), ('rho', C.c_float), ('eps', C.c_float)]


EXPORTS = ('tkr_version', 'tkr_sample_plan', 'tkr_bpr_step', 'tkr_bpr_run')


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


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def version():
    return lib().tkr_version()


def sample_plan(tr_users, row_ptr, pos_cols, cols_sorted, n_items, seed, first_triplet, n_batches, B,
                out_u, out_i, out_j, task, occ, ctl=None):
    for t in (tr_users, row_ptr, pos_cols, cols_sorted, out_u, out_i, out_j, task, occ):
        assert t.dtype == torch.int32
    assert out_u.numel() >= n_batches * B and task.numel() >= n_batches * 3 * B * 4 and occ.numel() >= n_batches * 3 * B * 2
    _check(lib().tkr_sample_plan(_p(tr_users), C.c_int32(tr_users.numel()), _p(row_ptr), _p(pos_cols),
                                 _p(cols_sorted), C.c_int32(n_items), C.c_uint64(seed), C.c_uint64(first_triplet),
                                 _p(ctl), C.c_int32(n_batches), C.c_int32(B), _p(out_u), _p(out_i), _p(out_j),
                                 _p(task), _p(occ), _stream()), 'tkr_sample_plan')


def bpr_step(state, task, occ, B, serial, loss_out=None):
    _check(lib().tkr_bpr_step(C.byref(state), _p(task), _p(occ), C.c_int32(B), C.c_int32(serial), _p(loss_out),
                              _stream()), 'tkr_bpr_step')


def bpr_run(state, task, occ, B, n_batches, first_serial, loss_out=None):
    _check(lib().tkr_bpr_run(C.byref(state), _p(task), _p(occ), C.c_int32(B), C.c_int32(n_batches),
                             C.c_int32(first_serial), _p(loss_out), _stream()), 'tkr_bpr_run')
