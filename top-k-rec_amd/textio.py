"""Bulk readers/writers of the reference's text formats on top of the native host parsers of
libtkr_hip.so (csrc/textio.hip).  No GPU is needed for anything here.

    IdMap(dict)                       the reference's token -> index dict, handed to the parser
    parse_ratings(path, users, items) "uid,iid:like,..." lines -> flat arrays (Ratings)   (+ stamped .csr.npz copy)
    read_matrix(path)                 '%f ' text matrix -> fp32 [lines, cols]   (+ stamped .npy copy)
    write_matrix(path, array)         fp32 array -> '%f ' text, byte-identical to utils.py:47-55

The text files stay authoritative.  ``read_matrix`` keeps a binary copy ``<path>.npy`` of what
it parsed and ``parse_ratings`` a copy ``<path>.csr.npz`` of its flat arrays (n1).  A copy is used
only when TKR_NO_CACHE is unset and the STAMP stored inside it -- byte size and mtime_ns of the
text file it was made from (ratings: also a digest of the two id tables) -- equals the text file's
current one; anything else (an older or newer text, a copy made with ``cp -p``, a coarse-mtime file
system, other id lists) is a miss and the text is parsed again.  ``write_matrix`` refreshes the copy
by parsing back the text it just wrote, so a cached read returns exactly what a fresh parse would.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

import tkr_hip


class TextFormatError(ValueError):
    """a line the reference itself would raise on (field without ':', non-integer like, ragged matrix row)"""


def _check(rc, what, path=None):
    if rc == 0:
        return
    if rc == -3:
        raise OSError('%s: cannot open %s' % (what, path))
    if rc == -4:
        raise TextFormatError('%s: malformed line in %s' % (what, path))
    raise tkr_hip.TkrError('%s failed: %d' % (what, rc))


class IdMap:
    """native token -> index table built from a reference-style dict (keys: str without newline)"""

    def __init__(self, table: dict):
        keys = list(table.keys())
        if any((not isinstance(t, str)) or ('\n' in t) for t in keys):
            raise TextFormatError('id tokens must be strings without newlines')
        blob = '\n'.join(keys).encode()
        index = np.fromiter(table.values(), dtype=np.int32, count=len(keys))
        import zlib
        self.digest = '%08x%08x%d' % (zlib.crc32(blob), zlib.crc32(index.tobytes()), len(keys))     # of tokens AND indices
        self._h = C.c_void_p()
        _check(tkr_hip.lib().tkr_idmap_create(blob, C.c_int64(len(blob)), index.ctypes.data_as(C.c_void_p),
                                              C.c_int64(len(keys)), C.byref(self._h)), 'tkr_idmap_create')
        self.size = len(keys)

    def __del__(self):
        if getattr(self, '_h', None) and tkr_hip is not None:      # (module globals are gone at interpreter shutdown)
            tkr_hip.lib().tkr_idmap_destroy(self._h)
            self._h = None


class Ratings:
    """one record per line of a ratings file: ``line_user`` [n_lines] (index or -1), ``line_ptr`` [n_lines+1],
    per entry ``item`` (index or -1) and ``like`` (int)"""

    def __init__(self, line_user, line_ptr, item, like):
        self.line_user, self.line_ptr, self.item, self.like = line_user, line_ptr, item, like

    @property
    def entry_line(self):
        return np.repeat(np.arange(len(self.line_user), dtype=np.int64), np.diff(self.line_ptr))

    @property
    def entry_user(self):
        return np.repeat(self.line_user, np.diff(self.line_ptr))


def _stamp(path, extra=''):
    st = os.stat(path)
    return '%d:%d:%s' % (st.st_size, st.st_mtime_ns, extra)


def parse_ratings(path: str, users, items) -> Ratings:
    users = users if isinstance(users, IdMap) else IdMap(users)
    items = items if isinstance(items, IdMap) else IdMap(items)
    cache = path + '.csr.npz'
    stamp = _stamp(path, users.digest + '/' + items.digest) if _cache_enabled() else None
    if stamp is not None and os.path.isfile(cache):
        try:
            with np.load(cache, allow_pickle=False) as z:
                if str(z['stamp']) == stamp:
                    return Ratings(z['line_user'], z['line_ptr'], z['item'], z['like'])
        except (OSError, ValueError, KeyError):
            pass
    out = _parse_ratings(path, users, items)
    # the copy is stored under the stamp taken BEFORE the parse, and only if the file still carries it afterwards (a text rewritten
    # in between would otherwise get a copy of the old contents under the new stamp: ADVICE r2); one writer per launch
    if stamp is not None and _cache_writer() and _stamp(path, users.digest + '/' + items.digest) == stamp:
        try:
            tmp = cache + '.tmp.%d.npz' % os.getpid()
            np.savez(tmp, stamp=np.array(stamp), line_user=out.line_user, line_ptr=out.line_ptr, item=out.item, like=out.like)
            os.replace(tmp, cache)
        except OSError:
            pass                               # read-only data directory: the text stays the only copy
    return out


def _parse_ratings(path, users, items) -> Ratings:
    lib = tkr_hip.lib()
    h = C.c_void_p()
    _check(lib.tkr_ratings_parse(os.fsencode(path), users._h, items._h, C.byref(h)), 'tkr_ratings_parse', path)
    try:
        n_lines, n_entries = C.c_int64(), C.c_int64()
        _check(lib.tkr_ratings_sizes(h, C.byref(n_lines), C.byref(n_entries)), 'tkr_ratings_sizes')
        line_user = np.empty(n_lines.value, dtype=np.int32)
        line_ptr = np.empty(n_lines.value + 1, dtype=np.int64)
        item = np.empty(n_entries.value, dtype=np.int32)
        like = np.empty(n_entries.value, dtype=np.int32)
        _check(lib.tkr_ratings_copy(h, *(a.ctypes.data_as(C.c_void_p) for a in (line_user, line_ptr, item, like))),
               'tkr_ratings_copy')
    finally:
        lib.tkr_ratings_destroy(h)
    return Ratings(line_user, line_ptr, item, like)


def _parse_matrix(path):
    lib = tkr_hip.lib()
    h = C.c_void_p()
    _check(lib.tkr_matrix_read(os.fsencode(path), C.byref(h)), 'tkr_matrix_read', path)
    try:
        rows, cols = C.c_int64(), C.c_int64()
        _check(lib.tkr_matrix_sizes(h, C.byref(rows), C.byref(cols)), 'tkr_matrix_sizes')
        out = np.empty((rows.value, cols.value), dtype=np.float32)
        _check(lib.tkr_matrix_copy(h, out.ctypes.data_as(C.c_void_p)), 'tkr_matrix_copy')
    finally:
        lib.tkr_matrix_destroy(h)
    return out


def _cache_path(path):
    return path + '.npy'


def _cache_enabled():
    return os.environ.get('TKR_NO_CACHE', '') in ('', '0')


def _cache_writer():
    """under a torch.distributed launcher every rank parses the same files: rank 0 alone writes the copies"""
    return os.environ.get('RANK', '0') == '0'


def read_matrix(path: str) -> np.ndarray:
    """every line of a '%f ' text matrix -> fp32 [n_lines, n_cols]"""
    cache = _cache_path(path)
    if _cache_enabled() and os.path.isfile(cache) and os.path.isfile(cache + '.stamp'):
        try:
            if open(cache + '.stamp').read() == _stamp(path):
                got = np.load(cache)
                if got.dtype == np.float32 and got.ndim == 2:
                    return got
        except (OSError, ValueError):
            pass
    stamp = _stamp(path) if _cache_enabled() else None
    out = _parse_matrix(path)
    _store_cache(path, out, stamp)
    return out


def _store_cache(path, parsed, stamp):
    """binary copy + the stamp (size, mtime_ns) the text carried BEFORE it was parsed -- stored only if it still carries it;
    the stamp file goes last, so a torn pair is a miss"""
    if not _cache_enabled() or not _cache_writer():
        return
    try:
        if stamp is None or _stamp(path) != stamp:
            return
        if os.path.exists(_cache_path(path) + '.stamp'):
            os.remove(_cache_path(path) + '.stamp')
        tmp = _cache_path(path) + '.tmp.%d' % os.getpid()
        with open(tmp, 'wb') as fh:
            np.save(fh, parsed)
        os.replace(tmp, _cache_path(path))
        with open(tmp, 'w') as fh:
            fh.write(stamp)
        os.replace(tmp, _cache_path(path) + '.stamp')
    except OSError:
        pass                                   # read-only data directory: the text stays the only copy


def write_matrix(path: str, array) -> None:
    array = np.ascontiguousarray(array, dtype=np.float32)
    if array.ndim != 2:
        raise ValueError('2-D array required')
    _check(tkr_hip.lib().tkr_matrix_write(os.fsencode(path), array.ctypes.data_as(C.c_void_p),
                                          C.c_int64(array.shape[0]), C.c_int64(array.shape[1])), 'tkr_matrix_write', path)
    if _cache_enabled():
        stamp = _stamp(path)
        _store_cache(path, _parse_matrix(path), stamp)      # the 6-decimal text is authoritative: cache what IT says
