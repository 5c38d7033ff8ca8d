"""Helpers and file formats of the BPR/VBPR path -- same names, arguments and results as the
reference's ``utils.py`` so callers switch over unchanged:

    tprint                  utils.py:6-7
    get_id_dict_from_file   utils.py:10-16   id token -> index (line order)
    get_data_from_file      utils.py:58-70   positive (uid, iid) pairs of a ratings file
    get_iv_dict_from_file   utils.py:18-24   line number -> id token
    get_embed_from_file     utils.py:28-44   '%f ' text matrix -> fp32 array
    export_embed_to_file    utils.py:47-55   fp32 array -> '%f ' text matrix
    get_history_from_file   utils.py:73-89   uid -> rated ids, iid -> number of likes
    get_score               utils.py:92-98   scores of every user against a sub id list (held on the GPU)
    evaluate                utils.py:101-127 hits and reciprocal-rank sums per bucket (K4 + K6 + K7)

The text formats stay authoritative; matrices are parsed and written by the native host code of
libtkr_hip.so (textio.py) with a binary ``.npy`` copy beside the text.  ``get_score`` returns a handle to the factors on the GPU instead of the dense [n_users, n_sub] matrix and
``evaluate`` ranks from it with the fused kernels: the score matrix and its argsort (utils.py:97,105) are never
materialised (``np.asarray(score)`` still works for callers that want the matrix).
"""
from __future__ import annotations

import os
from datetime import datetime

import numpy as np

import textio


def tprint(msg: str) -> None:
    print('%s: %s' % (datetime.now().strftime('%Y-%m-%d %H:%M:%S.%f'), msg))


def get_id_dict_from_file(file_path: str) -> dict:
    """token -> current dict size when its line is read (a repeated token is re-pointed and
    the size does not grow, exactly like the reference); missing file -> {}."""
    table: dict = {}
    if os.path.isfile(file_path):
        with open(file_path, 'r') as fh:
            for token in fh:
                table[token.strip()] = len(table)
    return table


def get_data_from_file(file_path: str, uids: dict, iids: dict) -> list:
    """[(uid, iid)] for every 'iid:1' field of every known user, in file order."""
    found = []
    if os.path.isfile(file_path):
        with open(file_path, 'r') as fh:
            for record in fh:
                head, *fields = record.strip().split(',')
                if not fields or head not in uids:
                    continue
                for field in fields:
                    parts = field.split(':')
                    if parts[1] == '1' and parts[0] in iids:
                        found.append((head, parts[0]))
    return found


def get_embed_from_file(file_path: str, ids: dict = None):
    """Rows of a '%f ' text matrix, addressed by ``ids`` values (or every line) -> fp32.  Parsed by the native
    reader (textio.read_matrix), which keeps a binary copy ``<file>.npy`` beside the authoritative text."""
    if not os.path.isfile(file_path):
        return None
    every = textio.read_matrix(file_path)
    if ids is None:
        return every if len(every) else None
    if not ids:
        return None
    rows = np.unique(np.fromiter(ids.values(), dtype=np.int64, count=len(ids)))
    out = np.zeros((len(ids), every.shape[1]), dtype=np.float32)
    out[rows, :] = every[rows]
    return out


def export_embed_to_file(file_path: str, embed) -> None:
    """One line per row, every element '%f' followed by a space (trailing space kept) -- byte-identical to
    utils.py:47-55, written by the native writer."""
    folder = os.path.dirname(file_path)
    if not os.path.isdir(folder):
        os.mkdir(folder)
    embed = np.asarray(embed)
    n_rows, n_cols = embed.shape
    textio.write_matrix(file_path, embed)


def get_iv_dict_from_file(file_path: str) -> dict:
    """line number -> stripped token; missing file -> {}."""
    ivt: dict = {}
    if os.path.isfile(file_path):
        with open(file_path) as fh:
            for token in fh:
                ivt[len(ivt)] = token.strip()
    return ivt


def get_history_from_file(file_path: str):
    """(uid -> set of every iid on the user's last line, iid -> number of fields liking it with the literal '1');
    missing file -> ({}, {})."""
    browsed: dict = {}
    counter: dict = {}
    if os.path.isfile(file_path):
        with open(file_path) as fh:
            for record in fh:
                head, *fields = record.strip().split(',')
                seen = browsed[head] = set()
                for field in fields:
                    parts = field.split(':')
                    seen.add(parts[0])
                    if parts[1] == '1':
                        counter[parts[0]] = counter.get(parts[0], 0) + 1
    return browsed, counter


class Score:
    """What ``get_score`` returns: user factors and the sub-list item factors resident on the GPU.  ``evaluate``
    ranks from it; ``np.asarray(score)`` / ``score[...]`` materialise the dense matrix for other callers
    (a plain device matmul, outside the measured path)."""

    def __init__(self, U, subV, device):
        import torch
        self.device = device
        self.U = torch.from_numpy(np.ascontiguousarray(U, dtype=np.float32)).to(device)
        self.subV = torch.from_numpy(np.ascontiguousarray(subV, dtype=np.float32)).to(device)
        self.shape = (self.U.shape[0], self.subV.shape[0])
        self._dense = None

    def __array__(self, dtype=None, copy=None):
        if self._dense is None:
            self._dense = (self.U @ self.subV.T).cpu().numpy()
        return self._dense if dtype is None else self._dense.astype(dtype)

    def __getitem__(self, key):
        return self.__array__()[key]


def get_score(U, V, iids: dict, sub_iids: dict):
    """utils.py:92-98: V's rows re-ordered to ``sub_iids`` (ids absent from ``iids`` stay zero) against every user."""
    import torch
    import tkr_hip
    if not torch.cuda.is_available():
        raise tkr_hip.TkrError('get_score keeps the factors on the GPU for evaluate(); no MI355X is visible')
    V = np.asarray(V, dtype=np.float32)
    subV = np.zeros((len(sub_iids), V.shape[1]), dtype=np.float32)
    src = [iids[t] for t in iids if t in sub_iids]
    dst = [sub_iids[t] for t in iids if t in sub_iids]
    subV[dst, :] = V[src, :]
    return Score(U, subV, torch.device('cuda', torch.cuda.current_device()))


def evaluate(score, rated: dict, likes: dict, uids: dict, te_iids: dict, te_ivt: dict, step: int, total: int, interval: int):
    """-> (hits, trrs, count) of utils.py:101-127: for every user with likes, the first ``total`` unrated items of
    the ranking; a liked one at raw rank t (rated items counted) adds 1 and 1/(t+1) to buckets t//step .. interval-1.

    Runs on the GPU: K4 (filtered top-``total``), K6 (raw ranks), K7 (hits and reciprocal ranks).  Ties are ordered
    canonically (higher column first; the reference inherits numpy's unspecified order).  ``score`` must come from
    ``get_score`` -- a dense matrix would have to be re-ranked on the host, which this build does not do."""
    import torch
    import tkr_hip
    if not isinstance(score, Score):
        raise TypeError('evaluate() ranks on the GPU from the handle returned by get_score(); got %s' % type(score).__name__)
    dev = score.device
    users, like_rows, like_cols, rated_rows, rated_cols = [], [], [], [], []
    count = 0
    for uid in likes:
        like = likes[uid]
        if len(like) == 0:
            continue
        row = len(users)
        users.append(uids[uid])
        lc = sorted({te_iids[t] for t in like if t in te_iids})      # a like outside the list can never be ranked
        like_rows += [row] * len(lc)
        like_cols += lc
        rc = sorted({te_iids[t] for t in rated[uid] if t in te_iids})
        rated_rows += [row] * len(rc)
        rated_cols += rc
        count += len(like)
    hits, trrs = [0.0] * interval, [0.0] * interval
    if not users or interval <= 0:
        return hits, trrs, count

    def csr(rows, cols):
        ptr = np.zeros(len(users) + 1, dtype=np.int64)
        np.cumsum(np.bincount(np.asarray(rows, dtype=np.int64), minlength=len(users)), out=ptr[1:])
        return torch.from_numpy(ptr).to(dev), torch.from_numpy(np.asarray(cols, dtype=np.int32)).to(dev)

    lptr, lcols = csr(like_rows, like_cols)
    rptr, rcols = csr(rated_rows, rated_cols)
    user_idx = torch.tensor(users, dtype=torch.int32, device=dev)
    n_te = len(te_iids)
    mask, pitch = tkr_hip.build_rated_mask(rptr, rcols, len(users), n_te)
    ids = tkr_hip.score_topk(score.U, score.subV, total, user_idx=user_idx, mask=mask, mask_pitch=pitch)
    h, r = [], []
    for a in range(0, ids.shape[1], 256):                           # K6 handles 256 kept columns per launch
        part = ids[:, a:a + 256].contiguous()
        raw = tkr_hip.raw_ranks(score.U, score.subV, part, rptr, rcols, user_idx=user_idx)
        if a:                                                       # the earlier kept columns rank before these
            raw = torch.where(part >= 0, raw + a, raw)
        hh, rr = tkr_hip.count_hits_rr(part, raw, lptr, lcols, step, interval)
        h.append(hh)
        r.append(rr)
    hits = [float(x) for x in torch.stack(h).sum(0).cpu().tolist()]
    trrs = [float(x) for x in torch.stack(r).sum(0).cpu().tolist()]
    return hits, trrs, count
