"""Helpers and file formats of the BPR/VBPR path -- same names, arguments and results as the
reference's ``utils.py`` so callers switch over unchanged:

    tprint                  utils.py:6-7
    get_id_dict_from_file   utils.py:10-16   id token -> index (line order)
    get_data_from_file      utils.py:58-70   positive (uid, iid) pairs of a ratings file
    get_embed_from_file     utils.py:28-44   '%f ' text matrix -> fp32 array
    export_embed_to_file    utils.py:47-55   fp32 array -> '%f ' text matrix

The text formats stay authoritative; matrices are parsed and written by the native host code of
libtkr_hip.so (textio.py) with a binary ``.npy`` copy beside the text.  The reference's unused helpers (``get_score``,
``evaluate``, ``get_history_from_file``, ``get_iv_dict_from_file``) have no callers on this
path and are not provided.
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
