"""evaluate.py -- accuracy@k of matrix-factorisation models, scored on MI355X.

Same command line and output as the reference's evaluate.py (flags :48-55, one stdout line per
scenario ``S,%.6f,...`` :113-117):

    python evaluate.py -d data -m embed/bpr -f 0 -s 5 -t 30 -sl im om

Same inputs: ``DATA/uid``, ``DATA/vid``, ``DATA/f{fold}tr.txt``, ``MODEL/final-U.dat``,
``final-V.dat``, optional ``final-B.dat``, ``DATA/f{fold}te.{S}.idl`` and ``.txt``.

What runs where: text parsing on the host; the scores, the rated-item filter, the top-``total``
selection (K4) and the hit counting (K5) on the GPU through libtkr_hip.so -- the
[n_users, n_items] score matrix and its argsort (evaluate.py:78-81) are never materialised.

Stated differences from the reference:
  * ``final-B.dat`` is honoured for every scenario by gathering the bias per test id; the
    reference adds the whole vid-ordered bias row (:79-80), which raises unless the scenario's
    id list equals ``vid`` -- where it is defined, results are identical.
  * tie order among exactly equal scores is defined (higher test column first); the reference
    inherits numpy's unspecified unstable order.
  * only users that appear in the scenario's test file with at least one like are ranked
    (the reference scores every user and then reads only those rows).
"""
from __future__ import annotations

import argparse
import os

import numpy as np
import torch

import tkr_hip


def read_ids(path):
    """id token -> index in file order (evaluate.py:5-10)"""
    table = {}
    with open(path) as fh:
        for line in fh:
            table[line.strip()] = len(table)
    return table


def read_matrix(path, ids):
    """'%f ' text matrix rows addressed through ``ids`` (evaluate.py:19-28) -> fp32 [len(ids), k]"""
    with open(path) as fh:
        rows = fh.readlines()
    out = None
    for r in sorted(set(ids.values())):
        vals = np.array(rows[r].split(), dtype=np.float32)
        if out is None:
            out = np.zeros((len(ids), vals.shape[0]), dtype=np.float32)
        out[r] = vals
    return out


def read_history(path):
    """uid -> list of every vid on the user's train line, like 0 or 1 (evaluate.py:30-45)"""
    rated = {}
    with open(path) as fh:
        for line in fh:
            head, *fields = line.strip().split(',')
            rated[head] = [f.split(':')[0] for f in fields]
    return rated


def read_test_lines(path, teids):
    """per test line with >= 1 like: (uid, sorted liked test columns) (evaluate.py:84-95)"""
    out = []
    with open(path) as fh:
        for line in fh:
            head, *fields = line.strip().split(',')
            liked = set()
            for f in fields:
                vid, like = f.split(':')[0], int(f.split(':')[1])
                if like == 1:
                    liked.add(teids[vid])
            if liked:
                out.append((head, sorted(liked)))
    return out


def _csr(lists, device):
    ptr = np.zeros(len(lists) + 1, dtype=np.int64)
    np.cumsum([len(x) for x in lists], out=ptr[1:])
    flat = np.fromiter((c for x in lists for c in x), dtype=np.int32, count=int(ptr[-1]))
    return torch.from_numpy(ptr).to(device), torch.from_numpy(flat).to(device)


def rank_scenario(umat_dev, vmat, bmat, uids, vids, rated, teids, tests, total, device, want_scores=False):
    """filtered top-`total` test columns of every test line -> int32 [n_lines, total] (device)"""
    te_rows = np.zeros(len(teids), dtype=np.int64)
    for vid, col in teids.items():
        te_rows[col] = vids[vid]                                     # evaluate.py:75-77
    Vt = torch.from_numpy(np.ascontiguousarray(vmat[te_rows])).to(device)
    bias = None
    if bmat is not None:
        bias = torch.from_numpy(np.ascontiguousarray(bmat.reshape(-1)[te_rows])).to(device)
    user_idx = torch.tensor([uids[uid] for uid, _ in tests], dtype=torch.int32, device=device)
    rated_cols = [sorted({teids[v] for v in rated[uid] if v in teids}) for uid, _ in tests]   # evaluate.py:98
    rptr, rcols = _csr(rated_cols, device)
    mask, pitch = tkr_hip.build_rated_mask(rptr, rcols, len(tests), len(teids))
    return tkr_hip.score_topk(umat_dev, Vt, total, bias=bias, user_idx=user_idx, mask=mask, mask_pitch=pitch,
                              want_scores=want_scores)


def evaluate_scenario(umat_dev, vmat, bmat, uids, vids, rated, data_dir, fold, scenario, step, total, device):
    idl = os.path.join(data_dir, 'f%dte.%s.idl' % (fold, scenario))
    teids = read_ids(idl)
    tests = read_test_lines(os.path.join(data_dir, 'f%dte.%s.txt' % (fold, scenario)), teids)
    interval = total // step
    tcount = sum(len(l) for _, l in tests)
    hits = np.zeros(interval, dtype=np.int64)
    if tests:
        ids = rank_scenario(umat_dev, vmat, bmat, uids, vids, rated, teids, tests, total, device)
        lptr, lcols = _csr([l for _, l in tests], device)
        hits = tkr_hip.count_hits(ids, lptr, lcols, step, interval).cpu().numpy()
    return [float(h) / tcount for h in hits]                          # ZeroDivisionError like evaluate.py:112


def main(argv=None):
    parser = argparse.ArgumentParser(description="Evaluate weighted matrix factorization based methods.")
    parser.add_argument('-d', '--data', required=True, help="The data path for the evaluation")
    parser.add_argument('-m', '--model', required=True, help="The work path for the model")
    parser.add_argument('-f', '--fold', type=int, default=0, help="The index of evaluation fold")
    parser.add_argument('-s', '--step', type=int, default=5, help="The number of evaluation step")
    parser.add_argument('-t', '--total', type=int, default=30, help="The number of total predictions")
    parser.add_argument('-sl', '--scenarios', nargs='+', default=None, help="The test scenario list")
    args = parser.parse_args(argv)

    if not torch.cuda.is_available():
        raise tkr_hip.TkrError('evaluate.py scores on the GPU through libtkr_hip.so; no MI355X is visible')
    device = torch.device('cuda', torch.cuda.current_device())
    uids = read_ids(os.path.join(args.data, 'uid'))
    vids = read_ids(os.path.join(args.data, 'vid'))
    rated = read_history(os.path.join(args.data, 'f%dtr.txt' % args.fold))
    umat = read_matrix(os.path.join(args.model, 'final-U.dat'), uids)
    vmat = read_matrix(os.path.join(args.model, 'final-V.dat'), vids)
    bmat = None
    if os.path.exists(os.path.join(args.model, 'final-B.dat')):
        bmat = read_matrix(os.path.join(args.model, 'final-B.dat'), vids)
    umat_dev = torch.from_numpy(umat).to(device)
    results = {}
    for scenario in args.scenarios:
        results[scenario] = evaluate_scenario(umat_dev, vmat, bmat, uids, vids, rated, args.data, args.fold,
                                              scenario, args.step, args.total, device)
    lines = []
    for scenario in args.scenarios:
        lines.append(scenario + ''.join(',%.6f' % v for v in results[scenario]))
        print(lines[-1])
    return lines


if __name__ == '__main__':
    main()
