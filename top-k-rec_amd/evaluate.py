"""evaluate.py -- accuracy@k of matrix-factorisation models, scored on MI355X.

Same command line and output as the reference's evaluate.py (flags :48-55, one stdout line per
scenario ``S,%.6f,...`` :113-117):

    python evaluate.py -d data -m embed/bpr -f 0 -s 5 -t 30 -sl im om

Same inputs: ``DATA/uid``, ``DATA/vid``, ``DATA/f{fold}tr.txt``, ``MODEL/final-U.dat``,
``final-V.dat``, optional ``final-B.dat``, ``DATA/f{fold}te.{S}.idl`` and ``.txt``.

What runs where: the text files are parsed by the native host parsers of libtkr_hip.so
(textio.py; one pass per file, flat arrays instead of dicts of sets -- SURVEY.md §8f n2; the
'%f ' matrices keep a binary ``.npy`` copy beside them, n1); the scores, the rated-item filter,
the top-``total`` selection (K4) and the hit counting (K5) run on the GPU -- the
[n_users, n_items] score matrix and its argsort (evaluate.py:78-81) are never materialised.

Stated differences from the reference:
  * ``final-B.dat`` is honoured for every scenario by gathering the bias per test id; the
    reference adds the whole vid-ordered bias row (:79-80), which raises unless the scenario's
    id list equals ``vid`` -- where it is defined, results are identical.
  * tie order among exactly equal scores is defined (higher test column first); the reference
    inherits numpy's unspecified unstable order.
  * only users that appear in the scenario's test file with at least one like are ranked
    (the reference scores every user and then reads only those rows).
  * under a torch.distributed launcher (one process per GPU) the test lines of every scenario are block-sharded over the
    ranks and the hit counters / like counts all-reduced; rank 0 prints.  The reference is single-process.
  * ids are resolved to indices while parsing, so the rated set of a user is keyed by the uid's
    index and a test column by its index in the id list; the reference keys both by token.  The
    two differ only if an id file repeats a token (the re-pointing quirk of evaluate.py:5-10).
"""
from __future__ import annotations

import argparse
import os

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')     # multi-process GPU work on this driver needs dmabuf IPC; must be set before
                                                             # the HIP runtime starts, i.e. before the first torch.cuda call
import numpy as np
import torch

import textio
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
    every = textio.read_matrix(path)
    rows = np.unique(np.fromiter(ids.values(), dtype=np.int64, count=len(ids)))
    out = np.zeros((len(ids), every.shape[1]), dtype=np.float32)
    out[rows] = every[rows]
    return out


class Scenario:
    """One test scenario as flat arrays.  ``users`` [n] = uid index of every test line with >= 1 like (file order),
    ``like_ptr``/``like_cols``: its liked test columns, ascending (evaluate.py:84-95); ``rated_ptr``/``rated_cols``:
    the test columns on the user's train line, like 0 or 1 (evaluate.py:30-45,98); ``tcount`` = sum of |likes|."""

    def __init__(self, teids, users, like_ptr, like_cols, rated_ptr, rated_cols):
        self.teids, self.users = teids, users
        self.like_ptr, self.like_cols, self.rated_ptr, self.rated_cols = like_ptr, like_cols, rated_ptr, rated_cols
        self.tcount = int(like_ptr[-1])


def _group(rows, cols, n_rows, n_cols):
    """unique (row, col) pairs -> CSR over rows with ascending cols"""
    key = np.unique(rows.astype(np.int64) * n_cols + cols)
    ptr = np.zeros(n_rows + 1, dtype=np.int64)
    np.cumsum(np.bincount(key // n_cols, minlength=n_rows), out=ptr[1:])
    return ptr, (key % n_cols).astype(np.int32)


def load_scenario(data_dir, fold, scenario, uids, umap=None):
    umap = umap or textio.IdMap(uids)
    teids = read_ids(os.path.join(data_dir, 'f%dte.%s.idl' % (fold, scenario)))
    temap = textio.IdMap(teids)
    n_te = max(len(teids), 1)
    te_path = os.path.join(data_dir, 'f%dte.%s.txt' % (fold, scenario))
    T = textio.parse_ratings(te_path, umap, temap)
    liked = T.like == 1
    if np.any(liked & (T.item < 0)):
        raise KeyError('%s likes an id that is not in the scenario id list' % te_path)        # teids[vid], :93
    lptr_all, lcols = _group(T.entry_line[liked], T.item[liked], len(T.line_user), n_te)
    lines = np.flatnonzero(np.diff(lptr_all) > 0)                                           # len(likes) != 0, :95
    users = T.line_user[lines].astype(np.int64)
    if np.any(users < 0):
        raise KeyError('%s: test user missing from the uid list' % te_path)                   # uids[uid], :98
    like_ptr = np.zeros(len(lines) + 1, dtype=np.int64)
    np.cumsum(np.diff(lptr_all)[lines], out=like_ptr[1:])
    # history: the LAST train line of a user is its rated set (rated[uid] = set() per line, :34)
    hist_path = os.path.join(data_dir, 'f%dtr.txt' % fold)
    H = textio.parse_ratings(hist_path, umap, temap)
    n_users = max(int(max(uids.values())) + 1 if uids else 0, 1)
    last = np.full(n_users, -1, dtype=np.int64)
    known = np.flatnonzero(H.line_user >= 0)
    np.maximum.at(last, H.line_user[known], known)
    hl = last[users]
    if np.any(hl < 0):
        raise KeyError('%s: test user without a line in %s' % (te_path, hist_path))           # rated[uid], :98
    seg = H.line_ptr[hl + 1] - H.line_ptr[hl]
    row = np.repeat(np.arange(len(lines), dtype=np.int64), seg)
    pos = np.arange(int(seg.sum()), dtype=np.int64) - np.repeat(np.cumsum(seg) - seg, seg) + np.repeat(H.line_ptr[hl], seg)
    item = H.item[pos]
    keep = item >= 0                                                                         # only test columns matter
    rated_ptr, rated_cols = _group(row[keep], item[keep], len(lines), n_te)
    return Scenario(teids, users, like_ptr, lcols, rated_ptr, rated_cols)


def rank_scenario(umat_dev, vmat, bmat, vids, sc, total, device, want_scores=False):
    """filtered top-`total` test columns of every test line -> int32 [n_lines, total] (device)"""
    te_rows = np.zeros(len(sc.teids), dtype=np.int64)
    for vid, col in sc.teids.items():
        te_rows[col] = vids[vid]                                     # evaluate.py:75-77
    Vt = torch.from_numpy(np.ascontiguousarray(vmat[te_rows])).to(device)
    bias = None
    if bmat is not None:
        bias = torch.from_numpy(np.ascontiguousarray(bmat.reshape(-1)[te_rows])).to(device)
    user_idx = torch.from_numpy(sc.users.astype(np.int32)).to(device)
    rptr, rcols = torch.from_numpy(sc.rated_ptr).to(device), torch.from_numpy(sc.rated_cols).to(device)
    mask, pitch = tkr_hip.build_rated_mask(rptr, rcols, len(sc.users), len(sc.teids))
    return tkr_hip.score_topk(umat_dev, Vt, total, bias=bias, user_idx=user_idx, mask=mask, mask_pitch=pitch,
                              want_scores=want_scores)


def shard_scenario(sc, rank, world):
    """the contiguous block of test lines rank `rank` of `world` ranks scores (users are independent: SURVEY.md §8e)"""
    n = len(sc.users)
    lo, hi = rank * n // world, (rank + 1) * n // world
    lp, rp = sc.like_ptr, sc.rated_ptr
    return Scenario(sc.teids, sc.users[lo:hi], lp[lo:hi + 1] - lp[lo], sc.like_cols[lp[lo]:lp[hi]],
                    rp[lo:hi + 1] - rp[lo], sc.rated_cols[rp[lo]:rp[hi]])


def _world():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def evaluate_scenario(umat_dev, vmat, bmat, uids, vids, data_dir, fold, scenario, step, total, device, umap=None):
    """accuracy@(step, 2*step, ...) of one scenario (evaluate.py:72-112).  Under torch.distributed every rank ranks its
    block of the scenario's test lines; what the reference accumulates over users -- tresults[k] and tcount, :106-112 --
    is summed over the ranks by one all-reduce of interval + 1 integers (no other exchange: the factors are replicated)."""
    full = load_scenario(data_dir, fold, scenario, uids, umap)
    return evaluate_loaded(umat_dev, vmat, bmat, vids, full, step, total, device)


def evaluate_loaded(umat_dev, vmat, bmat, vids, full, step, total, device):
    """the ranking + counting half of evaluate_scenario on an already parsed Scenario (the same on every rank)"""
    rank, world = _world()
    sc = shard_scenario(full, rank, world) if world > 1 else full
    interval = total // step
    hits = np.zeros(interval, dtype=np.int64)
    if len(sc.users):
        ids = rank_scenario(umat_dev, vmat, bmat, vids, sc, total, device)
        lptr, lcols = torch.from_numpy(sc.like_ptr).to(device), torch.from_numpy(sc.like_cols).to(device)
        hits = tkr_hip.count_hits(ids, lptr, lcols, step, interval).cpu().numpy()
    tcount = sc.tcount
    if world > 1:
        import torch.distributed as dist
        on_dev = dist.get_backend() == 'nccl'                        # RCCL reduces device tensors, gloo host tensors
        acc = torch.from_numpy(np.r_[hits, tcount].astype(np.int64))
        acc = acc.to(device) if on_dev else acc
        dist.all_reduce(acc, op=dist.ReduceOp.SUM)
        acc = acc.cpu().numpy()
        hits, tcount = acc[:interval], int(acc[interval])
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
    started_group = _init_distributed()
    rank, _ = _world()
    device = torch.device('cuda', torch.cuda.current_device())
    uids = read_ids(os.path.join(args.data, 'uid'))
    vids = read_ids(os.path.join(args.data, 'vid'))
    umat = read_matrix(os.path.join(args.model, 'final-U.dat'), uids)
    vmat = read_matrix(os.path.join(args.model, 'final-V.dat'), vids)
    bmat = None
    if os.path.exists(os.path.join(args.model, 'final-B.dat')):
        bmat = read_matrix(os.path.join(args.model, 'final-B.dat'), vids)
    umat_dev = torch.from_numpy(umat).to(device)
    umap = textio.IdMap(uids)
    results = {}
    for scenario in args.scenarios:
        acc = evaluate_scenario(umat_dev, vmat, bmat, uids, vids, args.data, args.fold, scenario, args.step, args.total, device, umap)
        if scenario not in results:                                  # evaluate.py:109-112 ACCUMULATES per scenario name: a scenario
            results[scenario] = [0.0] * len(acc)                     # listed twice ("-sl im im") prints doubled values, twice
        for k, v in enumerate(acc):
            results[scenario][k] += v
    lines = []
    for scenario in args.scenarios:
        lines.append(scenario + ''.join(',%.6f' % v for v in results[scenario]))
        if rank == 0:                                                # one report, like the single process
            print(lines[-1])
    if started_group:
        import torch.distributed as dist
        dist.destroy_process_group()
    return lines


def _init_distributed():
    """One process per GPU when started by a launcher (python -m torch.distributed.run --nproc-per-node N evaluate.py ...):
    RANK / WORLD_SIZE / LOCAL_RANK in the environment.  Backend 'nccl' (= RCCL over xGMI) unless TKR_DIST_BACKEND says
    otherwise; TKR_SINGLE_DEVICE=1 puts every rank on GPU 0 (tests on a one-GPU box, with gloo).  Returns True when this
    call created the process group."""
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world <= 1 or not dist.is_available() or dist.is_initialized():
        return False
    local = 0 if os.environ.get('TKR_SINGLE_DEVICE') == '1' else int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local)
    backend = os.environ.get('TKR_DIST_BACKEND', 'nccl')
    if backend == 'nccl':
        dist.init_process_group('nccl', device_id=torch.device('cuda', local))
    else:
        dist.init_process_group(backend)
    return True


if __name__ == '__main__':
    main()
