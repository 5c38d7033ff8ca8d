"""dev tool: randomized differential run of K1+K2 (RMSProp / SGD, l2 / l1) and K4 against the oracle over random shapes.
Not part of the test suite (minutes of oracle time); prints one line per case and exits non-zero on the first mismatch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'top-k-rec_amd')]
import numpy as np, torch
import tkr_hip
from oracle import plan_np as P, ref_np as R
from single import _engine

rng = np.random.Generator(np.random.PCG64(int(os.environ.get('SEED', 1))))
dev = torch.device('cuda', 0)
n_cases = int(os.environ.get('CASES', 40))
for case in range(n_cases):
    n_users, n_items = int(rng.integers(20, 3000)), int(rng.integers(8, 1500))
    k = int(rng.choice([1, 3, 16, 50, 64, 100, 128, 200, 256]))
    B = int(rng.choice([1, 7, 64, 256, 300, 1024, 1025, 4096, 8192]))
    nb = int(rng.integers(1, 12))
    mode, opt = str(rng.choice(['l2', 'l1'])), str(rng.choice(['rmsprop', 'sgd']))
    if opt == 'sgd':
        mode = 'l2'
    tr = {}
    for u in rng.permutation(n_users)[: max(1, n_users - int(rng.integers(0, 5)))]:
        deg = int(min(n_items - 1, rng.integers(1, 20)))
        tr[int(u)] = [int(x) for x in rng.integers(0, n_items, deg)]
    tr_users = list(tr.keys())
    hp = dict(lu=2.5e-3, li=2.5e-3, lj=2.5e-4, lb=1e-3, lr=0.03, mode=mode, opt=opt)
    seed = int(rng.integers(0, 2 ** 31))
    eng = _engine.BprEngine(n_users, n_items, k, hp, dev, seed=seed)
    ref = dict(U=eng.get('U')[0].cpu().numpy(), V=eng.get('V')[0].cpu().numpy(), b=eng.get('b')[0].cpu().numpy(),
               msU=np.ones((n_users, k), np.float32), msV=np.ones((n_items, k), np.float32), msb=np.ones(n_items, np.float32))
    row_ptr, pos, srt = P.build_csr(tr, n_users)
    csr = _engine.TrainingCSR.from_arrays(row_ptr, pos, np.asarray(tr_users, np.int32), dev)
    eng.run_batches(csr, nb, B, want_loss=False)
    torch.cuda.synchronize()
    u, i, j = P.sample_triplets(tr_users, row_ptr, pos, srt, n_items, seed, 0, nb * B)
    np.testing.assert_array_equal(eng.plan.u.cpu().numpy()[: nb * B], u)
    for b in range(nb):
        R.bpr_step(ref, u[b * B:(b + 1) * B], i[b * B:(b + 1) * B], j[b * B:(b + 1) * B], hp)
    for name in ('U', 'V', 'b'):
        np.testing.assert_allclose(eng.get(name)[0].cpu().numpy(), ref[name], rtol=3e-4, atol=2e-5, err_msg='%s case %d' % (name, case))
    # K4 on small-integer factors derived from the trained ones: exact id lists
    K = int(rng.integers(1, 33))
    qU = np.clip(np.round(ref['U'] * 256), -8, 8).astype(np.float32) / 8
    qV = np.clip(np.round(ref['V'] * 256), -8, 8).astype(np.float32) / 8
    rated = [sorted(set(tr.get(x, []))) for x in range(n_users)]
    ptr = np.zeros(n_users + 1, np.int64); np.cumsum([len(x) for x in rated], out=ptr[1:])
    mask, pitch = tkr_hip.build_rated_mask(torch.from_numpy(ptr).to(dev), torch.tensor([c for x in rated for c in x] or [0], dtype=torch.int32, device=dev)[: max(1, int(ptr[-1]))], n_users, n_items)
    for math in ('bf16x3', 'fp32'):
        tkr_hip.set_topk_math(math)
        ids = tkr_hip.score_topk(torch.from_numpy(qU).to(dev), torch.from_numpy(qV).to(dev), K, mask=mask, mask_pitch=pitch).cpu().numpy()
        s = np.dot(qU, qV.T)
        for x in range(0, n_users, max(1, n_users // 50)):
            want = R.filtered_topk(s[x], set(rated[x]), K, canonical=True)
            assert [c for c in ids[x].tolist() if c >= 0] == want, (case, math, x)
    tkr_hip.set_topk_math('bf16x3')
    print('case %2d ok: users %4d items %4d k %3d B %4d nb %2d %s %s K %2d' % (case, n_users, n_items, k, B, nb, mode, opt, K), flush=True)
print('all %d cases match' % n_cases)
