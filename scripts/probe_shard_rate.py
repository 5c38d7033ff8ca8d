"""dev probe: the persistent BPR step on rank 0's user shard of a 1-, 2-, 4-, 8-rank run (one GPU, no collective): does the per-rank rate
survive the smaller shard (more repeats of a user in consecutive batches)?  Measured: 105.4 / 106.4 / 108.1 / 106.4 M triplets/s."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'top-k-rec_amd')]
import torch, bench
dev = torch.device('cuda', 0)
for world in (1, 2, 4, 8):
    r, csr, eng, nnz = bench.build_problem('ml10m', 128, 0, world, dev)
    eng.run_batches(csr, 1024, 256, want_loss=False); torch.cuda.synchronize()
    t0 = time.perf_counter(); eng.run_batches(csr, 8192, 256, want_loss=False); torch.cuda.synchronize()
    w = time.perf_counter() - t0
    print('world %d (rank 0 shard, %d users): %.2f us/batch, %.1f M triplets/s' % (world, eng.n_users, w / 8192 * 1e6, 8192 * 256 / w / 1e6), flush=True)
    del eng
