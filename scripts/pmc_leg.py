"""the headline leg alone, for a counter pass: `rocprofv3 --kernel-trace --pmc <C> -- python scripts/pmc_leg.py k B shape batches`
runs `batches` (+512 of warm-up) batches through the engine's persistent step and prints how many batches its launches covered.
bench.py spawns it (live `roofline.traffic`); scripts/collect_profile.sh does the same with bench.py itself."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'top-k-rec_amd')]
import torch
import bench

k, B, shape, n = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
dev = torch.device('cuda', 0)
r, csr, eng, nnz = bench.build_problem(shape, k, 0, 1, dev)
eng.run_batches(csr, 512, B, want_loss=bench.WANT_LOSS)
eng.run_batches(csr, n, B, want_loss=bench.WANT_LOSS)
torch.cuda.synchronize()
eng.check()
print(json.dumps({'batches': 512 + n, 'kernel': bench.step_kernel(eng, B)[0]}))
