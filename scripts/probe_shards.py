"""bench.py's shards_on_one_gpu leg for ONE shard count (for rocprofv3 --kernel-trace).  usage: probe_shards.py S"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'top-k-rec_amd'))
import torch, synth
import bench
S = int(sys.argv[1]) if len(sys.argv) > 1 else 2
r = synth.make_ratings(seed=42, **dict(synth.ML10M))
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
import dist as tdist
orig = tdist.batches_per_rank
class _Only:      # run the leg's loop for this S only
    pass
src = bench.shards_on_one_gpu
import types, inspect
code = inspect.getsource(src).replace('for S in (2, 4, 8):', 'for S in (%d,):' % S)
ns = dict(bench.__dict__)
exec(code, ns)
out = ns['shards_on_one_gpu'](r, 128, dev, 256, 127e6)
print(json.dumps(out['S%d' % S]))
