"""n1/n2 at Netflix scale: write a ~10^8-rating train file in the reference's text format (uid,iid:like,...), then time
BPR.load_training_data's native one-pass parser on it, cold and from the stamped binary copy.  Host-only (no GPU needed).
    python scripts/time_parser_nf.py [scale]      scale 1.0 = 480,189 users, ~1e8 ratings; 0.1 for a quick run"""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'top-k-rec_amd')]
import numpy as np
import synth, textio
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
spec = dict(synth.NETFLIX, n_users=int(synth.NETFLIX['n_users'] * scale))
t0 = time.time()
r = synth.make_ratings(seed=42, **spec)
n_items = r['n_in'] + r['n_out']
print('generated %d train ratings for %d users in %.1f s' % (len(r['tr_u']), spec['n_users'], time.time() - t0), flush=True)
d = tempfile.mkdtemp(prefix='tkr_nf_')
uid_names = [str(x + 1) for x in range(spec['n_users'])]
vid_names = [str(1000 + 3 * x) for x in range(n_items)]
tok = [',%s:%d' % (v, l) for v in vid_names for l in (0, 1)]          # token of (item, like) = tok[2 * item + like]
path = os.path.join(d, 'f0tr.txt')
t0 = time.time()
idx = (2 * r['tr_i'] + r['tr_l']).tolist()
cuts = np.flatnonzero(np.r_[True, r['tr_u'][1:] != r['tr_u'][:-1], True]).tolist()
users = r['tr_u'][cuts[:-1]].tolist()
with open(path, 'w') as fh:
    get = tok.__getitem__
    for q, u in enumerate(users):
        fh.write(uid_names[u] + ''.join(map(get, idx[cuts[q]:cuts[q + 1]])) + '\n')
print('wrote %s (%.2f GB) in %.1f s' % (path, os.path.getsize(path) / 1e9, time.time() - t0), flush=True)
uids = {n: i for i, n in enumerate(uid_names)}
vids = {n: i for i, n in enumerate(vid_names)}
os.environ['TKR_NO_CACHE'] = '0'
t0 = time.time(); um, vm = textio.IdMap(uids), textio.IdMap(vids); t_maps = time.time() - t0
t0 = time.time(); R = textio.parse_ratings(path, um, vm); t_cold = time.time() - t0
t0 = time.time(); R2 = textio.parse_ratings(path, um, vm); t_warm = time.time() - t0
assert np.array_equal(R.item, R2.item) and len(R.item) == len(r['tr_u'])
print('id tables %.2f s; parse_ratings: %.2f s from text (%.0f M ratings/s, %.2f GB/s), %.2f s from the stamped copy (%s.csr.npz, %.2f GB)'
      % (t_maps, t_cold, len(R.item) / t_cold / 1e6, os.path.getsize(path) / t_cold / 1e9, t_warm, os.path.basename(path),
         os.path.getsize(path + '.csr.npz') / 1e9), flush=True)
import shutil; shutil.rmtree(d)
