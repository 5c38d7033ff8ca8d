"""End-to-end at MovieLens-10M shape through the reference's text formats and CLI, and the generator acceptance
check of SURVEY.md §8d: a BPR model trained on the synthetic ratings must beat the popularity-only ranking."""
import os, sys, time, tempfile, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'top-k-rec_amd')]
import numpy as np
import synth, evaluate as E
from single import BPR
from utils import export_embed_to_file

work = tempfile.mkdtemp(prefix='tkr_ml10m_')
data = os.path.join(work, 'data')
gen = dict(synth.ML10M); gen.update(alpha=float(os.environ.get('ALPHA', 0.9)), gain=float(os.environ.get('GAIN', 1.5)), select=float(os.environ.get('SELECT', 0.0)))
t0 = time.time(); r = synth.make_ratings(seed=42, **gen); t_gen = time.time() - t0
t0 = time.time(); synth.write_dataset(data, r); t_write = time.time() - t0
print('generate %.1fs, write text %.1fs (%d ratings)' % (t_gen, t_write, len(r['tr_u']) + len(r['im_u']) + len(r['om_u'])), flush=True)
out = {}
m = BPR(k=int(os.environ.get('K', 64)), lr=float(os.environ.get('LR', 1e-2)))
t0 = time.time(); m.load_training_data(data + '/uid', data + '/vid', data + '/f0tr.txt'); out['load_s'] = time.time() - t0
t0 = time.time(); m.train(epochs=int(os.environ.get('EPOCHS', 20)), batch_size=256, epoch_sample_limit=10e5, seed=1, verbose=False); out['train_s'] = time.time() - t0
t0 = time.time(); m.export_embeddings(work + '/bpr'); out['export_s'] = time.time() - t0
t0 = time.time(); out['bpr'] = E.main(['-d', data, '-m', work + '/bpr', '-sl', 'im', 'om']); out['eval_s'] = time.time() - t0
# popularity-only ranking: zero factors, bias = number of train likes
pop = np.bincount(r['tr_i'][r['tr_l'] == 1], minlength=m.n_items).astype(np.float32)
vid_pos = {v: i for i, v in enumerate(open(data + '/vid').read().split())}
os.mkdir(work + '/pop')
export_embed_to_file(work + '/pop/final-U.dat', np.zeros((m.n_users, 2), np.float32))
export_embed_to_file(work + '/pop/final-V.dat', np.zeros((m.n_items, 2), np.float32))
export_embed_to_file(work + '/pop/final-B.dat', pop.reshape(-1, 1))
out['popularity'] = E.main(['-d', data, '-m', work + '/pop', '-sl', 'im'])
# untrained model
m0 = BPR(k=8); m0.load_training_data(data + '/uid', data + '/vid', data + '/f0tr.txt')
rng = np.random.Generator(np.random.PCG64(0))
m0.fue = (rng.standard_normal((m0.n_users, 8)) * 0.01).astype(np.float32); m0.fie = (rng.standard_normal((m0.n_items, 8)) * 0.01).astype(np.float32)
m0.fib = np.zeros((m0.n_items, 1), np.float32)
m0.export_embeddings(work + '/rand')
out['random'] = E.main(['-d', data, '-m', work + '/rand', '-sl', 'im'])
print(json.dumps(out, indent=1))
