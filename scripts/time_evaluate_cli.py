"""dev probe: wall time of the evaluate.py CLI on an ML-10M-shaped dataset written in the reference's text formats
(the parsers are the wall once K4 takes milliseconds)."""
import os, sys, time, tempfile, io, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'top-k-rec_amd')]
import numpy as np
import synth, utils, evaluate as E, textio

d = tempfile.mkdtemp()
data, model = os.path.join(d, 'data'), os.path.join(d, 'model')
t = time.time(); r = synth.make_ratings(seed=42, **synth.ML10M); synth.write_dataset(data, r)
print('dataset written in %.1f s (%.0f MB train file)' % (time.time() - t, os.path.getsize(os.path.join(data, 'f0tr.txt')) / 1e6))
rng = np.random.Generator(np.random.PCG64(1))
n_items = r['n_in'] + r['n_out']
U = (rng.standard_normal((r['n_users'], 128)) * 0.1).astype(np.float32)
V = (rng.standard_normal((n_items, 128)) * 0.1).astype(np.float32)
os.makedirs(model)
t = time.time(); utils.export_embed_to_file(os.path.join(model, 'final-U.dat'), U); utils.export_embed_to_file(os.path.join(model, 'final-V.dat'), V)
print('export_embed_to_file U+V: %.2f s' % (time.time() - t))
for label, env in (('text only (TKR_NO_CACHE=1)', '1'), ('with .npy copies', '0'), ('with .npy copies, 2nd run', '0')):
    os.environ['TKR_NO_CACHE'] = env
    t = time.time()
    with contextlib.redirect_stdout(io.StringIO()) as out:
        E.main(['-d', data, '-m', model, '-sl', 'im', 'om'])
    print('evaluate.py -sl im om, %s: %.2f s   %s' % (label, time.time() - t, out.getvalue().strip().replace('\n', ' | ')[:80]))
t = time.time(); uids = E.read_ids(os.path.join(data, 'uid')); teids = E.read_ids(os.path.join(data, 'f0te.im.idl'))
R = textio.parse_ratings(os.path.join(data, 'f0tr.txt'), uids, teids)
print('parse_ratings(f0tr.txt): %.2f s for %d entries' % (time.time() - t, len(R.item)))
from single import BPR
m = BPR(k=128)
t = time.time(); m.load_training_data(os.path.join(data, 'uid'), os.path.join(data, 'vid'), os.path.join(data, 'f0tr.txt'))
print('BPR.load_training_data: %.2f s, %d positives' % (time.time() - t, m.epoch_sample_limit))
