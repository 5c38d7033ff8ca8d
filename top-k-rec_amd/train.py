"""Training driver -- the BPR/VBPR part of the reference's train.py (train.py:1-16): the same
hard-coded data/ and embed/ locations, k = 50, five epochs of batch 256 from scratch, export, then
five more epochs warm-started from the exported model.  The reference's lines 18-36 (WMF, CER,
DPM: ALS models) are outside the path this build accelerates.

    cd top-k-rec_amd && python train.py            # expects data/uid, data/vid, data/f0tr.txt (+ data/meta.pkl)

``epoch_sample_limit=10e5`` is kept as the reference writes it (a float): this build accepts
integral floats where single/bpr.py:111 would trip its own isinstance assert.
"""
import os

from single import *

DATA = dict(uid='data/uid', vid='data/vid', ratings='data/f0tr.txt', content='data/meta.pkl')
SCHEDULE = dict(epochs=5, batch_size=256, epoch_sample_limit=10e5)


def fit(model, out_dir, with_content=False):
    model.load_training_data(DATA['uid'], DATA['vid'], DATA['ratings'])
    if with_content:
        model.load_content_data(DATA['content'], DATA['vid'])
    model.train(**SCHEDULE)                                   # from scratch
    model.export_embeddings(out_dir)
    model.train(model_path=out_dir, **SCHEDULE)               # warm start from the exported embeddings
    return model


if __name__ == '__main__':
    fit(BPR(k=50), 'embed/bpr')
    if os.path.exists(DATA['content']):
        fit(VBPR(k=50, d=20000), 'embed/vbpr', with_content=True)
