"""Training driver -- the BPR/VBPR part of the reference's train.py (train.py:1-16), same
hard-coded data/ and embed/ paths and hyper-parameters.  The reference's lines 18-36 (WMF, CER,
DPM: ALS models) are outside the path this build accelerates.

    cd top-k-rec_amd && python train.py            # expects data/uid, data/vid, data/f0tr.txt (+ data/meta.pkl)

``epoch_sample_limit=10e5`` is kept as the reference writes it (a float): this build accepts
integral floats where single/bpr.py:111 would trip its own isinstance assert.
"""
import os

from single import *

if __name__ == '__main__':
    model = BPR(k=50)
    model.load_training_data('data/uid', 'data/vid', 'data/f0tr.txt')
    # Training from scratch
    model.train(epochs=5, batch_size=256, epoch_sample_limit=10e5)
    model.export_embeddings('embed/bpr')
    # Training from a pretrained model
    model.train(epochs=5, batch_size=256, epoch_sample_limit=10e5, model_path='embed/bpr')

    if os.path.exists('data/meta.pkl'):
        model = VBPR(k=50, d=20000)
        model.load_training_data('data/uid', 'data/vid', 'data/f0tr.txt')
        model.load_content_data('data/meta.pkl', 'data/vid')
        model.train(epochs=5, batch_size=256, epoch_sample_limit=10e5)
        model.export_embeddings('embed/vbpr')
        model.train(epochs=5, batch_size=256, epoch_sample_limit=10e5, model_path='embed/vbpr')
