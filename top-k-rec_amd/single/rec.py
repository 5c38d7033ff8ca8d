"""REC -- base class of the recommenders (API of the reference's single/rec.py:18-82).

Same abstract contract and the same text export/import of ``final-U.dat`` / ``final-V.dat`` /
``final-B.dat``; no TensorFlow: model state lives in PyTorch-ROCm tensors and the numerics
run in libtkr_hip.so.
"""
from __future__ import annotations

import os
import pickle
from abc import ABC, abstractmethod

import numpy as np

from utils import export_embed_to_file, get_embed_from_file, get_id_dict_from_file, tprint


class REC(ABC):
    @abstractmethod
    def load_training_data(self):
        pass

    def load_content_data(self, content_file: str, iid_file: str) -> None:
        """rec.py:23-33: pickled (latin1) dense or scipy-sparse [n_feat_items, d] matrix whose
        rows follow ``iid_file``; rows are placed at the model's item indices, items without
        features stay zero.  ``self.feat`` is the dense fp32 host copy the reference exposes."""
        import scipy.sparse as ss
        tprint('Load content data from %s' % (content_file))
        feature_ids = get_id_dict_from_file(iid_file)
        with open(content_file, 'rb') as fh:
            raw = pickle.load(fh, encoding='latin1')
        dst = [idx for iid, idx in self.iids.items() if iid in feature_ids]
        src = [feature_ids[iid] for iid in self.iids if iid in feature_ids]
        self.feat = np.zeros((self.n_items, self.d), dtype=np.float32)
        if ss.issparse(raw):
            self.feat[dst, :] = raw.tocsr()[src, :].toarray()
        else:
            self.feat[dst, :] = np.asarray(raw)[src, :]
        tprint('Loading finished!')

    @abstractmethod
    def build_graph(self):
        pass

    @abstractmethod
    def train(self):
        pass

    @abstractmethod
    def export_model(self, model_path: str) -> None:
        pass

    def export_embeddings(self, model_path: str) -> None:
        """rec.py:47-63: non-recursive mkdir, three '%f ' text matrices, then export_model."""
        if not os.path.exists(model_path):
            tprint('%s does not exist, create it instead' % model_path)
            os.mkdir(model_path)
        if not os.path.isdir(model_path):
            tprint('%s is not a folder' % model_path)
            return
        for attr, fname, what in (('fue', 'final-U.dat', 'user embeddings'),
                                  ('fie', 'final-V.dat', 'item embeddings'),
                                  ('fib', 'final-B.dat', 'item biases')):
            if hasattr(self, attr):
                target = os.path.join(model_path, fname)
                tprint('Saving %s to %s' % (what, target))
                export_embed_to_file(target, getattr(self, attr))
        self.export_model(model_path)

    @abstractmethod
    def import_model(self, model_path: str) -> None:
        pass

    def import_embeddings(self, model_path: str) -> None:
        """rec.py:69-82: read whichever of the three text matrices exist, then import_model."""
        for attr, fname, what, ids in (('fue', 'final-U.dat', 'user embeddings', self.uids),
                                       ('fie', 'final-V.dat', 'item embeddings', self.iids),
                                       ('fib', 'final-B.dat', 'item biases', self.iids)):
            source = os.path.join(model_path, fname)
            if os.path.exists(source):
                tprint('Loading %s from %s' % (what, source))
                setattr(self, attr, get_embed_from_file(source, ids))
        self.import_model(model_path)
