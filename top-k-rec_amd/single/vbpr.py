"""VBPR -- content-aware BPR trained on MI355X (class API of the reference's single/vbpr.py).

    model = VBPR(k=50, d=20000)
    model.load_training_data('data/uid', 'data/vid', 'data/f0tr.txt')
    model.load_content_data('data/meta.pkl', 'data/vid')
    model.train(epochs=5, batch_size=256, epoch_sample_limit=10e5)
    model.export_embeddings('embed/vbpr')

k//2 rating dimensions + k//2 content dimensions projected from the item features by ``cem [d, k//2]``
plus a content bias ``icb [d]`` (vbpr.py:29-74).  The feature matrix is uploaded once and stays in
HBM; the per-batch host gather + feed of two [B, d] slices (vbpr.py:114) does not exist here.
After training the content half is folded into the exported factors exactly like vbpr.py:124-126,
so evaluation is identical to BPR's:  fue = [ure|uce], fie = [ire | feat.cem], fib = irb + feat.icb.

Mirrored quirk that changes the arithmetic: the data term of the objective is a sum over ALL PAIRS of a batch,
``sum_{a,b} log(1 + exp(-(alpha_a + beta_b)))`` with alpha = irb_i - irb_j + (f_i - f_j).icb and beta = x_ui - x_uj --
the reference's item_rating_bias is [n_items, 1], so its ``irbb - jrbb + x_ui - x_uj + matmul(ic - jc, icb)`` (vbpr.py:61)
broadcasts to [B, B] before the reduce_sum of :64.  The kernels compute exactly that (csrc/vbpr_step.hip, pair kernel),
the oracle is checked against autograd on the literal expressions with the reference's shapes (tests/test_oracle_step.py).
BPR itself is not affected (its bias is 1-D, bpr.py:79).

Other mirrored quirks: an odd k silently drops a dimension (exported width 2*(k//2)); resuming from text
re-imports ``fib`` (which already contains feat.icb) into irb (vbpr.py:106-108), double-counting the
content bias unless the checkpoint (``weights``) is also present -- as in the reference, the
checkpoint holds irb itself but the text value wins.
"""
from __future__ import annotations

import numpy as np
import torch

from utils import tprint
from .bpr import BPR
from . import _engine


class VBPR(BPR):
    def __init__(self, k: int, d: int, lambda_u: float = 2.5e-3, lambda_i: float = 2.5e-3,
                 lambda_j: float = 2.5e-4, lambda_b: float = 0, lambda_e: float = 0, lr: float = 1.0e-4,
                 mode: str = 'l2') -> None:
        super().__init__(k, lambda_u, lambda_i, lambda_j, lambda_b, lr, mode)
        self.d = d
        self.le = lambda_e

    def _hyper(self):
        hp = super()._hyper()
        hp['le'] = self.le
        return hp

    def _make_engine(self, device, seed, n_users=None, user_seed=None):
        assert getattr(self, 'feat', None) is not None, 'call load_content_data() before train()'
        return _engine.VbprEngine(self.n_users if n_users is None else n_users, self.n_items, self.k, self.d, self.feat, self._hyper(), device, seed,
                                  user_seed=user_seed)

    def _warm_start(self):
        """vbpr.py:99-108: fue splits into ure|uce, the first half of fie is ire, fib goes to irb."""
        kh = self.k // 2
        if self.fue is not None:
            tprint('Initialize user embeddings')
            self._eng.set_users(U=np.ascontiguousarray(self._user_rows(self.fue)[:, :2 * kh]))
        if self.fie is not None:
            tprint('Initialize item embeddings')
            self._eng.set_items(I=np.ascontiguousarray(np.asarray(self.fie)[:, :kh]))
        if self.fib is not None:
            tprint('Initialize item biases')
            self._eng.set_items(irb=np.asarray(self.fib).ravel())

    def _collect(self):
        """vbpr.py:124-126 (two [n_items, d] x [d, .] products, once per train(): torch matmul)"""
        e = self._eng
        self.fue = self._collect_users(2 * (self.k // 2))
        ire, irb = e.get('I')[0], e.get('irb')[0]
        self.fie = torch.cat([ire, e.feat @ e.cem], dim=1).cpu().numpy()
        self.fib = (irb + e.feat @ e.icb).reshape(-1, 1).cpu().numpy()

    def _checkpoint_tensors(self):
        e = self._eng
        out = {}
        for name in ('U', 'I', 'irb', 'cem', 'icb'):
            p, ms = e.get(name)
            out[name], out['ms_' + name] = p.cpu(), ms.cpu()
        if getattr(self, '_global_users', None) is not None:
            out['U'], out['ms_U'] = self._global_users
        return out

    def _restore_tensors(self, blob):
        e = self._eng
        e.set_users(U=self._user_rows(blob['U']), msU=self._user_rows(blob['ms_U']))
        if getattr(self, '_owned', None) is not None:
            self._global_users = (torch.as_tensor(np.asarray(blob['U'])).clone(), torch.as_tensor(np.asarray(blob['ms_U'])).clone())
        e.set_items(I=blob['I'], irb=blob['irb'], msI=blob['ms_I'], msirb=blob['ms_irb'])
        e.set_dense(cem=blob['cem'], icb=blob['icb'], mscem=blob['ms_cem'], msicb=blob['ms_icb'])
