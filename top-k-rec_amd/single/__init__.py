"""``from single import *`` -- the reference's model package surface (single/__init__.py:1-8).

REC, BPR and VBPR are the MI355X-native implementations of this repo.  The reference's other
five exports are ALS / MLP models outside the BPR/VBPR hot path;
they are present as names only and raise on construction.
"""
from .rec import REC
from .bpr import BPR
from .vbpr import VBPR


def _out_of_scope(name):
    class _Stub:
        def __init__(self, *a, **kw):
            raise NotImplementedError('%s (ALS/MLP family) is outside the BPR/VBPR hot path this '
                                      'build accelerates; use the reference implementation' % name)
    _Stub.__name__ = name
    return _Stub


WMF, DPM, CER, ENCODER, MLP = (_out_of_scope(n) for n in ('WMF', 'DPM', 'CER', 'ENCODER', 'MLP'))
__all__ = ['REC', 'BPR', 'VBPR', 'WMF', 'DPM', 'CER', 'ENCODER', 'MLP']
