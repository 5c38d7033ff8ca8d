"""VBPR (content-aware BPR) -- class API of the reference's single/vbpr.py.  Filled in by K3."""
from .bpr import BPR


class VBPR(BPR):
    def __init__(self, k: int, d: int, lambda_u: float = 2.5e-3, lambda_i: float = 2.5e-3,
                 lambda_j: float = 2.5e-4, lambda_b: float = 0, lambda_e: float = 0, lr: float = 1.0e-4,
                 mode: str = 'l2') -> None:
        super().__init__(k, lambda_u, lambda_i, lambda_j, lambda_b, lr, mode)
        self.d = d
        self.le = lambda_e
