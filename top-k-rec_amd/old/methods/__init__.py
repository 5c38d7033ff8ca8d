from .bpr import BPR  # noqa: F401
