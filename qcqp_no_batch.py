"""Top-level `qcqp_no_batch` module (reference qcqp_no_batch.py: the single-problem twins of the autograd
Functions).  Re-exports diffqcqp_amd/qcqp_no_batch.py."""
from diffqcqp_amd.qcqp_no_batch import QCQPFn2, QPFn2  # noqa: F401

__all__ = ["QPFn2", "QCQPFn2"]
