"""Top-level `qcqp` module: the file name users of the reference import from (`from qcqp import QPFn2, QCQPFn2`,
reference README.md:31, test_script.py:11).  Re-exports the MI355X drop-in classes of diffqcqp_amd/qcqp.py."""
from diffqcqp_amd.qcqp import BoxQPFn2, QCQPFn2, QPFn2, SignedBoxQPFn2  # noqa: F401

__all__ = ["QPFn2", "QCQPFn2", "BoxQPFn2", "SignedBoxQPFn2"]
