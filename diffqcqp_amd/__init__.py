"""diffqcqp_amd -- MI355X-native batched differentiable QP / QCQP solver.

`diffqcqp_amd.qcqp` is the drop-in for the reference's `qcqp.py` (QPFn2,
QCQPFn2); `diffqcqp_amd.ops` exposes the same launches without autograd;
`diffqcqp_amd.parallel` shards a batch over the GPUs of a node.  The compute is
in libdiffqcqp_hip.so (csrc/, C ABI in include/diffqcqp_hip.h).
"""
__version__ = "0.2.0"
