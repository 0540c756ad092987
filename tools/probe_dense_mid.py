"""Developer probe: dense-P forward / backward at the sizes between the team kernels (N <= 16) and N = 64."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_problem
from diffqcqp_amd import _capi, ops
def timeit(fn, reps=7):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        a.record(); fn(); b.record(); b.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort(); return ts[len(ts) // 2]
for kind, N, B in (("qp", 20, 16384), ("qp", 24, 16384), ("qp", 32, 8192), ("qcqp", 32, 8192), ("qp", 40, 8192), ("qp", 48, 8192), ("qp", 56, 8192), ("box", 32, 8192)):
    d = {k: v.cuda() for k, v in make_problem(kind, B, N, 9100 + N, "dense").items()}
    row = {"kind": kind, "N": N, "B": B}
    for opt in (1, 0):
        _capi.set_option("dense_wave64", opt)
        if kind == "qp":
            f = lambda: ops.qp_forward(d["P"], d["q"], 1e-7, 1000, layout=1)
        elif kind == "qcqp":
            f = lambda: ops.qcqp_forward(d["P"], d["q"], d["l_n"], d["mu"], 1e-7, 1000, layout=1)
        else:
            f = lambda: ops.boxqp_forward(d["P"], d["q"], d["l_min"], d["l_max"], 1e-7, 1000, layout=1)
        row["fwd_ms_wave" if opt else "fwd_ms_old"] = round(timeit(f), 3)
    _capi.set_option("dense_wave64", 1)
    if kind == "qp":
        x = ops.qp_forward(d["P"], d["q"], 1e-7, 1000, layout=1)
        row["bwd_ms"] = round(timeit(lambda: ops.qp_backward(d["P"], d["q"], x, d["grad_x"], layout=1)), 3)
    elif kind == "qcqp":
        x = ops.qcqp_forward(d["P"], d["q"], d["l_n"], d["mu"], 1e-7, 1000, layout=1)
        row["bwd_ms"] = round(timeit(lambda: ops.qcqp_backward(d["P"], d["q"], d["l_n"], d["mu"], x, d["grad_x"], layout=1)), 3)
    print(json.dumps(row), flush=True)
