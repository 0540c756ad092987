"""Developer probe: the global-memory kernels (general_any.hip) beyond the register / LDS sizes."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_problem
from diffqcqp_amd import _capi, ops
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        a.record(); fn(); b.record(); b.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort(); return ts[len(ts) // 2]
for kind, N, B in (("qp", 96, 1024), ("qp", 128, 1024), ("qcqp", 64, 4096), ("qcqp", 128, 512), ("qp", 256, 512)):
    d = {k: v.cuda() for k, v in make_problem(kind, B, N, 9000 + N, "dense").items()}
    if kind == "qp":
        x, it = ops.qp_forward(d["P"], d["q"], 1e-7, 1000, layout=1, return_iters=True)
        tf = timeit(lambda: ops.qp_forward(d["P"], d["q"], 1e-7, 1000, layout=1))
        tb = timeit(lambda: ops.qp_backward(d["P"], d["q"], x, d["grad_x"], layout=1))
    else:
        x, it = ops.qcqp_forward(d["P"], d["q"], d["l_n"], d["mu"], 1e-7, 1000, layout=1, return_iters=True)
        tf = timeit(lambda: ops.qcqp_forward(d["P"], d["q"], d["l_n"], d["mu"], 1e-7, 1000, layout=1))
        tb = timeit(lambda: ops.qcqp_backward(d["P"], d["q"], d["l_n"], d["mu"], x, d["grad_x"], layout=1))
    print(json.dumps({"kind": kind, "N": N, "B": B, "fwd_ms": round(tf, 2), "bwd_ms": round(tb, 2), "iters_mean": round(float(it.float().mean()), 1)}), flush=True)
