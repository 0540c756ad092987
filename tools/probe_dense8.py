"""Dense 8x8 P, B=65536, DQQ_P_DENSE: iteration statistics and time of the forward (lane-per-problem kernel) and
backward (team kernel).  rocprofv3 target too.   python tools/probe_dense8.py [kind] [B]"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_problem
from diffqcqp_amd import ops
kind = sys.argv[1] if len(sys.argv) > 1 else "qcqp"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
d = {k: v.cuda() for k, v in make_problem(kind, B, 8, 7, structure="dense").items()}
def fwd(lay=1):
    if kind == "qp":
        return ops.qp_forward(d["P"], d["q"], 1e-7, 1000, layout=lay, return_iters=True)
    return ops.qcqp_forward(d["P"], d["q"], d["l_n"], d["mu"], 1e-7, 1000, layout=lay, return_iters=True)
x, it = fwd()
it = it.double()
print(kind, "iterations mean %.1f p50 %.0f p99 %.0f max %.0f; mean of max over 64: %.1f, over 16: %.1f"
      % (it.mean(), it.median(), torch.quantile(it, 0.99), it.max(), it.view(-1, 64).max(1).values.mean(),
         it.view(-1, 16).max(1).values.mean()))
def bwd(lay=1):
    if kind == "qp":
        return ops.qp_backward(d["P"], d["q"], x, d["grad_x"], layout=lay)
    return ops.qcqp_backward(d["P"], d["q"], d["l_n"], d["mu"], x, d["grad_x"], layout=lay)
for name, fn in (("forward", fwd), ("backward", bwd)):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): fn()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 5 * 1e6)
    print(name, "DENSE us", sorted(ts)[2])
