"""Split of the lane-per-problem forward (dense 8 x 8 declared dense, B = 65536) into its fixed part, its cost per trip and the
cost of the rho updates: times at max_iter = 1 ... 1000 and with adaptive_rho off."""
import os, sys, torch
LAYOUT = int(sys.argv[1]) if len(sys.argv) > 1 else 1   # 1: declared dense (lane kernel), 0: AUTO (fused group solve)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_problem
from diffqcqp_amd import ops, _capi
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); b.synchronize(); return a.elapsed_time(b) * 1e3 / n
for kind in ("qp", "qcqp"):
    d = {k: v.cuda() for k, v in make_problem(kind, 65536, 8, 4250, "dense").items()}
    xo = torch.empty(65536, 8, 1, dtype=torch.float64, device="cuda")
    run = (lambda mi, **kw: ops.qp_forward(d["P"], d["q"], 1e-7, mi, layout=LAYOUT, **kw)) if kind == "qp" else (lambda mi, **kw: ops.qcqp_forward(d["P"], d["q"], d["l_n"], d["mu"], 1e-7, mi, layout=LAYOUT, **kw))
    _, it = run(1000, return_iters=True)
    itw = it.view(-1, 64).max(1).values.float()
    print(kind, "iterations mean %.1f max %d, wave-max mean %.1f" % (it.float().mean(), it.max(), itw.mean()))
    for mi in (1, 2, 5, 10, 20, 40, 1000):
        print("   max_iter %4d: %.1f us" % (mi, t(lambda: run(mi, out=xo))))
    for adaptive in (False,):
        print("   adaptive_rho=False, 20 iterations: %.1f us" % t(lambda: ops.qcqp_forward(d["P"], d["q"], d["l_n"], d["mu"], 1e-7, 20, layout=LAYOUT, adaptive_rho=False, out=xo) if kind == "qcqp" else ops.qp_forward(d["P"], d["q"], 1e-7, 20, layout=LAYOUT, adaptive_rho=False, out=xo)))
