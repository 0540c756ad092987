"""Backward through DQQ_P_AUTO with the forward's hand-off (pdiag, flags), N = 8, B = 65536: batches with one non-diagonal problem
in `every` -- the fast path queues only the flag-2 problems of a mixed tile (bwd_diag.hip, by_problem)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_problem
from diffqcqp_amd import ops, _capi
def t(fn, n=40, reps=3):
    out = []
    for _ in range(reps):
        for _ in range(5): fn()
        torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): fn()
        b.record(); b.synchronize(); out.append(a.elapsed_time(b) * 1e3 / n)
    return sorted(out)[1]
print("lib", os.environ.get("DQQ_LIB", "shipped").split("/")[-1])
B, N = 65536, 8
for kind in ("qp", "qcqp"):
    dd = make_problem(kind, B, N, 4251, "dense")
    for every in (0, 100000, 1000, 100, 10):
        d = make_problem(kind, B, N, 4250, "diag")
        if every: d["P"][every // 2::every] = dd["P"][every // 2::every]
        g = {k: v.cuda() for k, v in d.items()}
        cache = ops.diag_cache(g["q"])
        if kind == "qp":
            x = ops.qp_forward(g["P"], g["q"], 1e-7, 1000, cache=cache)
            out = [torch.empty_like(g["P"]), torch.empty_like(g["q"])]
            run = lambda c: ops.qp_backward(g["P"], g["q"], x, g["grad_x"], out=out, cache=c)
        else:
            x = ops.qcqp_forward(g["P"], g["q"], g["l_n"], g["mu"], 1e-7, 1000, cache=cache)
            out = [torch.empty_like(g["P"]), torch.empty_like(g["q"]), torch.empty_like(g["l_n"]), torch.empty_like(g["mu"])]
            run = lambda c: ops.qcqp_backward(g["P"], g["q"], g["l_n"], g["mu"], x, g["grad_x"], out=out, cache=c)
        print("%-5s one non-diagonal problem in %6d: backward with the hand-off %.1f us, without %.1f us" % (kind, every, t(lambda: run(cache)), t(lambda: run(None))), flush=True)
