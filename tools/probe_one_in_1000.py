"""B = 65536, N = 8, diagonal batch with one dense problem in 1000 (and a fully dense one) through DQQ_P_AUTO: us per
forward / backward (the backward with the forward's flags)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_problem
from diffqcqp_amd import ops
B, N = 65536, 8
def t(fn, n=10):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / n * 1e6)
    return sorted(ts)[2]
for kind in ("qp", "qcqp"):
    dd, dg = make_problem(kind, B, N, 6100, "dense"), make_problem(kind, B, N, 6100, "diag")
    for name, sel in (("diagonal", torch.zeros(B, dtype=torch.bool)), ("1 in 1000 dense", torch.arange(B) % 1000 == 1),
                      ("dense", torch.ones(B, dtype=torch.bool))):
        d = dict(dd); d["P"] = torch.where(sel.view(B, 1, 1), dd["P"], dg["P"]).contiguous()
        g = {k: v.cuda() for k, v in d.items()}
        c = ops.diag_cache(g["q"])
        x = torch.empty(B, N, 1, dtype=torch.float64, device="cuda")
        if kind == "qp":
            f = lambda: ops.qp_forward(g["P"], g["q"], 1e-7, 1000, out=x, cache=c)
            b = lambda: ops.qp_backward(g["P"], g["q"], x, g["grad_x"], cache=c)
        else:
            f = lambda: ops.qcqp_forward(g["P"], g["q"], g["l_n"], g["mu"], 1e-7, 1000, out=x, cache=c)
            b = lambda: ops.qcqp_backward(g["P"], g["q"], g["l_n"], g["mu"], x, g["grad_x"], cache=c)
        print("%-5s %-16s forward %.1f us  backward %.1f us" % (kind, name, t(f), t(b)))
