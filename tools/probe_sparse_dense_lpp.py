"""Forward through DQQ_P_AUTO, N = 8, a batch with a FEW non-diagonal problems (one in `every`): two lanes per problem (the built-in
choice from 57344 problems on; a non-diagonal tile takes two passes of the four-lane general solve) against four (one pass)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_problem
from diffqcqp_amd import ops, _capi
def t(fn, n=30, reps=3):
    out = []
    for _ in range(reps):
        for _ in range(5): fn()
        torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): fn()
        b.record(); b.synchronize(); out.append(a.elapsed_time(b) * 1e3 / n)
    return sorted(out)[1]
for kind in ("qp", "qcqp"):
    for B in (65536, 131072, 262144):
        dd = make_problem(kind, B, 8, 4251, "dense")
        for every in (0, 100000, 1000, 100, 10, 4, 2, 1):
            d = make_problem(kind, B, 8, 4250, "diag")
            if every:
                idx = torch.arange(every // 2, B, every)
                d["P"][idx] = dd["P"][idx]
            g = {k: v.cuda() for k, v in d.items()}
            xo = torch.empty(B, 8, 1, dtype=torch.float64, device="cuda")
            run = (lambda: ops.qp_forward(g["P"], g["q"], 1e-7, 1000, out=xo)) if kind == "qp" else (lambda: ops.qcqp_forward(g["P"], g["q"], g["l_n"], g["mu"], 1e-7, 1000, out=xo))
            res = []
            for lpp in (2, 4, 1):
                _capi.set_option("fwd_lpp", lpp); res.append(t(run))
            _capi.set_option("fwd_lpp", 0)
            print("%-5s B=%6d one non-diagonal problem in %6d: two lanes %.1f us, four lanes %.1f us, one lane %.1f us" % (kind, B, every, res[0], res[1], res[2]), flush=True)
