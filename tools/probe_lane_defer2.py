"""lane_defer sweep of the forward lane-per-problem kernel (dense 8 x 8 declared dense, B = 65536) -- rerun when the cost of a
refactorisation pass changes (DQQ_LANE_FWD_FACTORED)."""
import os, sys, torch
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
print("lib", os.environ.get("DQQ_LIB", "shipped").split("/")[-1])
for kind in ("qp", "qcqp", "box"):
    for B in (65536, 262144):
        d = {k: v.cuda() for k, v in make_problem(kind, B, 8, 4250, "dense").items()}
        xo = torch.empty(B, 8, 1, dtype=torch.float64, device="cuda")
        if kind == "qp": run = lambda: ops.qp_forward(d["P"], d["q"], 1e-7, 1000, layout=1, out=xo)
        elif kind == "qcqp": run = lambda: ops.qcqp_forward(d["P"], d["q"], d["l_n"], d["mu"], 1e-7, 1000, layout=1, out=xo)
        else: run = lambda: ops.boxqp_forward(d["P"], d["q"], d["l_min"], d["l_max"], 1e-7, 1000, layout=1, out=xo)
        row = []
        for df in (1, 2, 3, 4, 6, 8):
            _capi.set_option("lane_defer", df)
            row.append("%d: %.1f" % (df, t(run)))
        _capi.set_option("lane_defer", 0)
        print("%-5s B=%6d  us at lane_defer = %s   (built-in: %.1f)" % (kind, B, "  ".join(row), t(run)), flush=True)
