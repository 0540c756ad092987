"""Sweep of the lane-per-problem kernel's option lane_defer on the dense 8 x 8 family (B = 65536, DQQ_P_DENSE), us per forward."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_problem
from diffqcqp_amd import ops, _capi
for N, B in ((8, 65536), (8, 262144), (4, 65536)):
    for kind in ("qp", "qcqp"):
        d = {k: v.cuda() for k, v in make_problem(kind, B, N, 7, structure="dense").items()}
        out = torch.empty(B, N, 1, dtype=torch.float64, device="cuda")
        def fwd():
            if kind == "qp":
                return ops.qp_forward(d["P"], d["q"], 1e-7, 1000, layout=1, out=out)
            return ops.qcqp_forward(d["P"], d["q"], d["l_n"], d["mu"], 1e-7, 1000, layout=1, out=out)
        res = []
        for defer in (1, 2, 3, 4, 5, 6, 8, 12):
            _capi.set_option("lane_defer", defer)
            fwd(); torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(5): fwd()
                torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 5 * 1e6)
            res.append("%d: %.1f" % (defer, sorted(ts)[2]))
        print("N=%d B=%d %s  " % (N, B, kind) + "  ".join(res))
        del d
_capi.set_option("lane_defer", 4)
