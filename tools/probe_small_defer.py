"""Sweep of lane_defer for the team-per-problem forward (fwd_small.hip, dense P, N = 10 .. 16), us per forward."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_problem
from diffqcqp_amd import ops, _capi
for N, B in ((16, 16384), (16, 65536), (12, 65536)):
    for kind in ("qp", "qcqp"):
        d = {k: v.cuda() for k, v in make_problem(kind, B, N, 7, structure="dense").items()}
        out = torch.empty(B, N, 1, dtype=torch.float64, device="cuda")
        def fwd():
            if kind == "qp":
                return ops.qp_forward(d["P"], d["q"], 1e-7, 1000, layout=1, out=out)
            return ops.qcqp_forward(d["P"], d["q"], d["l_n"], d["mu"], 1e-7, 1000, layout=1, out=out)
        res = []
        for defer in (1, 2, 3, 4, 6, 8, 12):
            _capi.set_option("lane_defer", defer)
            fwd(); torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(3): fwd()
                torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 3 * 1e6)
            res.append("%d: %.0f" % (defer, sorted(ts)[2]))
        print("N=%d B=%d %s  " % (N, B, kind) + "  ".join(res))
        del d
_capi.set_option("lane_defer", 0)
