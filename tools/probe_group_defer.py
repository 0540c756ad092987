"""Sweep of option lane_defer for the GROUP solve (group_dense.h): dense 8 x 8 through DQQ_P_AUTO at B = 65536 (the fused
forward's non-diagonal tiles) and through DQQ_P_DENSE at B = 4096 / 32768 (the group solve's own mapping), us per forward."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_problem
from diffqcqp_amd import ops, _capi
for B, lay in ((65536, 0), (4096, 1), (32768, 1)):
    for kind in ("qp", "qcqp"):
        d = {k: v.cuda() for k, v in make_problem(kind, B, 8, 7, structure="dense").items()}
        out = torch.empty(B, 8, 1, dtype=torch.float64, device="cuda")
        def fwd():
            if kind == "qp":
                return ops.qp_forward(d["P"], d["q"], 1e-7, 1000, layout=lay, out=out)
            return ops.qcqp_forward(d["P"], d["q"], d["l_n"], d["mu"], 1e-7, 1000, layout=lay, out=out)
        res = []
        for defer in (1, 2, 3, 4, 5, 6, 8):
            _capi.set_option("lane_defer", defer)
            fwd(); torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(5): fwd()
                torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 5 * 1e6)
            res.append("%d: %.1f" % (defer, sorted(ts)[2]))
        print("B=%d layout=%d %s  " % (B, lay, kind) + "  ".join(res))
_capi.set_option("lane_defer", 0)
