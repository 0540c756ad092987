"""lane_defer for the box kinds on the lane-per-problem forward (dense 8 x 8, B = 65536), us per forward."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_problem
from diffqcqp_amd import ops, _capi
for kind in ("box", "sbox"):
    B, N = 65536, 8
    d = {k: v.cuda() for k, v in make_problem(kind, B, N, 7, structure="dense").items()}
    def fwd():
        return ops.boxqp_forward(d["P"], d["q"], d["l_min"], d["l_max"], 1e-7, 1000, v=d.get("v"), layout=1)
    res = []
    for defer in (1, 2, 4, 6, 8):
        _capi.set_option("lane_defer", defer)
        fwd(); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5): fwd()
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 5 * 1e6)
        res.append("%d: %.1f" % (defer, sorted(ts)[2]))
    print(kind, "  ".join(res))
_capi.set_option("lane_defer", 0)
