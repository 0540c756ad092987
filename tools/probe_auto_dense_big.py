"""Dense P at N = 32 / 64 through DQQ_P_AUTO (verifying pass + work-list + general kernel) against DQQ_P_DENSE
(general kernel alone): forward and backward, us per call.   python tools/probe_auto_dense_big.py [kind]"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_problem
from diffqcqp_amd import ops

kind = sys.argv[1] if len(sys.argv) > 1 else "qp"


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e6)
    return sorted(ts)[len(ts) // 2]


for N, B in ((32, 65536), (64, 16384), (64, 65536)):
    d = {k: v.cuda() for k, v in make_problem(kind, B, N, 7, structure="dense").items()}
    g = d["grad_x"]
    if kind == "qp":
        fwd = lambda lay, cache=None: ops.qp_forward(d["P"], d["q"], 1e-7, 1000, layout=lay, cache=cache)
        x = fwd(1)
        bwd = lambda lay, cache=None: ops.qp_backward(d["P"], d["q"], x, g, layout=lay, cache=cache)
    else:
        fwd = lambda lay, cache=None: ops.qcqp_forward(d["P"], d["q"], d["l_n"], d["mu"], 1e-7, 1000, layout=lay, cache=cache)
        x = fwd(1)
        bwd = lambda lay, cache=None: ops.qcqp_backward(d["P"], d["q"], d["l_n"], d["mu"], x, g, layout=lay, cache=cache)
    c = ops.diag_cache(d["q"])
    fwd(0, c)
    print("%s N=%d B=%d  forward AUTO %.0f DENSE %.0f us   backward AUTO (forward's flags) %.0f AUTO (no flags) %.0f DENSE %.0f us"
          % (kind, N, B, timeit(lambda: fwd(0)), timeit(lambda: fwd(1)), timeit(lambda: bwd(0, c)), timeit(lambda: bwd(0)),
             timeit(lambda: bwd(1))))
    del d, x, c
    torch.cuda.empty_cache()
