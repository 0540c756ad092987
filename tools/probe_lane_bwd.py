"""Lane-per-problem backward (bwd_lane_dense.hip, option lane_bwd) against the team kernel on dense P declared dense:
bit identity of every output and timing.   python tools/probe_lane_bwd.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_problem
from diffqcqp_amd import _capi, ops

def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); b.synchronize(); return a.elapsed_time(b) * 1e3 / n

for kind in ("qcqp", "qp"):
    sizes = ((8, 65536), (8, 65536 + 37), (8, 262144), (6, 65536), (4, 65536), (2, 65536), (8, 16384))
    if "--sweep" in sys.argv:   # where does a lane per problem start to pay?  (bwd_lane_dense_supported's thresholds)
        sizes = tuple((N, B) for N in (8, 6, 4, 2) for B in (16384, 24576, 32768, 49152, 65536, 131072))
    for N, B in sizes:
        d = {k: v.cuda() for k, v in make_problem(kind, B, N, 4242 + N, "dense").items()}
        if kind == "qp":
            x = ops.qp_forward(d["P"], d["q"], 1e-7, 1000, layout=1)
            run = lambda: ops.qp_backward(d["P"], d["q"], x, d["grad_x"], layout=1, return_steps=True)
        else:
            x = ops.qcqp_forward(d["P"], d["q"], d["l_n"], d["mu"], 1e-7, 1000, layout=1)
            run = lambda: ops.qcqp_backward(d["P"], d["q"], d["l_n"], d["mu"], x, d["grad_x"], layout=1, return_steps=True)
        res, tm = {}, {}
        for opt in (0, 1):
            _capi.set_option("lane_bwd", opt)
            res[opt] = [o.clone() for o in run()]
            tm[opt] = t(run)
        _capi.set_option("lane_bwd", 1)
        same = all(torch.equal(a, b) for a, b in zip(res[0], res[1]))
        st = res[1][-1].float()
        print("%-5s N=%d B=%6d  team %7.1f us  lane %7.1f us  %s  steps mean %.2f max %d" %
              (kind, N, B, tm[0], tm[1], "bit-identical" if same else "DIFFERS %s" % [float((a.double() - b.double()).abs().max()) for a, b in zip(res[0], res[1])],
               st.mean().item(), int(st.max().item())), flush=True)
