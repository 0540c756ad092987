"""Does a DIAGONAL problem get the same backward bits from the general kernels (team / lane per problem) as from the diagonal fast
path?  (bwd_diag.hip queues whole tiles, so diagonal problems do land in the general kernel; a route chosen by a hint may send more.)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_problem
from diffqcqp_amd import ops, _capi
for kind in ("qp", "qcqp"):
    for N, B in ((8, 30000), (4, 20000), (2, 20000)):
        d = {k: v.cuda() for k, v in make_problem(kind, B, N, 800 + N, "diag").items()}
        if kind == "qp":
            x = ops.qp_forward(d["P"], d["q"], 1e-7, 1000)
            f = lambda layout: ops.qp_backward(d["P"], d["q"], x, d["grad_x"], layout=layout, return_steps=True)
        else:
            x = ops.qcqp_forward(d["P"], d["q"], d["l_n"], d["mu"], 1e-7, 1000)
            f = lambda layout: ops.qcqp_backward(d["P"], d["q"], d["l_n"], d["mu"], x, d["grad_x"], layout=layout, return_steps=True)
        fast = f(0)
        for name, lane in (("lane", 1), ("team", 0)):
            _capi.set_option("lane_bwd", lane)
            gen = f(1)
            diffs = [int((a != b).reshape(a.shape[0], -1).any(1).sum()) for a, b in zip(fast, gen)]
            worst = max(float((a.double() - b.double()).abs().max()) for a, b in zip(fast, gen))
            print(kind, N, B, name, "problems differing per output:", diffs, "max abs diff %.3g" % worst, flush=True)
        _capi.set_option("lane_bwd", 1)
