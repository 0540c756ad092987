import sys, torch
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_problem
from diffqcqp_amd import ops, _capi
for kind in ("qp", "qcqp"):
    d = {k: v.cuda() for k, v in make_problem(kind, 70000, 8, 796, "diag").items()}
    res = {}
    for lpp in (1, 2, 4):
        for rs in (0, 1):
            _capi.set_option("fwd_lpp", lpp); _capi.set_option("fwd_respread", 0 if rs == 0 else 12)
            if kind == "qp": res[(lpp, rs)] = ops.qp_forward(d["P"], d["q"], 1e-7, 1000, return_iters=True)
            else: res[(lpp, rs)] = ops.qcqp_forward(d["P"], d["q"], d["l_n"], d["mu"], 1e-7, 1000, return_iters=True)
    base = res[(2, 1)]
    for k, (x, it) in res.items():
        nd = (x != base[0]).flatten(1).any(1)
        print(kind, k, "problems differing from (2, respread):", int(nd.sum()), "max |dx| %.3g" % float((x - base[0]).abs().max()), "iters differ:", int((it != base[1]).sum()))
_capi.set_option("fwd_lpp", 0); _capi.set_option("fwd_respread", -1)
