import os, sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from conftest import make_problem
from diffqcqp_amd import ops, _capi
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); b.synchronize(); return a.elapsed_time(b) * 1e3 / n
for kind in ("qp", "qcqp"):
  for B in (32768, 65536, 131072):
    d = {k: v.cuda() for k, v in make_problem(kind, B, 8, 4250, "dense").items()}
    xo = torch.empty(B, 8, 1, dtype=torch.float64, device="cuda")
    run = (lambda **kw: ops.qp_forward(d["P"], d["q"], 1e-7, 1000, out=xo, **kw)) if kind == "qp" else (lambda **kw: ops.qcqp_forward(d["P"], d["q"], d["l_n"], d["mu"], 1e-7, 1000, out=xo, **kw))
    res = {}
    outs = {}
    for lpp in (0, 1, 2, 4):
        for wpb in (0, 1):
            _capi.set_option("fwd_lpp", lpp); _capi.set_option("wpb", wpb)
            res[(lpp, wpb)] = t(lambda: run(layout=0))
            outs[(lpp, wpb)] = run(layout=0).clone()
    _capi.set_option("fwd_lpp", 0); _capi.set_option("wpb", 0)
    dd = t(lambda: run(layout=1))
    same = all(torch.equal(outs[(0, 0)], v) for v in outs.values())
    print(kind, B, "AUTO us by (lpp, wpb):", {k: round(v, 1) for k, v in res.items()}, "declared dense %.1f" % dd, "all AUTO variants same bits:", same, flush=True)
