import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_problem
from diffqcqp_amd import _capi, ops
N, B = 8, 65536
d = {k: v.cuda() for k, v in make_problem("qcqp", B, N, 1002).items()}
pd = torch.diagonal(d["P"], dim1=1, dim2=2).contiguous()
x = torch.empty(B, N, 1, dtype=torch.float64, device="cuda")
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); b.synchronize(); return a.elapsed_time(b)*1e3/n
for lpp in (1, 2):
    _capi.set_option("fwd_lpp", lpp)
    for layout, Pin in ((0, d["P"]), (2, pd)):
        row = []
        for mi in (1, 2, 5, 10, 15, 20, 25, 1000):
            row.append("%d:%.1f" % (mi, t(lambda: ops.qp_forward(Pin, d["q"], 1e-7, mi, layout=layout, out=x))))
        print("QP   lpp", lpp, "layout", layout, " ".join(row))
        row = []
        for mi in (1, 2, 5, 10, 15, 20, 25, 1000):
            row.append("%d:%.1f" % (mi, t(lambda: ops.qcqp_forward(Pin, d["q"], d["l_n"], d["mu"], 1e-7, mi, layout=layout, out=x))))
        print("QCQP lpp", lpp, "layout", layout, " ".join(row))
