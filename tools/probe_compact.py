"""Developer probe: N = 8 forward with / without workgroup compaction at several iteration budgets."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_problem
from diffqcqp_amd import _capi, ops
B, N = 65536, 8
d = {k: v.cuda() for k, v in make_problem("qcqp", B, N, 1002).items()}
x = torch.empty(B, N, 1, dtype=torch.float64, device="cuda")
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); b.synchronize(); return a.elapsed_time(b) * 1e3 / n
for kind in ("qp", "qcqp"):
    for c in (0, 1):
        _capi.set_option("fwd_compact", c)
        row = []
        for mi in (1, 5, 10, 14, 15, 16, 18, 21, 24, 30, 1000):
            if kind == "qp": f = lambda: ops.qp_forward(d["P"], d["q"], 1e-7, mi, out=x)
            else: f = lambda: ops.qcqp_forward(d["P"], d["q"], d["l_n"], d["mu"], 1e-7, mi, out=x)
            row.append("%d:%.1f" % (mi, t(f)))
        print(kind, "compact", c, " ".join(row))
# heavy-tailed iteration counts: the reference's figure workload distribution, P = diag(exp(U(-10, 10)))
g = torch.Generator().manual_seed(3)
p = torch.exp(20 * torch.rand(B, N, generator=g, dtype=torch.float64) - 10)
dh = dict(d); dh["P"] = torch.diag_embed(p).cuda()
for eps in (1e-7, 1e-10):
    for kind in ("qp", "qcqp"):
        row = []
        for c in (0, 1):
            _capi.set_option("fwd_compact", c)
            if kind == "qp": f = lambda: ops.qp_forward(dh["P"], dh["q"], eps, 1000, out=x)
            else: f = lambda: ops.qcqp_forward(dh["P"], dh["q"], dh["l_n"], dh["mu"], eps, 1000, out=x)
            row.append("compact %d: %.1f us" % (c, t(f)))
        _, it = ops.qp_forward(dh["P"], dh["q"], eps, 1000, return_iters=True) if kind == "qp" else ops.qcqp_forward(dh["P"], dh["q"], dh["l_n"], dh["mu"], eps, 1000, return_iters=True)
        it = it.float()
        print("heavy tail eps %g %s: %s | iterations mean %.1f, mean of tile maxima %.1f, max %d" % (eps, kind, "  ".join(row), it.mean().item(), it.view(-1, 32).max(1).values.mean().item(), int(it.max().item())))
