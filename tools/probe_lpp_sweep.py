"""Developer probe: N = 8 diagonal forward, lanes per problem (option fwd_lpp) x batch size."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_problem
from diffqcqp_amd import _capi, ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
LPPS = {8: (1, 2, 4), 16: (2, 4, 8), 4: (1, 2), 32: (4, 8, 16), 64: (8, 16, 32)}[N]
def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); b.synchronize(); return a.elapsed_time(b) * 1e3 / n
for kind in ("qp", "qcqp"):
    for B in [b for b in (8192, 16384, 32768, 49152, 65536, 98304, 131072, 262144, 524288) if b * N * N * 8 <= (5 << 30)]:
        d = {k: v.cuda() for k, v in make_problem(kind, B, N, 11).items()}
        xo = torch.empty(B, N, 1, dtype=torch.float64, device="cuda")
        row = []
        for lpp in (0,) + LPPS + (0,):
            _capi.set_option("fwd_lpp", lpp)
            if kind == "qp": f = lambda: ops.qp_forward(d["P"], d["q"], 1e-7, 1000, out=xo)
            else: f = lambda: ops.qcqp_forward(d["P"], d["q"], d["l_n"], d["mu"], 1e-7, 1000, out=xo)
            row.append("lpp %d: %.1f us" % (lpp, t(f)))
        print(kind, "N", N, "B", B, " | ".join(row), flush=True)
        del d
_capi.set_option("fwd_lpp", 0)
