"""Developer probe: fused (in-kernel) vs work-list fallback of the forward fast path, small N, large batches."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_problem
from diffqcqp_amd import _capi, ops
def t(fn, n=20):
    for _ in range(8): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); b.synchronize(); return a.elapsed_time(b) * 1e3 / n
for N in [int(a) for a in sys.argv[1:]] or [2, 4]:
    for structure in ("diag", "dense"):
        for kind in ("qp", "qcqp"):
            row = []
            for B in (65536, 131072, 196608, 262144, 524288, 1048576):
                d = {k: v.cuda() for k, v in make_problem(kind, B, N, 5, structure=structure).items()}
                xo = torch.empty(B, N, 1, dtype=torch.float64, device="cuda")
                if kind == "qp": f = lambda: ops.qp_forward(d["P"], d["q"], 1e-7, 1000, out=xo)
                else: f = lambda: ops.qcqp_forward(d["P"], d["q"], d["l_n"], d["mu"], 1e-7, 1000, out=xo)
                r = []
                for fuse in (0, 1):
                    _capi.set_option("fuse_fallback", fuse)
                    r.append(t(f))
                _capi.set_option("fuse_fallback", -1)
                row.append("%dk: %.0f / %.0f" % (B // 1024, r[0], r[1]))
                del d
            print("N %d %-5s %-4s work-list / fused: %s" % (N, structure, kind, "  ".join(row)), flush=True)
