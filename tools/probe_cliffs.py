"""Developer probe: default routes, time per launch over a grid of batch sizes -- looks for cliffs (a batch that
takes longer than a larger one, or a jump in time per problem).  python tools/probe_cliffs.py [N ...]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_problem
from diffqcqp_amd import ops
NS = [int(a) for a in sys.argv[1:]] or [2, 4, 8, 16, 32, 64]
def t(fn, n=20):
    for _ in range(6): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); b.synchronize(); return a.elapsed_time(b) * 1e3 / n
GRID = (1024, 2048, 4096, 8192, 16384, 32768, 40960, 49152, 57344, 65536, 81920, 98304, 131072, 196608, 262144, 524288)
for N in NS:
    for structure in ("diag", "dense"):
        for kind in ("qp", "qcqp"):
            rows = []
            for B in GRID:
                if B * N * N * 8 > (3 << 30): break
                d = {k: v.cuda() for k, v in make_problem(kind, B, N, 5, structure=structure).items()}
                xo = torch.empty(B, N, 1, dtype=torch.float64, device="cuda")
                if kind == "qp":
                    f = lambda: ops.qp_forward(d["P"], d["q"], 1e-7, 1000, out=xo)
                    g = lambda: ops.qp_backward(d["P"], d["q"], xo, d["grad_x"])
                else:
                    f = lambda: ops.qcqp_forward(d["P"], d["q"], d["l_n"], d["mu"], 1e-7, 1000, out=xo)
                    g = lambda: ops.qcqp_backward(d["P"], d["q"], d["l_n"], d["mu"], xo, d["grad_x"])
                rows.append((B, t(f), t(g)))
                del d
            for col, name in ((1, "fwd"), (2, "bwd")):
                flags = []
                for i, r in enumerate(rows):
                    per = r[col] / r[0]
                    worse_than_larger = any(r[col] > 1.15 * s[col] for s in rows[i + 1:])
                    jump = i > 0 and per > 1.4 * (rows[i - 1][col] / rows[i - 1][0]) and r[0] >= 16384
                    flags.append("!" if (worse_than_larger or jump) else "")
                print("N %2d %-5s %-4s %s: " % (N, structure, kind, name) + "  ".join("%dk:%.0f%s" % (r[0] // 1024, r[col], fl) for r, fl in zip(rows, flags)), flush=True)
