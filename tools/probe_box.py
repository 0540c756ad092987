"""Timing of the box QP path at the bench shape: B=65536, N=8, diagonal P in (B,8,8)."""
import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from conftest import make_problem
from diffqcqp_amd import ops

def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n

for kind, B, N, st, layout in (("box", 65536, 8, "diag", 0), ("sbox", 65536, 8, "diag", 0), ("box", 32768, 32, "diag", 0),
                              ("box", 65536, 8, "dense", 0), ("box", 65536, 8, "dense", 1), ("box", 4096, 64, "dense", 1)):
    d = {k: v.cuda() for k, v in make_problem(kind, B, N, 1300, st).items()}
    cache = ops.diag_cache(d["q"])
    x = torch.empty_like(d["q"])
    f = lambda: ops.boxqp_forward(d["P"], d["q"], d["l_min"], d["l_max"], 1e-7, 1000, v=d.get("v"), out=x, cache=cache, layout=layout)
    tf = timed(f)
    _, it = ops.boxqp_forward(d["P"], d["q"], d["l_min"], d["l_max"], 1e-7, 1000, v=d.get("v"), return_iters=True, layout=layout)
    line = f"{kind} B={B} N={N} {st} layout={layout}: fwd {tf:.1f} us (iters mean {it.float().mean():.1f} max {int(it.max())})"
    if kind == "box" and N <= 21:
        out = tuple(torch.empty_like(t) for t in (d["P"], d["q"], d["q"], d["q"]))
        b = lambda: ops.boxqp_backward(d["P"], d["q"], d["l_min"], d["l_max"], x, d["grad_x"], out=out, cache=cache, layout=layout)
        line += f", bwd {timed(b):.1f} us"
    print(line)
