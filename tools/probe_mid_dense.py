"""Dense 8 < N <= 16 (5-8 contacts): forward / backward timings, B=65536."""
import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from conftest import make_problem
from diffqcqp_amd import ops, _capi
def timed(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
B = 65536
for kind in ("qp", "qcqp"):
    for N in (10, 12, 16):
        d = {k: v.cuda() for k, v in make_problem(kind, B, N, 1500, "dense").items()}
        if kind == "qp":
            f = lambda: ops.qp_forward(d["P"], d["q"], 1e-7, 1000, layout=1)
            x = f()
            b = lambda: ops.qp_backward(d["P"], d["q"], x, d["grad_x"], layout=1)
        else:
            f = lambda: ops.qcqp_forward(d["P"], d["q"], d["l_n"], d["mu"], 1e-7, 1000, layout=1)
            x = f()
            b = lambda: ops.qcqp_backward(d["P"], d["q"], d["l_n"], d["mu"], x, d["grad_x"], layout=1)
        line = f"{kind} N={N} dense B={B}:"
        for opt in (1, 0):
            _capi.set_option("small_fwd", opt)
            line += f"  fwd[small_fwd={opt}] {timed(f):.0f} us"
        _capi.set_option("small_fwd", 1)
        print(line + f"  bwd {timed(b):.0f} us")
