"""Dense small-N backward: statically sized team kernel (bwd_small.hip) vs the run-time sized wave/team kernel."""
import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from conftest import make_problem
from diffqcqp_amd import ops, _capi

def timed(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n

B = 65536
for kind in ("qp", "qcqp", "box"):
    for N in (4, 8):
        d = {k: v.cuda() for k, v in make_problem(kind, B, N, 1400, "dense").items()}
        x = d["q"].clone().abs() * 0.3
        line = f"{kind} N={N} B={B} dense:"
        for opt in (1, 0):
            _capi.set_option("small_bwd", opt)
            if kind == "qp":
                f = lambda: ops.qp_backward(d["P"], d["q"], x, d["grad_x"], layout=1)
            elif kind == "qcqp":
                f = lambda: ops.qcqp_backward(d["P"], d["q"], d["l_n"], d["mu"], x, d["grad_x"], layout=1)
            else:
                f = lambda: ops.boxqp_backward(d["P"], d["q"], d["l_min"], d["l_max"], x, d["grad_x"], layout=1)
            line += f"  small_bwd={opt}: {timed(f):.1f} us"
        _capi.set_option("small_bwd", 1)
        print(line)
