"""Developer probe: per-call device and host time of many identical launches -- looks for rare long stalls."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_problem
from diffqcqp_amd import ops
def run(kind, N, B, what, prealloc, n=1500):
    d = {k: v.cuda() for k, v in make_problem(kind, B, N, 5).items()}
    xo = torch.empty(B, N, 1, dtype=torch.float64, device="cuda")
    outs = None
    if kind == "qp":
        f = lambda: ops.qp_forward(d["P"], d["q"], 1e-7, 1000, out=xo)
        g = lambda: ops.qp_backward(d["P"], d["q"], xo, d["grad_x"])
    else:
        f = lambda: ops.qcqp_forward(d["P"], d["q"], d["l_n"], d["mu"], 1e-7, 1000, out=xo)
        g = lambda: ops.qcqp_backward(d["P"], d["q"], d["l_n"], d["mu"], xo, d["grad_x"])
    f(); fn = f if what == "fwd" else g
    for _ in range(5): fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    host = []
    for a, b in ev:
        t0 = time.perf_counter(); a.record(); fn(); b.record(); host.append((time.perf_counter() - t0) * 1e6)
        if len(host) % 50 == 0: torch.cuda.synchronize()
    torch.cuda.synchronize()
    dev = sorted((a.elapsed_time(b) * 1e3, i) for i, (a, b) in enumerate(ev))
    hs = sorted(host)
    print("%s N %d B %d %s: device us median %.1f p99 %.1f max %.1f (call %d) | host us median %.1f p99 %.1f max %.1f" % (
        kind, N, B, what, dev[n // 2][0], dev[int(n * .99)][0], dev[-1][0], dev[-1][1], hs[n // 2], hs[int(n * .99)], hs[-1]), flush=True)
for args in (("qp", 2, 98304, "bwd"), ("qcqp", 4, 65536, "bwd"), ("qcqp", 64, 2048, "bwd"), ("qp", 8, 32768, "fwd"), ("qcqp", 8, 65536, "fwd"), ("qcqp", 8, 65536, "bwd")):
    for rep in range(2):
        run(*args, prealloc=False)
import gc
print("gc counts", gc.get_count(), "objects tracked", len(gc.get_objects()))
t0 = time.perf_counter(); gc.collect(); print("full collection: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
gc.disable()
for rep in range(6):
    run("qcqp", 4, 65536, "bwd", prealloc=False)
