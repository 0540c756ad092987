"""Per-iteration / per-refactor cost of the workgroup dense kernel (dense_block.hip), N in {32, 64}.

Fixed iteration counts (eps = 0 never stops) with the rho adaptation off isolate the ADMM iteration;
max_iter = 1 isolates power iteration + one refactor.  Batch = one workgroup per slot (2 per CU).
"""
import sys, time, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from conftest import make_problem
from diffqcqp_amd import ops

def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3)
    return best

for N in (32, 64):
    for B in (256, 512):
        d = {k: v.cuda() for k, v in make_problem("qp", B, N, 1005, "dense").items()}
        t = {}
        for it in (1, 101, 301):
            t[it] = timed(lambda: ops.qp_forward(d["P"], d["q"], 0.0, it, adaptive_rho=False, layout=1))
        per_iter = (t[301] - t[101]) / 200
        print(f"N={N} B={B}: max_iter=1 {t[1]:.1f} us (PI + refactor + 1 iteration), "
              f"per iteration {per_iter*1e3:.0f} ns = {per_iter*2400:.0f} cycles @2.4GHz")
