"""Headline step and its backwards with the backward's drain launch (default), without it (auto_fallback = 0: measurement only) and
with the general routine fused into the fast kernel (fuse_fallback = 1: one launch, correct for any P)."""
import os, sys, time, torch
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8"); os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from diffqcqp_amd import _capi
dev = torch.device("cuda", 0)
chains = [bench.Chain("qp", 65536, 8, "diag", True, dev, 1000), bench.Chain("qcqp", 65536, 8, "diag", True, dev, 1031)]
main_s, side = torch.cuda.current_stream(), torch.cuda.Stream()
st = [main_s.cuda_stream, side.cuda_stream]
def b2b(fn, n=50, reps=5):
    out = []
    for _ in range(reps):
        for _ in range(10): fn()
        torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): fn()
        b.record(); b.synchronize(); out.append(a.elapsed_time(b) * 1e3 / n)
    return sorted(out)[len(out) // 2]
def step():
    chains[1].launch(0, st[1]); chains[0].launch(0, st[0]); chains[1].launch(1, st[1]); chains[0].launch(1, st[0])
def step_time(k=100, reps=9):
    for _ in range(30): step()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(k): step()
        side.synchronize(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / k * 1e6)
    return sorted(ts)[len(ts) // 2]
for rnd in range(2):
    for name, fuse, fb in (("two launches (default)", -1, 1), ("no drain (unsafe, measurement)", -1, 0), ("fused backward, one launch", 1, 1),
                           ("nothing fused: lean forwards + their drains", 0, 1), ("lean forwards, no drains at all (unsafe)", 0, 0)):
        _capi.set_option("fuse_fallback", fuse); _capi.set_option("auto_fallback", fb)
        try:
            qb = b2b(lambda: chains[0].launch(1, st[0])); cb = b2b(lambda: chains[1].launch(1, st[0]))
            qf = b2b(lambda: chains[0].launch(0, st[0])); cf = b2b(lambda: chains[1].launch(0, st[0]))
            s = step_time()
        finally:
            _capi.set_option("fuse_fallback", -1); _capi.set_option("auto_fallback", 1)
        print("%-32s backward alone QP %.2f QCQP %.2f us; forward QP %.2f QCQP %.2f; two-stream step %.2f us" % (name, qb, cb, qf, cf, s), flush=True)
