"""Re-sweep of the forward's tail thresholds on the final build: headline step (two streams) and the forwards alone."""
import os, sys, time, torch
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8"); os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from diffqcqp_amd import _capi
dev = torch.device("cuda", 0)
chains = [bench.Chain("qp", 65536, 8, "diag", True, dev, 1000), bench.Chain("qcqp", 65536, 8, "diag", True, dev, 1031)]
main_s, side = torch.cuda.current_stream(), torch.cuda.Stream()
st = [main_s.cuda_stream, side.cuda_stream]
def ev(fn, n=150):
    for _ in range(20): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); b.synchronize(); return a.elapsed_time(b) * 1e3 / n
def step():
    chains[1].launch(0, st[1]); chains[0].launch(0, st[0]); chains[1].launch(1, st[1]); chains[0].launch(1, st[0])
def step_time(k=100, reps=7):
    for _ in range(30): step()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(k): step()
        side.synchronize(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / k * 1e6)
    return sorted(ts)[len(ts) // 2]
for r1 in (0, 8, 12, 16):
    for r2 in (0, 4, 8):
        if r1 == 0 and r2: continue
        _capi.set_option("fwd_respread", r1); _capi.set_option("fwd_respread2", r2)
        print("respread %2d respread2 %d: qp_fwd %.2f qcqp_fwd %.2f step %.2f" % (r1, r2, ev(lambda: chains[0].launch(0, st[0])), ev(lambda: chains[1].launch(0, st[0])), step_time()), flush=True)
_capi.set_option("fwd_respread", 16); _capi.set_option("fwd_respread2", 8)
for wpb in (1, 4):
    _capi.set_option("wpb", wpb)
    print("wpb %d: qp_fwd %.2f qcqp_fwd %.2f step %.2f" % (wpb, ev(lambda: chains[0].launch(0, st[0])), ev(lambda: chains[1].launch(0, st[0])), step_time()), flush=True)
_capi.set_option("wpb", 0)
for lpp in (2, 4):
    _capi.set_option("fwd_lpp", lpp)
    print("fwd_lpp %d: qp_fwd %.2f qcqp_fwd %.2f step %.2f" % (lpp, ev(lambda: chains[0].launch(0, st[0])), ev(lambda: chains[1].launch(0, st[0])), step_time()), flush=True)
_capi.set_option("fwd_lpp", 0)
