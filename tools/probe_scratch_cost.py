"""What does a private (scratch) segment cost a kernel that never touches it?  The lean N = 8 forward (fuse_fallback = 0:
no general routine inside, no spills) against the same kernel with a dummy private array (-DDQQ_PROBE_SCRATCH=n,
tools/build_variant.sh), and against the shipped fused kernel (184 B of spill space in its never-taken dense branch).
    python tools/probe_scratch_cost.py            (DQQ_LIB selects the build)"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from diffqcqp_amd import _capi
dev = torch.device("cuda", 0)
chains = [bench.Chain("qp", 65536, 8, "diag", True, dev, 1000), bench.Chain("qcqp", 65536, 8, "diag", True, dev, 1031)]
main_s, side = torch.cuda.current_stream(), torch.cuda.Stream()
st = [main_s.cuda_stream, side.cuda_stream]

def ev(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); b.synchronize(); return a.elapsed_time(b) * 1e3 / n

def step():
    chains[0].launch(0, st[0]); chains[1].launch(0, st[1]); chains[0].launch(1, st[0]); chains[1].launch(1, st[1])

def step_time(k=100, reps=7):
    for _ in range(20): step()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(k): step()
        side.synchronize(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / k * 1e6)
    return sorted(ts)[len(ts) // 2]

print("lib", os.environ.get("DQQ_LIB", "shipped"))
for fuse, drains in ((-1, 1), (-1, 0), (0, 0), (0, 1)):
    _capi.set_option("fuse_fallback", fuse); _capi.set_option("auto_fallback", drains)
    r = {"qp_fwd": ev(lambda: chains[0].launch(0, st[0])), "qcqp_fwd": ev(lambda: chains[1].launch(0, st[0])),
         "qp_bwd": ev(lambda: chains[0].launch(1, st[0])), "qcqp_bwd": ev(lambda: chains[1].launch(1, st[0])), "step": step_time()}
    print("fuse_fallback %2d auto_fallback %d: " % (fuse, drains) + "  ".join("%s %.2f" % kv for kv in r.items()), flush=True)
_capi.set_option("fuse_fallback", -1); _capi.set_option("auto_fallback", 1)
