import os, sys, time, hashlib, torch
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8"); os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from diffqcqp_amd import _capi, ops
dev = torch.device("cuda", 0)
chains = [bench.Chain("qp", 65536, 8, "diag", True, dev, 1000), bench.Chain("qcqp", 65536, 8, "diag", True, dev, 1031)]
main_s, side = torch.cuda.current_stream(), torch.cuda.Stream()
st = [main_s.cuda_stream, side.cuda_stream]
def ev(fn, n=300):
    for _ in range(20): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); b.synchronize(); return a.elapsed_time(b) * 1e3 / n
def step():
    chains[1].launch(0, st[1]); chains[0].launch(0, st[0]); chains[1].launch(1, st[1]); chains[0].launch(1, st[0])
def step_time(k=200, reps=7):
    for _ in range(50): step()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(k): step()
        side.synchronize(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / k * 1e6)
    return sorted(ts)[len(ts) // 2]
step(); torch.cuda.synchronize()
h = hashlib.md5(b"".join(c.sets[0]["x"].cpu().numpy().tobytes() for c in chains)).hexdigest()[:10]
print(os.environ.get("DQQ_LIB", "shipped").split("/")[-1], "qp_fwd %.2f qcqp_fwd %.2f step %.2f md5 %s" % (ev(lambda: chains[0].launch(0, st[0])), ev(lambda: chains[1].launch(0, st[0])), step_time(), h), flush=True)
