"""What are the fixed ~100 us of a timed region of the headline step?  Host clock around [enqueue K steps, synchronize]
against HIP events recorded as the first / last work of the region, for the plain synchronize() and for polling
event.query() before it.   python tools/probe_region_dissect.py"""
import json, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

dev = torch.device("cuda", 0)
from diffqcqp_amd import build
build.build()
chains = [bench.Chain("qp", 65536, 8, "diag", True, dev, 1000), bench.Chain("qcqp", 65536, 8, "diag", True, dev, 1031)]
main_s, side = torch.cuda.current_stream(), torch.cuda.Stream()
st = [main_s.cuda_stream, side.cuda_stream]

def step():
    chains[0].launch(0, st[0]); chains[1].launch(0, st[1]); chains[0].launch(1, st[0]); chains[1].launch(1, st[1])

for _ in range(30): step()
torch.cuda.synchronize()
out = {}
for mode in ("sync", "poll"):
    for K in (1, 5, 20):
        rows = []
        for rep in range(9):
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            e0.record(main_s)
            for _ in range(K): step()
            e1.record(main_s); e2.record(side)
            t1 = time.perf_counter()
            if mode == "poll":
                while not (e1.query() and e2.query()): pass
            side.synchronize(); torch.cuda.synchronize()
            t2 = time.perf_counter()
            gpu = max(e0.elapsed_time(e1), e0.elapsed_time(e2)) * 1e3
            rows.append(((t2 - t0) * 1e6, (t1 - t0) * 1e6, gpu))
        rows.sort()
        h, q, g = rows[len(rows) // 2]
        out["%s_K%d" % (mode, K)] = {"host_us": h, "enqueue_us": q, "gpu_span_us": g, "host_minus_gpu_us": h - g}
print(json.dumps(out, indent=1))
