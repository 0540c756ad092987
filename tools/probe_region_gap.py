"""Does the fixed cost of a timed region of the headline step depend on how long the GPU idled before it (clock ramp)?
T(20 steps) after an idle gap of g microseconds (busy-wait on the host after the synchronize), and with a pre-roll of a few
untimed steps + synchronize right before t0.   python tools/probe_region_gap.py"""
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
    chains[1].launch(0, st[1]); chains[0].launch(0, st[0]); chains[1].launch(1, st[1]); chains[0].launch(1, st[0])
def sync():
    while not (main_s.query() and side.query()): pass
    side.synchronize(); torch.cuda.synchronize()
def region(K, gap_us=0.0, preroll=0):
    for _ in range(preroll): step()
    sync()
    if gap_us > 0:
        t = time.perf_counter()
        while (time.perf_counter() - t) * 1e6 < gap_us: pass
    t0 = time.perf_counter()
    for _ in range(K): step()
    sync()
    return (time.perf_counter() - t0) * 1e6
for _ in range(50): step()
sync()
out = {}
for K in (20, 100):
    for name, kw in (("gap0", {}), ("gap50us", {"gap_us": 50}), ("gap200us", {"gap_us": 200}), ("gap1ms", {"gap_us": 1000}),
                     ("gap10ms", {"gap_us": 10000}), ("preroll5", {"preroll": 5}), ("preroll20", {"preroll": 20})):
        ts = sorted(region(K, **kw) for _ in range(11))
        out["K%d_%s" % (K, name)] = {"us_per_step_median": ts[5] / K, "min": ts[0] / K}
print(json.dumps(out, indent=1))
