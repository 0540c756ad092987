"""Headline step with each family's batch split into S independent slices on their own streams (2 S chains):
does finer interleaving of the launch chains buy anything?   python tools/probe_split_streams.py"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

dev = torch.device("cuda", 0)
for S in (1, 2, 4):
    B = 65536 // S
    chains = []
    for i in range(S):
        chains.append(bench.Chain("qp", B, 8, "diag", True, dev, 1000 + i))
        chains.append(bench.Chain("qcqp", B, 8, "diag", True, dev, 1031 + i))
    streams = [torch.cuda.Stream() for _ in chains]
    def step():
        for c, s in zip(chains, streams):
            c.launch(0, s.cuda_stream)
        for c, s in zip(chains, streams):
            c.launch(1, s.cuda_stream)
    for _ in range(20): step()
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(100): step()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 100 * 1e6)
    ts.sort()
    # host-side launch cost alone
    t0 = time.perf_counter()
    for _ in range(100): step()
    host = (time.perf_counter() - t0) / 100 * 1e6
    torch.cuda.synchronize()
    print("slices per family %d (B=%d each, %d streams): us/step min %.2f med %.2f max %.2f; host launch time %.1f us/step"
          % (S, B, len(chains), ts[0], ts[3], ts[-1], host))
