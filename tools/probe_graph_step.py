#!/usr/bin/env python3
"""Headline step (bench.py workload 0) launched eagerly vs replayed from a HIP graph that holds U steps
(two capture streams forked from / joined to the capture origin once per graph, so that steps pipeline across the
two chains inside a graph exactly as the eager launches do).   python tools/probe_graph_step.py [U ...]"""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    chains = [bench.Chain("qp", 65536, 8, "diag", True, dev, 1000), bench.Chain("qcqp", 65536, 8, "diag", True, dev, 1031)]
    main_s, side = torch.cuda.Stream(), torch.cuda.Stream()
    unrolls = [int(a) for a in sys.argv[1:]] or [1, 4, 10, 25]

    def step(s0, s1):
        chains[0].launch(0, s0)
        chains[1].launch(0, s1)
        chains[0].launch(1, s0)
        chains[1].launch(1, s1)

    def timed(fn, calls, steps_per_call):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(calls):
                fn()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / (calls * steps_per_call) * 1e6)
        ts.sort()
        return ts[0], ts[len(ts) // 2], ts[-1]

    # workspaces are per stream: resolve them before capture (no allocation inside)
    for c in chains:
        c.workspace(main_s.cuda_stream)
        c.workspace(side.cuda_stream)
    with torch.cuda.stream(main_s):
        print("eager two streams   us/step min/med/max: %.2f %.2f %.2f"
              % timed(lambda: step(main_s.cuda_stream, side.cuda_stream), 100, 1))
        print("eager one stream    us/step min/med/max: %.2f %.2f %.2f"
              % timed(lambda: step(main_s.cuda_stream, main_s.cuda_stream), 100, 1))
    for U in unrolls:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=main_s):
            side.wait_stream(main_s)
            for _ in range(U):
                step(main_s.cuda_stream, side.cuda_stream)
            main_s.wait_stream(side)
        print("graph of %3d steps  us/step min/med/max: %.2f %.2f %.2f" % ((U,) + timed(g.replay, max(100 // U, 4), U)))
        g1 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g1, stream=main_s):
            for _ in range(U):
                step(main_s.cuda_stream, main_s.cuda_stream)
        print("graph of %3d steps, one stream:          %.2f %.2f %.2f" % ((U,) + timed(g1.replay, max(100 // U, 4), U)))


if __name__ == "__main__":
    main()
