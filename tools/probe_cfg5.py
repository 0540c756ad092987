#!/usr/bin/env python3
"""Developer probe: BASELINE configs[4] (N=64 dense QP) forward / backward launch times at several batch sizes."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diffqcqp_amd import _capi, ops  # noqa: E402


def make(B, N=64, seed=1005):
    gen = torch.Generator(device="cuda").manual_seed(seed)
    S = torch.rand(B, N, N, generator=gen, dtype=torch.float64, device="cuda")
    P = torch.bmm(S, S.transpose(1, 2)) / N
    del S
    P.diagonal(dim1=1, dim2=2).add_(0.1)
    q = 2 * torch.rand(B, N, 1, generator=gen, dtype=torch.float64, device="cuda") - 1
    gx = torch.randn(B, N, 1, generator=gen, dtype=torch.float64, device="cuda")
    return P, q, gx


def timeit(fn, reps=5, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


if __name__ == "__main__":
    sizes = [int(s) for s in sys.argv[1:]] or [4096, 16384, 65536]
    for B in sizes:
        P, q, gx = make(B)
        for opt in (1, 0):
            _capi.set_option("dense_wave64", opt)
            x, it = ops.qp_forward(P, q, 1e-7, 1000, layout=1, return_iters=True)
            tf = timeit(lambda: ops.qp_forward(P, q, 1e-7, 1000, layout=1))
            tb = timeit(lambda: ops.qp_backward(P, q, x, gx, layout=1))
            print(json.dumps({"B": B, "dense_wave64": opt, "fwd_ms": round(tf, 3), "bwd_ms": round(tb, 3),
                              "iters_mean": round(float(it.float().mean()), 2), "iters_max": int(it.max()),
                              "fwdbwd_solves_per_s": round(B / (tf + tb) * 1e3)}), flush=True)
        _capi.set_option("dense_wave64", 1)
