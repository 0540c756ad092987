#!/usr/bin/env python3
"""Developer probe: launch times of the non-headline configurations (BASELINE configs[3], [4] and dense N=8)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_problem  # noqa: E402
from diffqcqp_amd import _capi, ops  # noqa: E402


def timeit(fn, reps=20, warm=3):
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
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def run(tag, kind, B, N, structure, layout, reps=20):
    d = {k: v.cuda() for k, v in make_problem(kind, B, N, 1000 + N, structure).items()}
    if kind == "qp":
        f = lambda: ops.qp_forward(d["P"], d["q"], 1e-7, 1000, layout=layout)
        x, it = ops.qp_forward(d["P"], d["q"], 1e-7, 1000, layout=layout, return_iters=True)
        b = lambda: ops.qp_backward(d["P"], d["q"], x, d["grad_x"], layout=layout)
    else:
        f = lambda: ops.qcqp_forward(d["P"], d["q"], d["l_n"], d["mu"], 1e-7, 1000, layout=layout)
        x, it = ops.qcqp_forward(d["P"], d["q"], d["l_n"], d["mu"], 1e-7, 1000, layout=layout, return_iters=True)
        b = lambda: ops.qcqp_backward(d["P"], d["q"], d["l_n"], d["mu"], x, d["grad_x"], layout=layout)
    tf, tb = timeit(f, reps), timeit(b, reps)
    bytes_f = (N * N + 2 * N) * 8 * B
    bytes_b = (2 * N * N + 4 * N) * 8 * B
    print(json.dumps({"tag": tag, "kind": kind, "B": B, "N": N, "P": structure, "layout": layout,
                      "fwd_us": round(tf, 1), "bwd_us": round(tb, 1), "iters_mean": float(it.float().mean()),
                      "iters_max": int(it.max()), "fwd_GBps": round(bytes_f / tf / 1e3, 1),
                      "bwd_GBps": round(bytes_b / tb / 1e3, 1), "fwd_solves_per_s": round(B / tf * 1e6)}), flush=True)


if __name__ == "__main__":
    run("cfg4 shard", "qp", 32768, 32, "diag", 0)
    run("cfg4 shard qcqp", "qcqp", 32768, 32, "diag", 0)
    run("n8 dense auto", "qp", 65536, 8, "dense", 0)
    run("n8 dense layout", "qp", 65536, 8, "dense", 1)
    run("n8 dense qcqp", "qcqp", 65536, 8, "dense", 1)
    run("n32 dense", "qp", 8192, 32, "dense", 1, reps=5)
    run("cfg5 1/16", "qp", 4096, 64, "dense", 1, reps=3)
