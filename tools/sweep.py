#!/usr/bin/env python3
"""Tuning sweep (developer tool): mean launch time of every lanes-per-problem /
waves-per-workgroup variant of the diagonal kernels, for the bench shapes.
Writes JSON lines to stdout.  Usage: python tools/sweep.py [N B] ..."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_problem  # noqa: E402
from diffqcqp_amd import _capi, ops  # noqa: E402

LPPS = {2: (1,), 4: (1, 2), 8: (1, 2, 4), 16: (2, 4, 8), 32: (4, 8, 16), 64: (8, 16, 32)}


def timeit(fn, reps=50):
    for _ in range(5):
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
    return sum(ts) / len(ts), ts[len(ts) // 2], ts[0]


def main():
    shapes = [(8, 16384), (8, 65536), (8, 262144), (32, 32768)]
    if len(sys.argv) > 2:
        shapes = [(int(sys.argv[i]), int(sys.argv[i + 1])) for i in range(1, len(sys.argv) - 1, 2)]
    _capi.set_option("auto_fallback", 0)
    for N, B in shapes:
        d = {k: v.cuda() for k, v in make_problem("qcqp", B, N, 1002).items()}
        pd = torch.diagonal(d["P"], dim1=1, dim2=2).contiguous()
        x = torch.empty(B, N, 1, dtype=torch.float64, device="cuda")
        for kind in ("qp", "qcqp"):
            for wpb in (1, 4):
                for lpp in LPPS[N]:
                    for layout in (0, 2):
                        _capi.set_option("fwd_lpp", lpp)
                        _capi.set_option("wpb", wpb)
                        Pin = d["P"] if layout == 0 else pd
                        if kind == "qp":
                            fn = lambda: ops.qp_forward(Pin, d["q"], 1e-7, 1000, layout=layout, out=x)
                        else:
                            fn = lambda: ops.qcqp_forward(Pin, d["q"], d["l_n"], d["mu"], 1e-7, 1000, layout=layout, out=x)
                        mean, med, mn = timeit(fn)
                        print(json.dumps({"op": kind + "_fwd", "N": N, "B": B, "lpp": lpp, "wpb": wpb,
                                          "layout": layout, "mean_us": mean, "median_us": med, "min_us": mn}), flush=True)
            xs = ops.qp_forward(d["P"], d["q"], 1e-7, 1000) if kind == "qp" else ops.qcqp_forward(d["P"], d["q"], d["l_n"], d["mu"], 1e-7, 1000)
            for wpb in (1, 4):
                _capi.set_option("wpb", wpb)
                for layout in (0, 2):
                    Pin = d["P"] if layout == 0 else pd
                    if kind == "qp":
                        outs = ops.qp_backward(Pin, d["q"], xs, d["grad_x"], layout=layout)
                        fn = lambda: ops.qp_backward(Pin, d["q"], xs, d["grad_x"], layout=layout, out=outs)
                    else:
                        outs = ops.qcqp_backward(Pin, d["q"], d["l_n"], d["mu"], xs, d["grad_x"], layout=layout)
                        fn = lambda: ops.qcqp_backward(Pin, d["q"], d["l_n"], d["mu"], xs, d["grad_x"], layout=layout, out=outs)
                    mean, med, mn = timeit(fn)
                    print(json.dumps({"op": kind + "_bwd", "N": N, "B": B, "wpb": wpb, "layout": layout,
                                      "mean_us": mean, "median_us": med, "min_us": mn}), flush=True)
    _capi.set_option("auto_fallback", 1)
    _capi.set_option("fwd_lpp", 0)
    _capi.set_option("wpb", 0)


if __name__ == "__main__":
    main()
