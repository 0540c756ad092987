#!/usr/bin/env python3
"""The reference's own two workloads (its `test_script.py`), run on the HIP path:

  1. the finite-difference gradient check of `test_script.py:23-43` (n=2, torch.manual_seed(5),
     P = S S^T, q = -rand-0.1, eps=1e-12, max_iter=10000; analytic dP of x[1] vs central differences);
  2. the workload behind the only published figure (`qcqp_runtime.png`, `test_script.py:91-123`):
     ONE QCQP, N=8 (4 contacts), P = diag(exp(U(-10,10))), q ~ U(-1,1), l_n, mu ~ U(0,1), eps=1e-10,
     max_iter=1e6 -- forward and backward wall time through `QCQPFn2` (mean of 10, like timeit there),
     and the same problem family at B = 1 ... 65536 to show where a GPU batch pays.

  3. the finite-difference check the reference's C++ driver runs for the box QP (`Solver.cpp:802-853`:
     G = R R^T, l_min in [-1.5,-0.5], l_max in [0.5,1.5], gradient of x[1] w.r.t. q, l_min, l_max), through
     `BoxQPFn2` -- whose backward does not run in the reference's Python.  The
     analytic values are the reference's Tikhonov-regularised solve (mu_ir = 1e-7, Solver.cpp:15-44): they
     track the finite differences to a percent or so when a bound is active, by construction.

Published CPU numbers read off the figure: forward ~9e-5 s, backward ~2.7e-4 s at B=1.
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffqcqp_amd.qcqp import BoxQPFn2, QCQPFn2, QPFn2  # noqa: E402  (sets the default dtype to float64)


def fd_check():
    torch.manual_seed(5)
    n = 2
    S = torch.rand(1, n, n) + 0.01
    P = torch.bmm(S, S.transpose(1, 2)).cuda().requires_grad_(True)
    q = (-torch.rand((1, n, 1)) - 0.1).cuda()
    ws = torch.zeros_like(q)
    lf = QPFn2.apply(P, q, ws, 1e-12, 10000)
    lf[0, 1].backward()
    print("x        ", lf.detach().cpu().flatten().tolist())
    print("Pgrad    ", P.grad.cpu().flatten().tolist())
    with torch.no_grad():
        num = torch.zeros(n, n)
        for i in range(n):
            for j in range(n):
                d = torch.zeros_like(P)
                d[0, i, j] = 1e-8
                num[i, j] = (QPFn2.apply(P + d, q, ws, 1e-12, 10000)[0, 1] - QPFn2.apply(P - d, q, ws, 1e-12, 10000)[0, 1]).item() / 2e-8
    print("grad_num ", num.flatten().tolist())


def box_fd_check():
    torch.manual_seed(7)
    n = 4
    R = 2 * torch.rand(1, n, n) - 1
    P = (torch.bmm(R, R.transpose(1, 2)) + torch.eye(n)).cuda()
    leaves = {"q": (0.9 * torch.rand(1, n, 1) - 0.45).cuda().requires_grad_(True),
              "l_min": (-(torch.rand(1, n, 1) + 0.5) * 0.25).cuda().requires_grad_(True),
              "l_max": ((torch.rand(1, n, 1) + 0.5) * 0.25).cuda().requires_grad_(True)}
    ws = torch.zeros(1, n, 1).cuda()

    def solve(**over):
        a = {k: over.get(k, v) for k, v in leaves.items()}
        return BoxQPFn2.apply(P, a["q"], a["l_min"], a["l_max"], ws, 1e-12, 100000)

    x = solve()
    x[0, 1].backward()
    print("box x    ", x.detach().cpu().flatten().tolist())
    with torch.no_grad():
        for name, t in leaves.items():
            num = []
            for i in range(n):
                d = torch.zeros_like(t)
                d[0, i, 0] = 1e-6
                num.append((solve(**{name: t + d})[0, 1] - solve(**{name: t - d})[0, 1]).item() / 2e-6)
            print("grad %-6s" % name, [round(v, 6) for v in t.grad.cpu().flatten().tolist()], " FD", [round(v, 6) for v in num])


def figure_workload():
    g = torch.Generator().manual_seed(0)
    print("%8s %14s %14s %16s" % ("B", "forward [us]", "backward [us]", "fwd+bwd solves/s"))
    for B in (1, 16, 256, 4096, 65536):
        P = torch.diag_embed(torch.exp(torch.rand(B, 8, generator=g) * 20 - 10)).cuda().requires_grad_(True)
        q = (torch.rand(B, 8, 1, generator=g) * 2 - 1).cuda().requires_grad_(True)
        l_n = torch.rand(B, 4, 1, generator=g).cuda().requires_grad_(True)
        mu = torch.rand(B, 4, 1, generator=g).cuda().requires_grad_(True)
        ws = torch.rand(B, 8, 1).cuda()
        target = torch.ones(B, 8, 1).cuda()

        def fwd():
            return QCQPFn2.apply(P, q, l_n, mu, ws, 1e-10, 1000000)

        for _ in range(3):
            ((fwd() - target) ** 2).mean().backward()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            l1 = fwd()
        torch.cuda.synchronize()
        tf = (time.perf_counter() - t0) / 10
        L = ((l1 - target) ** 2).mean()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            L.backward(retain_graph=True)
        torch.cuda.synchronize()
        tb = (time.perf_counter() - t0) / 10
        print("%8d %14.1f %14.1f %16.3g" % (B, tf * 1e6, tb * 1e6, B / (tf + tb)))


if __name__ == "__main__":
    fd_check()
    box_fd_check()
    figure_workload()
