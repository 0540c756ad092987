"""Generates tests/golden/*.npz -- seeded inputs + the oracle's outputs.

PROVENANCE: the reference (quentinll/diffqcqp) cannot be built or imported in
this image (no Eigen, empty pybind11 submodule), so these are NOT outputs of the
reference binary; they are outputs of oracle/diffqcqp_oracle.c (our C
restatement of qcqplib/Solver.cpp), generated here with this script:

    python tests/golden/make_golden.py

Input distributions follow SURVEY.md 8(d) / the reference's scripts:
  README.md:35-38 (P=diag(U(0,1)), q=U(0,1)), test_script.py:23-29 (seed-5, n=2),
  test_script.py:91-102 (P=diag(exp(U(-10,10))), q=U(-1,1), l_n,mu=U(0,1)).
The fixtures let the GPU box check (a) oracle-vs-fixture (the oracle builds
there too) and (b) HIP-vs-fixture without the reference.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
from conftest import make_problem  # noqa: E402
from oracle import oracle as O  # noqa: E402

torch.set_default_dtype(torch.double)


def run(kind, d, eps, max_iter):
    P, q, g = d["P"].numpy(), d["q"].numpy(), d["grad_x"].numpy()
    out = {"P": P, "q": q, "grad_x": g, "eps": eps, "max_iter": max_iter}
    if kind == "qp":
        x, it = O.qp_fwd_batch(P, q, eps, max_iter)
        gP, gq, st = O.qp_bwd_batch(P, q, x, g)
        out.update(x=x, iters=it, grad_P=gP, grad_q=gq, ir_steps=st)
    elif kind in ("box", "sbox"):
        lo, hi = d["l_min"].numpy(), d["l_max"].numpy()
        out.update(l_min=lo, l_max=hi)
        if kind == "sbox":  # forward only (no backward in the reference, qcqp.py:111)
            x, it = O.boxqp_fwd_batch(P, q, lo, hi, eps, max_iter, v=d["v"].numpy())
            out.update(v=d["v"].numpy(), x=x, iters=it)
        else:
            x, it = O.boxqp_fwd_batch(P, q, lo, hi, eps, max_iter)
            gP, gq, glo, ghi, gam, st = O.boxqp_bwd_batch(P, q, lo, hi, x, g)
            out.update(x=x, iters=it, grad_P=gP, grad_q=gq, grad_l_min=glo, grad_l_max=ghi, gamma=gam, ir_steps=st)
    else:
        ln, mu = d["l_n"].numpy(), d["mu"].numpy()
        x, it = O.qcqp_fwd_batch(P, q, ln, mu, eps, max_iter)
        gP, gq, gl, gm, st = O.qcqp_bwd_batch(P, q, ln, mu, x, g)
        out.update(l_n=ln, mu=mu, x=x, iters=it, grad_P=gP, grad_q=gq, grad_l_n=gl, grad_mu=gm, ir_steps=st)
    return out


def main():
    cases = {
        "qp_diag_n8": run("qp", make_problem("qp", 48, 8, 1002), 1e-7, 1000),
        "qcqp_diag_n8": run("qcqp", make_problem("qcqp", 48, 8, 1003), 1e-7, 1000),
        "qp_diag_n32": run("qp", make_problem("qp", 12, 32, 1004), 1e-7, 1000),
        "qcqp_diag_n32": run("qcqp", make_problem("qcqp", 12, 32, 1014), 1e-7, 1000),
        "qp_dense_n8": run("qp", make_problem("qp", 24, 8, 1105, "dense"), 1e-7, 1000),
        "qcqp_dense_n8": run("qcqp", make_problem("qcqp", 24, 8, 1106, "dense"), 1e-7, 1000),
        "qp_dense_n64": run("qp", make_problem("qp", 3, 64, 1005, "dense"), 1e-7, 1000),
        "qp_stress_n8": run("qp", make_problem("qp", 48, 8, 1007, p_lo=0.0), 1e-7, 1000),
        # Solver::solveBoxQP / solveSignedBoxQP / solveDerivativesBoxQP
        "box_diag_n8": run("box", make_problem("box", 48, 8, 1201), 1e-7, 1000),
        "box_dense_n8": run("box", make_problem("box", 24, 8, 1202, "dense"), 1e-7, 1000),
        "sbox_diag_n8": run("sbox", make_problem("sbox", 48, 8, 1203), 1e-7, 1000),
    }
    # README.md:35-38 verbatim: degenerate (q >= 0 => x = 0 after one iteration)
    g = torch.Generator().manual_seed(1001)
    B, N = 10, 8
    d = {"P": torch.diag_embed(torch.rand(B, N, generator=g)), "q": torch.rand(B, N, 1, generator=g),
         "grad_x": torch.ones(B, N, 1)}
    cases["qp_readme"] = run("qp", d, 1e-7, 1000)
    # test_script.py:23-29: the only quasi known-answer inputs of the reference
    torch.manual_seed(5)
    S = torch.rand(1, 2, 2) + 0.01
    P5 = torch.bmm(S, S.transpose(1, 2))
    q5 = -torch.rand((1, 2, 1)) - 0.1
    g5 = torch.zeros(1, 2, 1)
    g5[0, 1, 0] = 1.0  # lf[0,1].backward()
    cases["qp_seed5"] = run("qp", {"P": P5, "q": q5, "grad_x": g5}, 1e-12, 10000)
    # test_script.py:91-113: ill-conditioned QCQP workload of the published figure
    g = torch.Generator().manual_seed(1008)
    B = 24
    Pd = torch.diag_embed(torch.exp(torch.rand(B, 8, generator=g) * 20 - 10))
    d = {"P": Pd, "q": torch.rand(B, 8, 1, generator=g) * 2 - 1, "l_n": torch.rand(B, 4, 1, generator=g),
         "mu": torch.rand(B, 4, 1, generator=g), "grad_x": torch.randn(B, 8, 1, generator=g)}
    cases["qcqp_figure_n8"] = run("qcqp", d, 1e-10, 1000000)
    for name, c in cases.items():
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **c)
        print("%-16s B=%d N=%d iters mean %.1f max %d ir_steps %s" % (
            name, c["q"].shape[0], c["q"].shape[1], c["iters"].mean(), c["iters"].max(),
            np.bincount(np.asarray(c["ir_steps"]).reshape(-1)).tolist() if "ir_steps" in c else "-"))


if __name__ == "__main__":
    main()
