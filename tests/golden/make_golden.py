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
import reference_inputs as R  # noqa: E402

torch.set_default_dtype(torch.double)


def run(kind, d, eps, max_iter):
    P, q, g = d["P"].numpy(), d["q"].numpy(), d["grad_x"].double().numpy()
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


def stack(problems, kind, seed):
    """[(P, q[, radius])...] of one size -> a batch in the layouts of qcqp.py (radius as l_n with mu = 1, the way the
    C++ driver hands `l_n` straight to Solver::solveQCQP); grad_x: row 0 = e_1 (`grad_l3[1] = 1.`, Solver.cpp:771),
    the others seeded N(0,1) (the driver leaves `grad_l` uninitialised, Solver.cpp:723)."""
    n = problems[0][1].size
    B = len(problems)
    g = torch.Generator().manual_seed(seed)
    gx = torch.randn(B, n, 1, generator=g)
    gx[0] = 0.0
    gx[0, 1, 0] = 1.0
    d = {"P": torch.from_numpy(np.stack([p[0] for p in problems])).contiguous(),
         "q": torch.from_numpy(np.stack([p[1] for p in problems])).reshape(B, n, 1).contiguous(), "grad_x": gx}
    if kind == "qcqp":
        d["l_n"] = torch.from_numpy(np.stack([p[2] for p in problems])).reshape(B, n // 2, 1).contiguous()
        d["mu"] = torch.ones(B, n // 2, 1)
    return d


def reference_cases():
    """ref_*.npz: the matrices Solver::test() hard-codes (reference_inputs.py), every way that driver (or a line it
    comments out) solves them; rd_*.npz: seeded rank-deficient dense batches, the regime G4 stands for."""
    cases = {}
    Pm, qm, lm = R.m2_singular()
    Pf, qf, lf = R.m2_first()
    cases["ref_m2_qp"] = run("qp", stack([(Pm, qm), (Pf, qf)], "qp", 2001), 1e-10, 1000)            # :712
    cases["ref_m2_qp_maxiter1"] = run("qp", stack([(Pm, qm), (Pf, qf)], "qp", 2001), 1e-10, 1)      # :729
    cases["ref_m2_qcqp"] = run("qcqp", stack([(Pm, qm, lm), (Pf, qf, lf)], "qcqp", 2002), 1e-10, 1000)  # :711
    P2, q2 = R.g2_product()
    g = torch.Generator().manual_seed(2003)
    l_ng = ((2 * torch.rand(6, generator=g) - 1) + 1).numpy() * 0.1     # :803-804 (Random + Ones) * .1
    cases["ref_g2_qp"] = run("qp", stack([(P2, q2)], "qp", 2004), 1e-10, 10000)                     # :811
    cases["ref_g2_qcqp"] = run("qcqp", stack([(P2, q2, l_ng), (P2, q2, l_ng * 100000)], "qcqp", 2005), 1e-10,
                               100000)                                                             # :805, :813
    Pb, qb, radii = R.g_blockdiag()
    cases["ref_gblock_qp"] = run("qp", stack([(Pb, qb)], "qp", 2006), 1e-10, 10000)
    cases["ref_gblock_qcqp"] = run("qcqp", stack([(Pb, qb, r) for r in radii], "qcqp", 2007), 1e-10, 100000)  # :871
    P4, q4, l4 = R.g4_delassus()
    cases["ref_g4_qp"] = run("qp", stack([(P4, q4), (P4, -q4)], "qp", 2008), 1e-10, 1000)
    cases["ref_g4_qcqp"] = run("qcqp", stack([(P4, q4, l4), (P4, -q4, l4)], "qcqp", 2009), 1e-10, 1000)
    for fam in ("lowrank", "duprows"):
        for N, B in ((8, 24), (32, 8), (64, 3)):
            for kind in ("qp", "qcqp"):
                cases["rd_%s_%s_n%d" % (fam, kind, N)] = run(kind, R.rank_deficient(kind, B, N, 7000 + N, fam), 1e-7,
                                                             1000)
    return cases


def main():
    cases = reference_cases()
    cases.update({
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
    })
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
