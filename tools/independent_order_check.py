#!/usr/bin/env python3
"""Independent-order check of the oracle's TRAJECTORY (build container only: numpy / scipy, no GPU).

oracle/diffqcqp_oracle.c is a scalar-loop C restatement of the reference (Solver.cpp) whose parity with the real
reference cannot be pinned here (the reference needs Eigen, which this image does not have).  The real reference
evaluates the same formulas through Eigen: vectorised dot products, blocked LLT, `norm()` with SIMD partial sums,
gemv that folds scalars into operands.  Every one of those changes the ORDER of the floating-point sums, and the
quantities the tests pin -- ADMM iteration counts, rho schedule, iterative-refinement step counts -- are decided by
comparisons of such sums against thresholds.  This script evaluates the reference algorithm a SECOND time with a
deliberately different arithmetic: numpy `@` (OpenBLAS gemv / gemm, SIMD + blocked), `scipy.linalg.cholesky` (LAPACK
dpotrf) + `scipy.linalg.solve_triangular` against the identity for the explicit inverse (Solver.cpp:76-77),
`np.linalg.norm` for the 2-norms, `np.abs(.).max()` for the inf-norms, scalars folded the way Eigen's expression
templates do (`Kinv @ (mu * x)`), and reports how often the discrete outcomes survive.

Usage:  python tools/independent_order_check.py [--n 10000] [--out oracle/independent_order_check.json]
The summary table is kept in oracle/README.md.
"""
import argparse
import glob
import json
import os
import sys
import time

import numpy as np
import scipy.linalg as sla

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

ALPHA, MU_THRESH, EPS_REL = 1.5, 10.0, 1e-4


def chol_inv(M):
    """llt() + solveInPlace(Identity): LAPACK factor, two triangular solves against I."""
    L = sla.cholesky(M, lower=True, check_finite=False)
    Y = sla.solve_triangular(L, np.eye(M.shape[0]), lower=True, check_finite=False)
    return sla.solve_triangular(L.T, Y, lower=False, check_finite=False)


def power_iteration(P, its):
    n = P.shape[0]
    v = np.full(n, 1.0 / np.sqrt(n))
    nv = np.linalg.norm(v)
    if nv > 0:
        v = v / nv
    for _ in range(its):
        v = P @ v
        nv = np.linalg.norm(v)
        if nv > 0:
            v = v / nv
    return float(v @ (P @ v))


def solve(P, q, eps, max_iter, mu=1e-7, kind="qp", rad=None):
    """Solver::solveQP / solveQCQP (Solver.cpp:61-123 / 521-582), BLAS-ordered.  -> x, iterations, rho updates."""
    n = q.size
    qp_like = kind == "qp"
    L = power_iteration(P, 10 if qp_like else 100)
    rho = np.sqrt(mu * L) * (L / mu) ** 0.4
    tau_inc = tau_dec = (L / mu) ** 0.15
    M = np.tril(P) + np.tril(P, -1).T          # llt reads the lower triangle
    M = M + (rho + mu) * np.eye(n)
    Minv = chol_inv(M)
    u = np.zeros(n)
    l2 = np.zeros(n)
    l2p = np.zeros(n)
    qprox = q.copy()
    rho_up, cpt, nref, it = 0, 0, 0, 0
    for it in range(1, max_iter + 1):
        l = Minv @ (rho * l2 - u - qprox)
        qprox = q - mu * l
        z = ALPHA * l + (1 - ALPHA) * l2 + u / rho
        if qp_like:
            l2 = np.maximum(z, 0.0)
        else:
            l2 = z.copy()
            nrm = np.sqrt(z[0::2] ** 2 + z[1::2] ** 2)
            sc = np.where(nrm > rad, rad / np.where(nrm > 0, nrm, 1.0), 1.0)
            l2[0::2] *= sc
            l2[1::2] *= sc
        w = ALPHA * l + (1 - ALPHA) * l2p
        u = u + rho * (w - l2)
        res_dual = np.abs(rho * (l2 - l2p)).max() if qp_like else rho * np.abs(l2 - l2p).max()
        res_prim = np.abs(l2 - w).max()
        l2p = l2
        if qp_like:
            if res_dual < eps:
                break
        elif res_prim < eps + EPS_REL * np.linalg.norm(l) and res_dual < eps:
            break
        inc = res_prim > MU_THRESH * res_dual
        dec = (not inc) and res_dual > MU_THRESH * res_prim
        if inc or dec:
            if cpt % 5 == 0:
                if rho_up == (-1 if inc else 1):
                    ti, td = 1 + .8 * (tau_inc - 1), 1 + .8 * (tau_dec - 1)
                    if qp_like:
                        tau_inc, tau_dec = ti, td
                    elif inc:
                        tau_inc = ti
                    else:
                        tau_dec = td
                if inc:
                    M = M + rho * (tau_inc - 1) * np.eye(n)
                    rho *= tau_inc
                    rho_up = 1
                else:
                    M = M + rho * (1. / tau_dec - 1) * np.eye(n)
                    rho /= tau_dec
                    rho_up = -1
                Minv = chol_inv(M)
                nref += 1
            cpt += 1
    return l2, it, nref


def iterative_refinement(A, b, mu_ir=1e-7, eps=1e-10):
    """Solver::iterative_refinement (Solver.cpp:15-44) with gemm / gemv / LAPACK and Eigen-style scalar folding."""
    Ab = A.T @ b
    K = A.T @ A + mu_ir * np.eye(A.shape[1])
    Kinv = chol_inv(K)
    KinvAb = Kinv @ Ab
    x = np.zeros(A.shape[1])
    res_pred, not_improved, steps = np.finfo(float).max, 0, 0
    for steps in range(1, 11):
        x = Kinv @ (mu_ir * x) + KinvAb
        res = np.linalg.norm(K @ x - Ab)
        if res_pred - res < eps:
            not_improved += 1
        else:
            res_pred, not_improved = res, 0
        if res < eps or not_improved == 2:
            break
    return x, steps


def qp_backward(P, q, x, g):
    gamma = -(P @ x + q)
    gamma[x > 1e-10] = 0.0
    act = gamma < -1e-10
    ia, ii = np.nonzero(act)[0], np.nonzero(~act)[0]
    n, na = q.size, ia.size
    A = np.zeros((n, n))
    A[np.arange(na), np.arange(na)] = x[ia]
    A[na:, na:] = P[np.ix_(ii, ii)]
    A = A.T
    dd = np.concatenate([np.zeros(na), g[ii]])
    b, steps = iterative_refinement(A, dd)
    dl = np.zeros(n)
    dl[ii] = b[na:]
    return -dl, steps


def qcqp_backward(P, q, l_n, mu, x, g):
    n, nc = q.size, q.size // 2
    r = l_n * mu
    xa, xb = x[0::2], x[1::2]
    nrm = np.sqrt(xa * xa + xb * xb)
    plq = P @ x + q
    gamma = np.zeros(nc)
    for c in range(nc):
        if not (r[c] - nrm[c] > 1e-10 or r[c] < 1e-10):
            ca, cb = 2 * xa[c], 2 * xb[c]
            gamma[c] = -((ca * plq[2 * c] + cb * plq[2 * c + 1]) / (ca * ca + cb * cb))
    S = nrm * nrm - r * r
    act = [c for c in range(nc) if S[c] > -1e-10 and r[c] > 1e-10]
    na = len(act)
    A = np.zeros((na + n, na + n))
    for k, c in enumerate(act):
        A[k, k] = S[c]
        A[k, na + 2 * c] = gamma[c] * 2 * x[2 * c]
        A[k, na + 2 * c + 1] = gamma[c] * 2 * x[2 * c + 1]
        A[na + 2 * c, k] = 2 * x[2 * c]
        A[na + 2 * c + 1, k] = 2 * x[2 * c + 1]
    A[na:, na:] = P + np.diag(np.repeat(2 * gamma, 2))
    b, steps = iterative_refinement(A.T, np.concatenate([np.zeros(na), g]))
    return -b[na:], steps


def run_family(name, kind, d, O, eps=1e-7, max_iter=1000, backward=True):
    P, q = d["P"].numpy(), d["q"].numpy()[:, :, 0]
    B = q.shape[0]
    g = d["grad_x"].numpy()[:, :, 0]
    if kind == "qp":
        xo, ito = O.qp_fwd_batch(P, d["q"].numpy(), eps, max_iter, nthreads=8)
    else:
        ln, mu = d["l_n"].numpy()[:, :, 0], d["mu"].numpy()[:, :, 0]
        xo, ito = O.qcqp_fwd_batch(P, d["q"].numpy(), d["l_n"].numpy(), d["mu"].numpy(), eps, max_iter, nthreads=8)
    t0 = time.time()
    its = np.empty(B, dtype=np.int64)
    dx = np.empty(B)
    for i in range(B):
        x, its[i], _ = solve(P[i], q[i], eps, max_iter, kind=kind, rad=None if kind == "qp" else ln[i] * mu[i])
        dx[i] = np.abs(x - xo[i, :, 0]).max()
    xs = np.maximum(1.0, np.abs(xo).max(axis=(1, 2)))
    dx = dx / xs   # relative to the solution's scale (the reference's own m2 problem has x = 1.6e7)
    out = {"family": name, "problems": B, "N": q.shape[1],
           "iteration_count_agreement": float((its == ito).mean()),
           "iteration_count_max_abs_difference": int(np.abs(its - ito).max()),
           "x_max_abs_difference": float(dx.max()), "x_median_abs_difference": float(np.median(dx)),
           "x_max_abs_difference_where_counts_agree": float(dx[its == ito].max())}
    if backward:
        if kind == "qp":
            ref = O.qp_bwd_batch(P, d["q"].numpy(), xo, d["grad_x"].numpy(), nthreads=8)
        else:
            ref = O.qcqp_bwd_batch(P, d["q"].numpy(), d["l_n"].numpy(), d["mu"].numpy(), xo, d["grad_x"].numpy(), nthreads=8)
        st = np.empty(B, dtype=np.int64)
        dg = np.empty(B)
        for i in range(B):
            if kind == "qp":
                gq, st[i] = qp_backward(P[i], q[i], xo[i, :, 0], g[i])
            else:
                gq, st[i] = qcqp_backward(P[i], q[i], ln[i], mu[i], xo[i, :, 0], g[i])
            dg[i] = np.abs(gq - ref[1][i, :, 0]).max() / max(1.0, np.abs(ref[1][i]).max())
        same = st == ref[-1]
        # where the exits differ: the oracle made to run THIS evaluation's number of bodies
        dgf = np.zeros(B)
        for steps in np.unique(st[~same]):
            sel = np.nonzero((~same) & (st == steps))[0]
            O.set_force_ir_steps(int(steps))
            try:
                if kind == "qp":
                    fr = O.qp_bwd_batch(P[sel], d["q"].numpy()[sel], xo[sel], d["grad_x"].numpy()[sel], nthreads=8)
                else:
                    fr = O.qcqp_bwd_batch(P[sel], d["q"].numpy()[sel], d["l_n"].numpy()[sel], d["mu"].numpy()[sel], xo[sel],
                                          d["grad_x"].numpy()[sel], nthreads=8)
            finally:
                O.set_force_ir_steps(0)
            for k, i in enumerate(sel):
                gq = (qp_backward(P[i], q[i], xo[i, :, 0], g[i]) if kind == "qp" else
                      qcqp_backward(P[i], q[i], ln[i], mu[i], xo[i, :, 0], g[i]))[0]
                dgf[i] = np.abs(gq - fr[1][k, :, 0]).max() / max(1.0, np.abs(fr[1][k]).max())
        out["grad_q_max_rel_difference_at_this_evaluations_exit"] = float(dgf.max())
        out.update({"refinement_step_agreement": float(same.mean()),
                    "oracle_step_histogram": {int(k): int(v) for k, v in zip(*np.unique(ref[-1], return_counts=True))},
                    "grad_q_max_rel_difference_where_steps_agree": float(dg[same].max()) if same.any() else None,
                    "grad_q_median_rel_difference": float(np.median(dg))})
    out["seconds"] = round(time.time() - t0, 1)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=10000, help="problems per config distribution")
    ap.add_argument("--out", default=os.path.join(ROOT, "oracle", "independent_order_check.json"))
    args = ap.parse_args()
    import torch
    from conftest import make_problem
    from oracle import oracle as O
    res = []
    # every golden fixture (inputs only; the expected outputs in them are the oracle's own)
    for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*.npz"))):
        z = np.load(path)
        name = os.path.basename(path)
        if name.startswith(("box", "sbox")) or "P" not in z:
            continue
        kind = "qcqp" if "l_n" in z.files else "qp"
        d = {k: torch.from_numpy(np.ascontiguousarray(z[k])) for k in ("P", "q", "l_n", "mu") if k in z.files}
        B, N = d["q"].shape[0], d["q"].shape[1]
        d["q"] = d["q"].reshape(B, N, 1)
        for k in ("l_n", "mu"):
            if k in d:
                d[k] = d[k].reshape(B, N // 2, 1)
        d["grad_x"] = torch.from_numpy(np.ascontiguousarray(z["grad_x"])).reshape(B, N, 1) if "grad_x" in z.files else \
            torch.randn(B, N, 1, dtype=torch.float64, generator=torch.Generator().manual_seed(1))
        eps = float(z["eps"]) if "eps" in z.files else 1e-7
        mi = int(z["max_iter"]) if "max_iter" in z.files else 1000
        res.append(run_family("golden/" + name, kind, d, O, eps=eps, max_iter=mi))
        print(json.dumps(res[-1]), flush=True)
    # the BASELINE config distributions
    # the regime the reference's own G4 stands for: rank-deficient dense P (tests/golden/reference_inputs.py)
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import reference_inputs as R
    for fam in ("lowrank", "duprows"):
        for N, nb in ((8, 2000), (32, 1000), (64, 300)):
            for kind in ("qp", "qcqp"):
                d = R.rank_deficient(kind, min(nb, max(args.n // 5, 50)), N, 7000 + N, fam)
                res.append(run_family("rank-deficient %s %s N=%d" % (fam, kind.upper(), N), kind, d, O))
                print(json.dumps(res[-1]), flush=True)
    n = args.n
    fams = [("cfg2/3 QP diag N=8, p~U(.1,1.1)", "qp", 8, "diag", n), ("cfg3 QCQP diag N=8", "qcqp", 8, "diag", n),
            ("QP dense N=8 (S S^T/N + .1 I)", "qp", 8, "dense", n), ("QCQP dense N=8", "qcqp", 8, "dense", n),
            ("cfg4 QP diag N=32", "qp", 32, "diag", n), ("cfg5 QP dense N=64", "qp", 64, "dense", max(n // 5, 200))]
    for name, kind, N, structure, nb in fams:
        d = make_problem(kind, nb, N, 31000 + N + (7 if kind == "qcqp" else 0), structure)
        res.append(run_family(name, kind, d, O))
        print(json.dumps(res[-1]), flush=True)
    json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
