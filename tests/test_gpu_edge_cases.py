"""GPU parity on the edge cases the reference's code paths spell out (run with -m gpu): zero-radius contacts,
adaptative_rho=False, a non-symmetric P, max_iter exhaustion, N = 1 through the unbatched twin, and the QCQP
refinement exits (1 vs 3 loop bodies) evaluated at BOTH exits against the reference formula and the exact KKT
derivative."""
import numpy as np
import pytest
import torch

from conftest import make_problem, knob
from test_gpu_parity import (check_backward_exact, check_dense_backward, check_forward, dev, hip_bwd, hip_fwd, npy, oracle_bwd, oracle_fwd,
                             ops)  # noqa: F401  (ops is a fixture)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,structure,layout", [(8, "diag", 0), (8, "dense", 1), (32, "diag", 0), (16, "dense", 1),
                                                (64, "dense", 1), (8, "mixed", 0)])
def test_zero_radius_contacts(oracle, ops, N, structure, layout):
    """l_n = 0 or mu = 0 makes a contact's disk a point: prox_circle scales by 0/|.| (Solver.cpp:505-519), the dual
    recovery skips the contact through `l_n(i) < epsilon` (:597) and the derivative system drops it through
    `l_n(i) > 1e-10` (:639).  Also a whole problem with every radius zero (x = 0)."""
    B = 96
    d = make_problem("qcqp", B, N, 4100 + N, structure)
    d["l_n"][0::4, 0, 0] = 0.0          # first contact: zero normal force
    d["mu"][1::4, -1, 0] = 0.0          # last contact: zero friction coefficient
    d["l_n"][2] = 0.0                   # problem 2: every radius zero
    d["mu"][3] = 0.0                    # problem 3: likewise through mu
    g = dev(d)
    xo, ito = oracle_fwd(oracle, "qcqp", d)
    xh, ith = hip_fwd(ops, "qcqp", g, layout=layout)
    check_forward(xh, ith, xo, ito, min_match=0.98)
    assert np.all(npy(xh)[2] == 0.0) and np.all(npy(xh)[3] == 0.0)
    assert np.all(npy(xh)[0::4, 0:2, 0] == 0.0)
    grads, st = hip_bwd(ops, "qcqp", g, torch.from_numpy(xo).cuda(), layout=layout)
    if structure == "diag":
        check_backward_exact(grads, st, oracle_bwd(oracle, "qcqp", d, xo), exact=True)
    else:
        check_dense_backward(oracle, "qcqp", N, d, xo, grads, st, oracle_bwd(oracle, "qcqp", d, xo))
    assert np.all(npy(grads[2])[0::4, 0, 0] == 0.0) and np.all(npy(grads[3])[1::4, -1, 0] == 0.0)


@pytest.mark.parametrize("kind", ["qp", "qcqp", "box", "sbox"])
@pytest.mark.parametrize("N,structure,layout", [(8, "diag", 0), (8, "dense", 1), (12, "dense", 1), (32, "dense", 1),
                                                (64, "dense", 1), (20, "dense", 1)])
def test_fixed_rho(oracle, ops, kind, N, structure, layout):
    """adaptative_rho=False (pybindings.cpp:76-82 kwarg; Solver.cpp:90 `if (adaptative_rho)`): rho stays at its
    initial value, no refactorisation, usually many more iterations.  All four kinds, fast path and general path."""
    B = 40
    d = make_problem(kind, B, N, 4200 + N, structure)
    g = dev(d)
    P, q = d["P"].numpy(), d["q"].numpy()
    xo, ito = np.empty((B, N, 1)), np.empty(B, dtype=np.int64)
    for i in range(B):
        if kind == "qp":
            r = oracle.solveQP(P[i], q[i], None, 1e-7, 1e-7, 4000, False, return_iters=True)
        elif kind == "qcqp":
            r = oracle.solveQCQP(P[i], q[i], d["l_n"][i].numpy(), d["mu"][i].numpy(), None, 1e-7, 1e-7, 4000, False,
                                 return_iters=True)
        elif kind == "box":
            r = oracle.solveBoxQP(P[i], q[i], d["l_min"][i].numpy(), d["l_max"][i].numpy(), None, 1e-7, 1e-7, 4000, False,
                                  return_iters=True)
        else:
            r = oracle.solveSignedBoxQP(P[i], q[i], d["l_min"][i].numpy(), d["l_max"][i].numpy(), d["v"][i].numpy(), None,
                                        1e-7, 1e-7, 4000, False, return_iters=True)
        xo[i, :, 0], ito[i] = r
    if kind == "qp":
        xh, ith = ops.qp_forward(g["P"], g["q"], 1e-7, 4000, adaptive_rho=False, layout=layout, return_iters=True)
    elif kind == "qcqp":
        xh, ith = ops.qcqp_forward(g["P"], g["q"], g["l_n"], g["mu"], 1e-7, 4000, adaptive_rho=False, layout=layout,
                                   return_iters=True)
    else:
        xh, ith = ops.boxqp_forward(g["P"], g["q"], g["l_min"], g["l_max"], 1e-7, 4000, v=g.get("v"), adaptive_rho=False,
                                    layout=layout, return_iters=True)
    assert np.abs(npy(xh) - xo).max() <= 1e-6
    assert (npy(ith) == ito).mean() >= 0.9 and np.abs(npy(ith) - ito).max() <= 2


@pytest.mark.parametrize("kind", ["qp", "qcqp"])
@pytest.mark.parametrize("N,B", [(8, 64), (12, 40), (20, 24), (32, 24), (64, 24), (70, 4)])
def test_non_symmetric_p(oracle, ops, kind, N, B):
    """A non-symmetric P: the power iteration multiplies by the full P (Solver.cpp:51), llt() reads the lower triangle
    only (:76), the dual recovery and the backward system use the full P again (:127, :156), grad_P = -dl x^T is
    not symmetrised (qcqp.py:49).  Through DQQ_P_DENSE and DQQ_P_AUTO, every general forward kernel."""
    from diffqcqp_amd import _capi
    d = make_problem(kind, B, N, 4300 + N, "dense")
    gen = torch.Generator().manual_seed(4300 + N)
    U = torch.triu(torch.rand(B, N, N, generator=gen, dtype=torch.float64), diagonal=1) * 0.05
    d["P"] = (d["P"] + U).contiguous()   # upper triangle perturbed: P != P^T
    assert not torch.equal(d["P"], d["P"].transpose(1, 2))
    g = dev(d)
    xo, ito = oracle_fwd(oracle, kind, d)
    layouts = (_capi.P_DENSE, _capi.P_AUTO) if N in (8, 32, 64) else (_capi.P_DENSE,)
    for layout in layouts:
        xh, ith = hip_fwd(ops, kind, g, layout=layout)
        check_forward(xh, ith, xo, ito, min_match=0.95)
        grads, st = hip_bwd(ops, kind, g, torch.from_numpy(xo).cuda(), layout=layout)
        if kind == "qcqp":
            check_dense_backward(oracle, kind, N, {k: v.numpy() for k, v in d.items()}, xo, grads, st,
                                 oracle_bwd(oracle, kind, d, xo))
        else:
            check_backward_exact(grads, st, oracle_bwd(oracle, kind, d, xo), exact=False)
        assert not torch.equal(grads[0], grads[0].transpose(1, 2))


@pytest.mark.parametrize("kind", ["qp", "qcqp"])
@pytest.mark.parametrize("N,structure,layout", [(8, "diag", 0), (8, "dense", 1), (32, "diag", 0), (64, "dense", 1),
                                                (16, "dense", 1)])
def test_max_iter_exhaustion_returns_the_last_iterate(oracle, ops, kind, N, structure, layout):
    """max_iter reached (Solver.cpp:79 loop bound, :122 return): the last l_2 is returned, no failure is signalled."""
    d = make_problem(kind, 50, N, 4400 + N, structure)
    g = dev(d)
    for cap in (1, 3, 7):
        xo, ito = oracle_fwd(oracle, kind, d, max_iter=cap)
        xh, ith = hip_fwd(ops, kind, g, layout=layout, max_iter=cap)
        assert int(npy(ith).max()) <= cap and np.array_equal(npy(ith), ito)
        assert np.abs(npy(xh) - xo).max() <= 1e-9


def test_n_equal_one_through_the_unbatched_twin(oracle, ops):
    """N = 1 (P of shape (1,1)): the reference's unbatched QPFn2 takes its `P.size()[0] == 1` branch
    (qcqp_no_batch.py:44-45, grad_P = -(dl*l).unsqueeze(-1)), which is the 1x1 outer product."""
    from diffqcqp_amd.qcqp_no_batch import QPFn2
    for p, qv in ((0.7, -0.4), (1.3, 0.5), (2.0, -3.0)):
        P = torch.tensor([[p]], requires_grad=True)
        q = torch.tensor([[qv]], requires_grad=True)
        l = QPFn2.apply(P, q, torch.zeros(1, 1), 1e-9, 1000)
        assert l.shape == (1,)
        xo, _ = oracle.solveQP(np.array([[p]]), np.array([qv]), None, 1e-9, 1e-7, 1000, True, return_iters=True)
        assert abs(float(l.detach()) - xo[0]) <= 1e-9 and abs(float(l.detach()) - max(-qv / p, 0.0)) < 1e-5
        l.sum().backward()
        assert P.grad.shape == (1, 1) and q.grad.shape == (1, 1)
        dl = oracle.solveDerivativesQP(np.array([[p]]), np.array([qv]), xo, np.ones(1))
        assert abs(float(q.grad) + dl[0]) <= 1e-12 and abs(float(P.grad) + dl[0] * xo[0]) <= 1e-12


def _exact_qcqp_derivative(P, q, l_n, mu, x, g):
    """grad_q of the QCQP from the EXACT solve of the reference's KKT system A^T b = [0; g] (Solver.cpp:619-681
    without the Tikhonov term), numpy LU.  Returns None where A is singular to working precision."""
    n = q.size
    nc = n // 2
    r = l_n * mu
    xa, xb = x[0::2], x[1::2]
    nrm = np.sqrt(xa * xa + xb * xb)
    plq = P @ x + q
    gamma = np.zeros(nc)
    for c in range(nc):
        if not (r[c] - nrm[c] > 1e-10 or r[c] < 1e-10):
            gamma[c] = -(2 * xa[c] * plq[2 * c] + 2 * xb[c] * plq[2 * c + 1]) / (4 * nrm[c] ** 2)
    S = nrm ** 2 - r ** 2
    act = [c for c in range(nc) if S[c] > -1e-10 and r[c] > 1e-10]
    na = len(act)
    A = np.zeros((na + n, na + n))
    for k, c in enumerate(act):
        A[k, k] = S[c]
        A[k, na + 2 * c] = gamma[c] * 2 * x[2 * c]
        A[k, na + 2 * c + 1] = gamma[c] * 2 * x[2 * c + 1]
        A[na + 2 * c, k] = 2 * x[2 * c]
        A[na + 2 * c + 1, k] = 2 * x[2 * c + 1]
    D = P.copy()
    for i in range(n):
        D[i, i] += 2 * gamma[i // 2]
    A[na:, na:] = D
    rhs = np.concatenate([np.zeros(na), g])
    if np.linalg.cond(A) > 1e12:
        return None
    b = np.linalg.solve(A.T, rhs)
    return -b[na:]


def test_qcqp_refinement_exit_flips_are_the_reference_at_the_other_exit(oracle, ops):
    """End to end (x from the HIP forward, ~1e-15 from the oracle's) the refinement loop leaves after 1 body on one
    side and 3 on the other for a few percent of the problems: its exit test compares rounding noise with 1e-10
    (Solver.cpp:30-41), and the reference itself takes either exit on about half of random N=8 QCQPs.  Against the
    exact KKT derivative the two exits are NOT equally good -- 1 body is the plain Tikhonov solve (median relative
    error 3e-4 on this family), 3 bodies have refined it (4e-8) -- so a flip does change the gradient at the 1e-4
    level; that is the reference's own noise floor (a 2e-16 perturbation of x flips 4-5 % of its exits too).  What
    is asserted for exactly the flipped problems: (1) the HIP gradients equal what the reference formula gives when
    it is made to run the HIP kernel's number of bodies: the kernel returned a reference answer, the one behind
    the other exit; (2) flips go both ways and leave the share of refined (3-body) answers unchanged -- the HIP
    path is not biased towards the less accurate exit; (3) every flipped answer is at least as close to the exact
    derivative as the reference's 1-body answer for that problem."""
    B, N = 4096, 8
    d = make_problem("qcqp", B, N, 4500)
    g = dev(d)
    xh, _ = hip_fwd(ops, "qcqp", g)
    xhn = npy(xh)
    grads, st = hip_bwd(ops, "qcqp", g, xh)
    ref = oracle_bwd(oracle, "qcqp", d, xhn)           # the oracle on the SAME x: identical exits (bit-exact path)
    check_backward_exact(grads, st, ref, exact=True)
    xo, _ = oracle_fwd(oracle, "qcqp", d)
    ref_o = oracle_bwd(oracle, "qcqp", d, xo)          # the oracle on ITS x: what an end-to-end comparison sees
    sth = npy(st)
    flipped = np.nonzero(sth != ref_o[-1])[0]
    assert 0 < flipped.size <= 0.08 * B
    forced = {}
    for steps in (1, 3):
        oracle.set_force_ir_steps(steps)
        try:
            forced[steps] = oracle_bwd(oracle, "qcqp", {k: v[flipped] for k, v in d.items()}, xo[flipped])
        finally:
            oracle.set_force_ir_steps(0)
    for steps in (1, 3):                               # (1)
        sel = np.nonzero(sth[flipped] == steps)[0]
        for a, b in zip(grads, forced[steps][:-1]):
            a, b = npy(a)[flipped][sel], b[sel]
            if a.size:
                assert np.abs(a - b).max() <= 1e-6 * max(1.0, np.abs(b).max()), \
                    "HIP after %d bodies != reference formula after %d bodies" % (steps, steps)
    to3 = (sth[flipped] == 3).mean()                   # (2)
    assert 0.2 <= to3 <= 0.8, to3
    assert abs((sth == 3).mean() - (ref_o[-1] == 3).mean()) <= 0.03
    for k, i in enumerate(flipped):                    # (3)
        ex = _exact_qcqp_derivative(d["P"][i].numpy(), d["q"][i].numpy()[:, 0], d["l_n"][i].numpy()[:, 0],
                                    d["mu"][i].numpy()[:, 0], xo[i, :, 0], d["grad_x"][i].numpy()[:, 0])
        if ex is None:
            continue
        scale = max(1.0, np.abs(ex).max())
        e_hip = np.abs(npy(grads[1])[i, :, 0] - ex).max() / scale
        e_one = np.abs(forced[1][1][k, :, 0] - ex).max() / scale
        assert e_hip <= e_one * (1 + 1e-6) + 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["qp", "qcqp", "box", "sbox"])
@pytest.mark.parametrize("N", [2, 4, 6, 8, 10, 12, 16])
def test_lane_kernel_deferred_refactorisation_is_bit_identical(oracle, ops, kind, N):
    """The lane-per-problem forward (dense P, N <= 8; N = 10 .. 16: the team-per-problem forward, csrc/fwd_small.hip,
    with the same deferral) defers the refactorisation after a rho update so that one pass
    serves the lanes that fired over several trips of the wave's loop (option lane_defer, csrc/fwd_lane_dense.hip).
    A lane only sits out meanwhile: x and the iteration counts must not depend on the deferral -- every setting, at
    the reference's default budget, at budgets that run out mid-solve (max_iter exhaustion, Solver.cpp:79 / :538) and
    at eps = 1e-10; checked against the oracle once."""
    from diffqcqp_amd import _capi
    B = 3001
    d = make_problem(kind, B, N, 9100 + N, "dense")
    g = dev(d)

    def fwd(eps, max_iter):
        if kind in ("box", "sbox"):
            return ops.boxqp_forward(g["P"], g["q"], g["l_min"], g["l_max"], eps, max_iter, v=g.get("v"), layout=1,
                                     return_iters=True)
        return hip_fwd(ops, kind, g, layout=1, eps=eps, max_iter=max_iter)

    knob("fuse_fallback", 0)   # (N = 8, small B: DQQ_P_DENSE would otherwise take the group solve)
    try:
        for eps, max_iter in ((1e-7, 1000), (1e-10, 1000), (1e-7, 23), (1e-7, 7), (1e-7, 1)):
            knob("lane_defer", 1)
            x1, it1 = fwd(eps, max_iter)
            for defer in (0, 2, 3, 4, 7, 64):
                knob("lane_defer", defer)
                xd, itd = fwd(eps, max_iter)
                assert torch.equal(it1, itd), (eps, max_iter, defer)
                assert torch.equal(torch.nan_to_num(x1, nan=12345.0), torch.nan_to_num(xd, nan=12345.0)), (eps, max_iter, defer)
            if max_iter == 1000 and eps == 1e-7 and kind in ("qp", "qcqp"):
                xo, ito = oracle_fwd(oracle, kind, d)
                check_forward(xd, itd, xo, ito, min_match=0.99)
            if max_iter < 1000:
                assert int(it1.max()) <= max_iter
    finally:
        knob("lane_defer", 0)
        knob("fuse_fallback", -1)
