"""Finite differences of the HIP FORWARD against the analytic gradients of QPFn2 / QCQPFn2 / BoxQPFn2 (-m gpu).

The reference's own gradient methodology, asserted instead of printed: `test_script.py:23-43` (seed-5 problem, central
differences of P at 1e-8, eps = 1e-12, max_iter = 10000) and `Solver.cpp:830-851` (q, l_min, l_max, P of the box QP at
1e-5).  SURVEY.md 8(c)(iv).  Nothing here touches the oracle: both sides of every comparison are the product -- the
autograd Functions' backward (one HIP launch) against central differences of their forward (one HIP launch over a batch
that holds every perturbed copy of every problem).

What an honest tolerance is.  The reference does not solve the differentiated KKT system A^T b = [0; g]: it returns the
Tikhonov iterate b = (A A^T + 1e-7 I)^-1 (1e-7 b + A [0; g]) after 1 or 3 bodies (Solver.cpp:15-44).  Per singular
value s of A that is the exact answer times 1 - (1e-7 / (s^2 + 1e-7))^bodies:
  * QP, box QP: the inactive block of A is P_II, s >= lambda_min(P) >= 0.1 on these families => relative error <= 1e-5;
    the comparison with finite differences is direct.
  * QCQP: A carries the cone rows (2 gamma x_c, slack ~ 0), s_min ~ 1e-3..1e-2, and the exit after ONE body is off by
    up to a few percent.  So the check is split, as tests/test_oracle.py does for the restatement: (1) finite
    differences against an EXACT solve of the KKT matrix assembled from the kernel's own x and gamma -- that pins the
    system the backward solves, and the forward; (2) the Function's gradient against that exact solve within the bound
    the Tikhonov term implies for the problem's s_min and the kernel's own step count, (1e-7/(s_min^2+1e-7))^bodies.
Problems where a perturbation of 1e-6 could change the active set (a coordinate or a multiplier within 1e-3 -- box QP:
1e-4 -- of zero, a contact within 1e-3 of its cone) are not differentiable in the sense finite differences measure and are left out; at
least 60 % of the seeded problems must remain.
"""
import numpy as np
import pytest
import torch

from conftest import make_problem

pytestmark = pytest.mark.gpu

EPS, MAX_ITER, H = 1e-12, 100000, 1e-6
MU_IR = 1e-7      # Solver.cpp:15 `mu_ir` as every call site passes it


def fns():
    from diffqcqp_amd.qcqp import BoxQPFn2, QCQPFn2, QPFn2
    return {"qp": QPFn2, "qcqp": QCQPFn2, "box": BoxQPFn2}


def device():
    return torch.device("cuda", 0)


def solve(kind, t):
    """One forward launch over the batch in t (no autograd)."""
    F = fns()[kind]
    ws = torch.zeros_like(t["q"])
    with torch.no_grad():
        if kind == "qp":
            return F.apply(t["P"], t["q"], ws, EPS, MAX_ITER)
        if kind == "qcqp":
            return F.apply(t["P"], t["q"], t["l_n"], t["mu"], ws, EPS, MAX_ITER)
        return F.apply(t["P"], t["q"], t["l_min"], t["l_max"], ws, EPS, MAX_ITER)


def analytic(kind, t, g):
    """Gradients of sum(g * x) w.r.t. every input, through the Function's backward."""
    F = fns()[kind]
    leaves = {k: v.clone().requires_grad_(True) for k, v in t.items()}
    ws = torch.zeros_like(t["q"])
    if kind == "qp":
        x = F.apply(leaves["P"], leaves["q"], ws, EPS, MAX_ITER)
    elif kind == "qcqp":
        x = F.apply(leaves["P"], leaves["q"], leaves["l_n"], leaves["mu"], ws, EPS, MAX_ITER)
    else:
        x = F.apply(leaves["P"], leaves["q"], leaves["l_min"], leaves["l_max"], ws, EPS, MAX_ITER)
    (x * g).sum().backward()
    return x.detach(), {k: v.grad for k, v in leaves.items()}


def entries_for(kind, N):
    """Every scalar direction that is differentiated: (input name, [index tuples that move together]).  P is perturbed
    SYMMETRICALLY (the factorisation reads its lower triangle only, Solver.cpp:76; test_script.py's one-sided P[i,j]
    perturbation of the upper triangle would measure nothing of the solve), so the analytic value of a pair is
    grad_P[i,j] + grad_P[j,i]."""
    e = [("q", [(i, 0)]) for i in range(N)]
    e += [("P", [(i, j)] if i == j else [(i, j), (j, i)]) for i in range(N) for j in range(i + 1)]
    if kind == "qcqp":
        e += [("l_n", [(c, 0)]) for c in range(N // 2)] + [("mu", [(c, 0)]) for c in range(N // 2)]
    if kind == "box":
        e += [("l_min", [(i, 0)]) for i in range(N)] + [("l_max", [(i, 0)]) for i in range(N)]
    return e


def central_differences(kind, t, g, entries, h=H):
    """(nb, len(entries)): d sum(g x) / d entry, all problems and all directions in ONE forward launch."""
    nb, m = g.shape[0], len(entries)
    rep = {k: v.repeat_interleave(2 * m, dim=0).clone() for k, v in t.items()}
    base = torch.arange(nb, device=g.device) * (2 * m)
    for e, (name, idx) in enumerate(entries):
        for s, sign in enumerate((1.0, -1.0)):
            rows = base + 2 * e + s
            for ix in idx:
                rep[name][(rows,) + tuple(ix)] += sign * h
    x = solve(kind, rep)
    val = (x * g.repeat_interleave(2 * m, dim=0)).sum(dim=(1, 2)).view(nb, m, 2)
    return (val[:, :, 0] - val[:, :, 1]) / (2 * h)


def gather_analytic(grads, entries):
    cols = []
    for name, idx in entries:
        cols.append(sum(grads[name][(slice(None),) + tuple(ix)] for ix in idx))
    return torch.stack(cols, dim=1)


def to_dev(d, keys):
    return {k: d[k].to(device()).contiguous() for k in keys}


# ------------------------------------------------------------------------------------------------------------------
def test_seed5_problem_of_the_reference_script():
    """test_script.py:23-43 verbatim (n = 2, torch.manual_seed(5), P = S S^T, q = -rand - 0.1, eps 1e-12, max_iter 1e4,
    d x[0,1] / dP by central differences at 1e-8).  The script only PRINTS its analytic and numeric gradients and nothing under
    /root/reference holds their values: the constant below, grad_P[1,1] = -16.0827925 (central differences: -16.08282), is what
    SURVEY.md section 4 / Appendix C's throwaway numpy restatement of Solver.cpp computed for these inputs -- a second reading of
    the source, not an output of the reference binary.  What this test pins by itself is the loop below: the HIP forward's
    own central differences against the Function's gradients."""
    QPFn2 = fns()["qp"]
    torch.manual_seed(5)
    n = 2
    S = torch.rand(1, n, n, dtype=torch.float64) + 0.01
    P = torch.bmm(S, S.transpose(1, 2)).to(device()).requires_grad_(True)
    q = (-torch.rand((1, n, 1), dtype=torch.float64) - 0.1).to(device()).requires_grad_(True)
    ws = torch.zeros_like(q)
    lf = QPFn2.apply(P, q, ws, 1e-12, 10000)
    lf[0, 1].backward()
    gP, gq = P.grad.clone(), q.grad.clone()
    assert abs(gP[0, 1, 1].item() - (-16.0827925)) < 2e-3      # (SURVEY.md App. C's numpy restatement; NOT a reference output)
    with torch.no_grad():
        Pd, qd = P.detach(), q.detach()
        for i in range(n):
            for j in range(i + 1):
                d = torch.zeros_like(Pd)
                d[0, i, j] = d[0, j, i] = 1e-8
                num = (QPFn2.apply(Pd + d, qd, ws, 1e-12, 10000)[0, 1] - QPFn2.apply(Pd - d, qd, ws, 1e-12, 10000)[0, 1]).item() / 2e-8
                ana = gP[0, i, j].item() if i == j else (gP[0, i, j] + gP[0, j, i]).item()
                assert abs(num - ana) < 1e-4 * max(1.0, abs(ana)), ("P", i, j, num, ana)
        for i in range(n):
            d = torch.zeros_like(qd)
            d[0, i, 0] = 1e-6
            num = (QPFn2.apply(Pd, qd + d, ws, 1e-12, 10000)[0, 1] - QPFn2.apply(Pd, qd - d, ws, 1e-12, 10000)[0, 1]).item() / 2e-6
            assert abs(num - gq[0, i, 0].item()) < 1e-4 * max(1.0, abs(gq[0, i, 0].item())), ("q", i, num)


@pytest.mark.parametrize("structure,seed", [("dense", 9101), ("diag", 9102)])
def test_qp_gradients_match_central_differences(structure, seed):
    """32 seeded N = 8 QPs: d sum(g x) / d(q, P) of QPFn2 against central differences of its forward (delta 1e-6,
    eps 1e-12).  Tolerance: 2e-4 absolute + 1e-3 relative (the Tikhonov term contributes <= 1e-5 relative; the rest is
    the forward's own accuracy, ~1e-10, divided by the step)."""
    nb, N = 32, 8
    d = make_problem("qp", nb, N, seed, structure)
    t, g = to_dev(d, ("P", "q")), d["grad_x"].to(device())
    x, grads = analytic("qp", t, g)
    # strict complementarity margin: every coordinate either clearly positive or with a clearly positive multiplier
    gamma = -(torch.bmm(t["P"], x) + t["q"])          # dualFromPrimalQP, Solver.cpp:125-134 (sign as the reference's)
    margin = torch.maximum(x, -gamma).abs().amin(dim=(1, 2))
    keep = (margin > 1e-3).cpu().numpy()
    assert keep.mean() >= 0.6, keep.mean()
    entries = entries_for("qp", N)
    fd = central_differences("qp", t, g, entries).cpu().numpy()[keep]
    an = gather_analytic(grads, entries).cpu().numpy()[keep]
    err = np.abs(fd - an) - (2e-4 + 1e-3 * np.abs(an))
    assert err.max() <= 0, (float(np.abs(fd - an).max()), np.unravel_index(err.argmax(), err.shape))
    assert np.abs(an).max() > 0.1      # (not a comparison of zeros)


def _qcqp_kkt_exact(P, q, l_n, mu, x, gamma, g):
    """Exact solve of the differentiated KKT system of Solver.cpp:619-681 for ONE problem (numpy): returns
    (dl (N), dgamma (nc), s_min of A, slack margin)."""
    N, nc = x.size, x.size // 2
    r = l_n * mu
    S = np.array([x[2 * c] ** 2 + x[2 * c + 1] ** 2 - r[c] ** 2 for c in range(nc)])
    act = [c for c in range(nc) if S[c] > -1e-10 and r[c] > 1e-10]           # Solver.cpp:639
    na = len(act)
    A = np.zeros((N + na, N + na))
    for k, c in enumerate(act):
        A[k, k] = S[c]
        A[k, na + 2 * c: na + 2 * c + 2] = gamma[c] * 2 * x[2 * c: 2 * c + 2]
        A[na + 2 * c: na + 2 * c + 2, k] = 2 * x[2 * c: 2 * c + 2]
    A[na:, na:] = P + np.diag(2 * np.repeat(gamma, 2))
    b = np.linalg.solve(A.T, np.concatenate([np.zeros(na), g]))
    dgam = np.zeros(nc)
    dgam[act] = b[:na]
    smin = np.linalg.svd(A, compute_uv=False).min()
    return b[na:], dgam, smin, act


def test_qcqp_gradients_match_central_differences():
    """32 seeded dense N = 8 QCQPs (a Delassus-like P = S S^T/8 + 0.1 I): see the module docstring for the split."""
    from diffqcqp_amd import ops
    nb, N, nc = 32, 8, 4
    d = make_problem("qcqp", nb, N, 9103, "dense")
    d["l_n"] = d["l_n"] * 0.5 + 0.05
    d["mu"] = d["mu"] * 0.5 + 0.25           # radii in [0.0125, 0.41]: most contacts slide, none is degenerate
    t, g = to_dev(d, ("P", "q", "l_n", "mu")), d["grad_x"].to(device())
    x, grads = analytic("qcqp", t, g)
    gam, dgam = torch.empty_like(t["l_n"]), torch.empty_like(t["l_n"])
    steps = ops.qcqp_backward(t["P"], t["q"], t["l_n"], t["mu"], x, g, return_steps=True, duals=(gam, dgam))[-1].cpu().numpy()
    entries = entries_for("qcqp", N)
    fd = central_differences("qcqp", t, g, entries).cpu().numpy()
    an = gather_analytic(grads, entries).cpu().numpy()
    h = {k: v.cpu().numpy() for k, v in t.items()}
    xs, gs, gams = x.cpu().numpy()[:, :, 0], g.cpu().numpy()[:, :, 0], gam.cpu().numpy()[:, :, 0]
    kept = 0
    worst_fd, worst_tik = 0.0, 0.0
    for b in range(nb):
        ln, m_ = h["l_n"][b, :, 0], h["mu"][b, :, 0]
        r = ln * m_
        nrm = np.hypot(xs[b, 0::2], xs[b, 1::2])
        # differentiable in the finite-difference sense: every contact clearly inside its cone or clearly sliding
        inside, sliding = nrm < r - 1e-3, (np.abs(nrm - r) < 1e-7) & (gams[b] > 1e-3)
        if not np.all(inside | sliding):
            continue
        kept += 1
        dl, dgm, smin, act = _qcqp_kkt_exact(h["P"][b], h["q"][b, :, 0], ln, m_, xs[b], gams[b], gs[b])
        # exact gradients in the layout of `entries`: grad_q = -dl, grad_P = -dl x^T, grad_l_n = E2 dgamma, grad_mu = E1 dgamma
        gP = -np.outer(dl, xs[b])
        ex = [-dl[i] for i in range(N)]
        ex += [gP[i, j] if i == j else gP[i, j] + gP[j, i] for i in range(N) for j in range(i + 1)]
        ex += [2 * gams[b, c] * ln[c] * m_[c] ** 2 * dgm[c] for c in range(nc)]       # E2, Solver.cpp:683-691
        ex += [2 * gams[b, c] * ln[c] ** 2 * m_[c] * dgm[c] for c in range(nc)]       # E1
        ex = np.array(ex)
        # (1) the system (and the forward): finite differences against the exact solve
        e1 = np.abs(fd[b] - ex) - (5e-5 + 2e-3 * np.abs(ex))
        assert e1.max() <= 0, ("FD vs exact KKT solve", b, float(np.abs(fd[b] - ex).max()))
        worst_fd = max(worst_fd, float(np.abs(fd[b] - ex).max()))
        # (2) the Function's gradient is that solve up to the Tikhonov term, at the kernel's own step count
        damp = (MU_IR / (smin ** 2 + MU_IR)) ** int(steps[b])
        scale = max(1.0, float(np.abs(ex).max()))
        bound = 2.0 * damp * scale * np.sqrt(N + len(act)) + 1e-6 * scale
        e2 = float(np.abs(an[b] - ex).max())
        assert e2 <= bound, ("analytic vs exact beyond the Tikhonov bound", b, e2, bound, smin, int(steps[b]))
        worst_tik = max(worst_tik, e2 / scale)
    assert kept >= 0.6 * nb, kept
    assert set(np.unique(steps)) <= {1, 3}, np.unique(steps)     # the two exits the reference's loop takes


def test_box_qp_gradients_match_central_differences():
    """The reference's box-QP driver (Solver.cpp:802-853: G = R R^T, gradients w.r.t. q, l_min, l_max, P by one-sided
    differences of 1e-5) on 32 seeded dense N = 8 problems, central differences at 1e-6, through BoxQPFn2 -- whose
    backward does not run in the reference's Python; signs as finite differences say
    (grad_l_max = +dgamma_hi gamma_hi).  The reference's system A = [[0, diag(gamma_act) E^T], [E, P]] (Solver.cpp:341-350)
    has singular values of the order of the active multipliers, so its Tikhonov answer is off by 1e-7 / gamma^2 -- "a
    percent or so" near weakly active bounds; the check is split like the QCQP's: finite differences against the EXACT
    derivative (closed form: dl_F = P_FF^-1 g_F, d/dbound_i = g_i - P_Fi . dl_F), then the Function against that within
    the Tikhonov bound at the kernel's own step count."""
    from diffqcqp_amd import ops
    nb, N = 32, 8
    d = make_problem("box", nb, N, 9104, "dense")
    d["l_min"], d["l_max"] = 2.5 * d["l_min"], 2.5 * d["l_max"]     # (about half of the coordinates end up between their bounds)
    t, g = to_dev(d, ("P", "q", "l_min", "l_max")), d["grad_x"].to(device())
    x, grads = analytic("box", t, g)
    steps = ops.boxqp_backward(t["P"], t["q"], t["l_min"], t["l_max"], x, g, return_steps=True)[-1].cpu().numpy()
    entries = entries_for("box", N)
    fd = central_differences("box", t, g, entries).cpu().numpy()
    an = gather_analytic(grads, entries).cpu().numpy()
    h = {k: v.cpu().numpy() for k, v in t.items()}
    xs, gs = x.cpu().numpy()[:, :, 0], g.cpu().numpy()[:, :, 0]
    kept, bound_grads = 0, 0.0
    for b in range(nb):
        P, lo, hi = h["P"][b], h["l_min"][b, :, 0], h["l_max"][b, :, 0]
        r = P @ xs[b] + h["q"][b, :, 0]                     # = gamma_lo - gamma_hi at the solution
        at_lo, at_hi = np.abs(xs[b] - lo) < 1e-9, np.abs(xs[b] - hi) < 1e-9
        free = ~(at_lo | at_hi)
        ok = np.where(free, np.minimum(xs[b] - lo, hi - xs[b]) > 1e-4, np.abs(r) > 1e-4)
        if not ok.all():
            continue
        kept += 1
        F = np.where(free)[0]
        dl = np.zeros(N)
        if F.size:
            dl[F] = np.linalg.solve(P[np.ix_(F, F)].T, gs[b, F])
        gP = -np.outer(dl, xs[b])
        gb = gs[b] - P[F, :].T @ dl[F]                      # d sum(g x) / d (the bound coordinate i sits on)
        ex = [-dl[i] for i in range(N)]
        ex += [gP[i, j] if i == j else gP[i, j] + gP[j, i] for i in range(N) for j in range(i + 1)]
        ex += [gb[i] if at_lo[i] else 0.0 for i in range(N)] + [gb[i] if at_hi[i] else 0.0 for i in range(N)]
        ex = np.array(ex)
        e1 = np.abs(fd[b] - ex) - (5e-5 + 1e-3 * np.abs(ex))
        assert e1.max() <= 0, ("FD vs exact derivative", b, float(np.abs(fd[b] - ex).max()))
        # the reference's system for this active set and its smallest singular value
        act = [i for i in range(N) if at_lo[i]] + [i for i in range(N) if at_hi[i]]
        na = len(act)
        A = np.zeros((na + N, na + N))
        for k, i in enumerate(act):
            A[k, na + i] = abs(r[i])
            A[na + i, k] = 1.0
        A[na:, na:] = P
        smin = np.linalg.svd(A, compute_uv=False).min()
        damp = (MU_IR / (smin ** 2 + MU_IR)) ** int(steps[b, 1])
        scale = max(1.0, float(np.abs(ex).max()))
        bound = 2.0 * damp * scale * np.sqrt(N + na) + 1e-6 * scale
        e2 = float(np.abs(an[b] - ex).max())
        assert e2 <= bound, ("analytic vs exact beyond the Tikhonov bound", b, e2, bound, smin, steps[b].tolist())
        bound_grads = max(bound_grads, float(np.abs(ex[-2 * N:]).max()))
    assert kept >= 0.6 * nb, kept
    assert bound_grads > 0.05      # some bounds are active on these problems: the bound gradients are not all zero
