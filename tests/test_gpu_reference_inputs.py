"""-m gpu: the reference's OWN hard-coded matrices and singular / rank-deficient dense P through the HIP path.

Inputs: tests/golden/reference_inputs.py (the literals of Solver::test(), Solver.cpp:697-923: a singular 4x4 P with
a 1.6e7-scale solution, the 12x12 G2 G2^T with cond ~1e22, the 4x4 block-diagonal G with a zero-radius contact, the
rank-3 8x8 Delassus matrix G4) and seeded rank-deficient batches (P = S S^T with S of rank N/2; duplicated Jacobian
rows like G4) at N = 8, 32, 64.  Expected values: the oracle's (fixtures ref_*.npz / rd_*.npz from make_golden.py,
and the oracle run on the spot for the large batches).  Every case goes through DQQ_P_AUTO (what QPFn2 / QCQPFn2 pass)
and DQQ_P_DENSE.

Tolerances (float64), stated relative to the problem's own scale s = max(1, max|value|) because these solutions
reach 1.6e7:
  forward   |x - x_oracle| <= 1e-6 s and IDENTICAL ADMM iteration counts (the trajectory is the reference's);
  backward  on the oracle's x: gradients within 1e-6 s of the reference formula.  On these systems the refinement loop
            (Solver.cpp:32-41) runs 1, 3, 4, 5 ... bodies depending on a residual that is rounding noise -- K = A^T A +
            1e-7 I has cond ~1e9 and beyond, and the oracle's own exit flips under a differently ordered evaluation
            (oracle/README.md).  A kernel that evaluates the sums in the reference's order (N <= 16)
            must reproduce the step counts exactly; a re-associating kernel (matrix cores: 16 < N <= 64) may leave
            the loop at another body, and is then compared with the reference formula run for ITS number of bodies
            (orc_set_force_ir_steps) -- every problem is checked, none is excluded.
"""
import glob
import os
import sys

import numpy as np
import pytest

from conftest import knob
import torch

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLDEN)
import reference_inputs as R  # noqa: E402

TOL = 1e-6
# the QCQP's contact gradients (grad_l_n, grad_mu) through a re-associating kernel: the evaluation-order noise of the
# reference's own formulas on these systems is up to 8.6e-6 (tests/test_gpu_parity.py: REASSOC_TOL, oracle/README.md)
TOL_CONTACT_REASSOC = 2e-5


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "these tests need the GPU"
    from diffqcqp_amd import build, ops as _ops, _capi
    build.build()
    _capi.lib()
    return _ops


def _scale(b):
    return np.maximum(1.0, np.abs(b).reshape(b.shape[0], -1).max(1)).reshape((-1,) + (1,) * (b.ndim - 1))


def _rel(a, b):
    return float((np.abs(a - b) / _scale(b)).max()) if b.size else 0.0


def _oracle_bwd(O, kind, d, x, nthreads=8):
    if kind == "qp":
        return O.qp_bwd_batch(d["P"], d["q"], x, d["grad_x"], nthreads=nthreads)
    return O.qcqp_bwd_batch(d["P"], d["q"], d["l_n"], d["mu"], x, d["grad_x"], nthreads=nthreads)


def _hip(ops, kind, t, eps, max_iter, layout, x_for_bwd):
    if kind == "qp":
        xh, ith = ops.qp_forward(t["P"], t["q"], eps, max_iter, layout=layout, return_iters=True)
        *gh, sh = ops.qp_backward(t["P"], t["q"], x_for_bwd, t["grad_x"], layout=layout, return_steps=True)
    else:
        xh, ith = ops.qcqp_forward(t["P"], t["q"], t["l_n"], t["mu"], eps, max_iter, layout=layout, return_iters=True)
        *gh, sh = ops.qcqp_backward(t["P"], t["q"], t["l_n"], t["mu"], x_for_bwd, t["grad_x"], layout=layout,
                                    return_steps=True)
    torch.cuda.synchronize()
    return xh.cpu().numpy(), ith.cpu().numpy(), [g.cpu().numpy() for g in gh], sh.cpu().numpy()


def check_case(O, ops, kind, d, xo, ito, ref, eps, max_iter, exact_order, max_flip=1.0, label=""):
    """d: numpy inputs; xo/ito/ref: the oracle's forward and backward (on xo).  exact_order: the route evaluates the
    backward in the reference's summation order => step counts must be identical."""
    *gref, sref = ref
    t = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in d.items()}
    xs = torch.from_numpy(np.ascontiguousarray(xo)).cuda()
    for layout in (0, 1):
        tag = "%s layout=%d" % (label, layout)
        xh, ith, gh, sh = _hip(ops, kind, t, eps, max_iter, layout, xs)
        assert np.isfinite(xh).all(), tag
        assert np.array_equal(ith, ito), "%s: ADMM iteration counts differ on %d problems" % (tag, (ith != ito).sum())
        assert _rel(xh, xo) <= TOL, "%s: x off by %.2e (relative to the solution scale)" % (tag, _rel(xh, xo))
        for g in gh:
            assert np.isfinite(g).all(), tag + ": non-finite gradient"
        same = sh == sref
        if exact_order:
            assert same.all(), "%s: refinement step counts differ on %d problems" % (tag, (~same).sum())
        else:
            assert 1.0 - same.mean() <= max_flip, "%s: refinement exit differs on %.1f%%" % (tag, 100 * (1 - same.mean()))
        tols = [TOL, TOL] + [TOL if exact_order else TOL_CONTACT_REASSOC] * 2
        for a, b, tol in zip(gh, gref, tols):
            assert _rel(a[same], b[same]) <= tol, "%s: gradient off by %.2e (same exit)" % (tag, _rel(a[same], b[same]))
        for steps in np.unique(sh[~same]):            # the reference formula at the exit the kernel took
            sel = np.nonzero((~same) & (sh == steps))[0]
            O.set_force_ir_steps(int(steps))
            try:
                forced = _oracle_bwd(O, kind, {k: v[sel] for k, v in d.items()}, xo[sel])
            finally:
                O.set_force_ir_steps(0)
            for a, b, tol in zip(gh, forced[:-1], tols):
                assert _rel(a[sel], b) <= tol, "%s: gradient off by %.2e at the kernel's own exit (%d bodies)" % (
                    tag, _rel(a[sel], b), steps)


def _reference_order(kind, N):
    # QP and QCQP backward for 16 < N <= 64 run on the matrix cores (re-associated sums); everything else evaluates the
    # refinement in the reference's order
    return not (16 < N <= 64)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "ref_*.npz")) +
                                        glob.glob(os.path.join(GOLDEN, "rd_*.npz"))), ids=os.path.basename)
def test_hip_reproduces_reference_inputs(oracle, ops, path):
    z = np.load(path)
    kind = "qcqp" if "l_n" in z.files else "qp"
    d = {k: z[k] for k in ("P", "q", "grad_x", "l_n", "mu") if k in z.files}
    ref = [z[k] for k in (("grad_P", "grad_q", "grad_l_n", "grad_mu") if kind == "qcqp" else ("grad_P", "grad_q"))]
    N = d["q"].shape[1]
    check_case(oracle, ops, kind, d, z["x"], z["iters"], ref + [z["ir_steps"]], float(z["eps"]), int(z["max_iter"]),
               _reference_order(kind, N), label=os.path.basename(path))


@pytest.mark.parametrize("B", [1, 33, 4096])
def test_reference_matrices_as_batches(oracle, ops, B):
    """The same literals replicated into batches (B = 1: one problem per launch; 33: a ragged tile; 4096: the
    multi-wave routes), each problem with its own seeded grad_x -- the routing depends on B, the answers must not."""
    rng = np.random.default_rng(B)

    def rep(P, q, rad=None):
        n = q.size
        d = {"P": np.broadcast_to(P, (B, n, n)).copy(), "q": np.broadcast_to(q.reshape(n, 1), (B, n, 1)).copy(),
             "grad_x": rng.standard_normal((B, n, 1))}
        if rad is not None:
            d["l_n"] = np.broadcast_to(rad.reshape(-1, 1), (B, n // 2, 1)).copy()
            d["mu"] = np.ones((B, n // 2, 1))
        return d

    Pm, qm, lm = R.m2_singular()
    P2, q2 = R.g2_product()
    Pb, qb, radii = R.g_blockdiag()
    P4, q4, l4 = R.g4_delassus()
    cases = [("m2 qp", "qp", rep(Pm, qm), 1000), ("m2 qp max_iter=1", "qp", rep(Pm, qm), 1),
             ("m2 qcqp", "qcqp", rep(Pm, qm, lm), 1000), ("g2 qp", "qp", rep(P2, q2), 1000),
             ("g2 qcqp", "qcqp", rep(P2, q2, np.full(6, 0.1)), 1000), ("gblock qcqp r=(.00966, 0)", "qcqp", rep(Pb, qb, radii[0]), 1000),
             ("g4 qp", "qp", rep(P4, -q4), 1000), ("g4 qcqp", "qcqp", rep(P4, q4, l4), 1000)]
    for label, kind, d, mi in cases:
        if kind == "qp":
            xo, ito = oracle.qp_fwd_batch(d["P"], d["q"], 1e-10, mi, nthreads=8)
        else:
            xo, ito = oracle.qcqp_fwd_batch(d["P"], d["q"], d["l_n"], d["mu"], 1e-10, mi, nthreads=8)
        check_case(oracle, ops, kind, d, xo, ito, _oracle_bwd(oracle, kind, d, xo), 1e-10, mi, True,
                   label="%s B=%d" % (label, B))


@pytest.mark.parametrize("kind", ["qp", "qcqp"])
@pytest.mark.parametrize("N", [8, 32, 64])
@pytest.mark.parametrize("family", ["lowrank", "duprows", "psd_eps"])
def test_rank_deficient_batches(oracle, ops, family, N, kind):
    """B >= 4096 (2048 at N = 64) problems with a singular (or 1e-9-away-from-singular) dense P: forward trajectories
    identical to the oracle's -- a third of the QPs run into max_iter = 1000 --, backward within 1e-6 at the exit the
    kernel took."""
    B = 4096 if N < 64 else 2048
    d = {k: v.numpy() for k, v in R.rank_deficient(kind, B, N, 7000 + N, family).items()}
    nt = min(32, os.cpu_count() or 1)
    if kind == "qp":
        xo, ito = oracle.qp_fwd_batch(d["P"], d["q"], 1e-7, 1000, nthreads=nt)
    else:
        xo, ito = oracle.qcqp_fwd_batch(d["P"], d["q"], d["l_n"], d["mu"], 1e-7, 1000, nthreads=nt)
    ref = _oracle_bwd(oracle, kind, d, xo, nthreads=nt)
    check_case(oracle, ops, kind, d, xo, ito, ref, 1e-7, 1000, _reference_order(kind, N), max_flip=0.4,
               label="%s %s N=%d" % (family, kind, N))


@pytest.mark.parametrize("kind", ["qp", "qcqp"])
@pytest.mark.parametrize("family", ["lowrank", "duprows", "psd_eps"])
def test_rank_deficient_batches_on_one_two_and_four_lanes_per_problem(ops, family, kind):
    """The forward picks its lane layout at N = 8 by batch size and by a hint (DQQ_F_EXPECT_DENSE); x and the iteration counts
    must be the same bits on every layout -- also on singular P, where a third of the QPs run into max_iter and some
    factorisations fail (NaN outputs: compared as bit patterns).  A third of the batch is made diagonal: both branches of
    the fused kernel run in every layout."""
    from diffqcqp_amd import _capi
    B, N = 3000, 8
    t = R.rank_deficient(kind, B, N, 7100, family)
    P = t["P"].clone()
    idx = torch.arange(64 * 7, 64 * 7 + 960)
    P[idx] = torch.diag_embed(torch.diagonal(P[idx], dim1=1, dim2=2).abs() + 0.05)
    g = {k: v.cuda() for k, v in t.items()}
    g["P"] = P.cuda()
    out = {}
    try:
        for lpp in (1, 2, 4):
            knob("fwd_lpp", lpp)
            if kind == "qp":
                out[lpp] = ops.qp_forward(g["P"], g["q"], 1e-7, 1000, return_iters=True)
            else:
                out[lpp] = ops.qcqp_forward(g["P"], g["q"], g["l_n"], g["mu"], 1e-7, 1000, return_iters=True)
    finally:
        knob("fwd_lpp", 0)
    for lpp in (1, 4):
        assert torch.equal(out[lpp][0].view(torch.int64), out[2][0].view(torch.int64)), "x differs on %d lanes" % lpp
        assert torch.equal(out[lpp][1], out[2][1]), "iteration counts differ on %d lanes" % lpp


@pytest.mark.parametrize("kind", ["qp", "qcqp"])
@pytest.mark.parametrize("family", ["lowrank", "duprows"])
def test_rank_deficient_backward_lane_kernel_is_the_team_kernel_bit_for_bit(ops, family, kind):
    """The lane-per-problem backward (declared dense, and the routes the feedback word opens through DQQ_P_AUTO) against the
    team kernel on singular P at a batch size that selects it: K = A A^T + 1e-7 I with cond ~1e9, refinement loops that leave
    at 1 ... 5 bodies -- every output and every step count the same bits (NaNs compared as bit patterns)."""
    from diffqcqp_amd import _capi
    B, N = 24576 + 192, 8
    t = R.rank_deficient(kind, B, N, 7200, family)
    g = {k: v.cuda() for k, v in t.items()}
    g["grad_x"] = torch.randn(B, N, 1, dtype=torch.float64, generator=torch.Generator().manual_seed(7201)).cuda()
    if kind == "qp":
        x = ops.qp_forward(g["P"], g["q"], 1e-7, 1000, layout=1)
        run = lambda: ops.qp_backward(g["P"], g["q"], x, g["grad_x"], layout=1, return_steps=True)
    else:
        x = ops.qcqp_forward(g["P"], g["q"], g["l_n"], g["mu"], 1e-7, 1000, layout=1)
        run = lambda: ops.qcqp_backward(g["P"], g["q"], g["l_n"], g["mu"], x, g["grad_x"], layout=1, return_steps=True)
    out = {}
    try:
        for opt in (1, 0):
            knob("lane_bwd", opt)
            out[opt] = run()
    finally:
        knob("lane_bwd", 1)
    for a, b in zip(out[1], out[0]):
        a64 = a.view(torch.int64) if a.dtype is torch.float64 else a
        b64 = b.view(torch.int64) if b.dtype is torch.float64 else b
        assert torch.equal(a64, b64)
