"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through
the C ABI (diffqcqp_amd.ops -> libdiffqcqp_hip.so), against the CPU oracle on the
same seeded inputs, against the committed golden fixtures, and -- at the full
BASELINE.json batch sizes -- through size-independent properties.

Tolerances (float64):
  x          |x_hip - x_oracle| <= 1e-6 everywhere (the north-star tolerance); in practice the
             trajectories coincide (same iteration count) and the difference is ~1e-13, which the
             tests also assert through the median and the iteration-count match rate;
  gradients  with IDENTICAL x fed to both sides the backward is required to be bit-exact on the
             diagonal fast path and within 1e-9 relative on the dense path.  End to end (x from the
             HIP forward, ~1e-15 away from the oracle's x) the QP gradients must agree to 1e-6 of the
             gradient scale.  For the QCQP the reference's refinement loop (Solver.cpp:32-41) exits
             after 1 or 3 Tikhonov steps depending on whether a residual that is pure rounding noise
             is below 1e-10; a 2e-16 relative perturbation of x flips that decision on ~3-4% of the
             problems IN THE ORACLE ITSELF and changes those gradients by up to tens of percent
             (the two exits differ by mu*K^-1 x).  So end to end the QCQP gradients are compared on
             the problems whose step count agrees (1e-6) and the flip rate is bounded (<= 8%).
"""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import KNOB_DEFAULTS, knob, make_problem

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
# ref_*.npz / rd_*.npz (the reference's own matrices, rank-deficient P) have their own tests with scale-relative tolerances
GENERIC_FIXTURES = sorted(p for p in glob.glob(os.path.join(GOLDEN, "*.npz"))
                          if not os.path.basename(p).startswith(("ref_", "rd_")))
X_TOL = 1e-6


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "these tests need the GPU"
    from diffqcqp_amd import build, ops as _ops, _capi
    build.build()
    _capi.lib()
    from conftest import OpsWithFlags
    yield OpsWithFlags(_ops)
    for name, value in list(KNOB_DEFAULTS.items()) + [("dense_wave64", 1), ("wave_qcqp_bwd", 1)]:
        knob(name, value)


def dev(d):
    return {k: v.cuda() for k, v in d.items()}


def npy(t):
    return None if t is None else t.detach().cpu().numpy()


def oracle_fwd(O, kind, d, eps=1e-7, max_iter=1000):
    if kind == "qp":
        return O.qp_fwd_batch(d["P"].numpy(), d["q"].numpy(), eps, max_iter, nthreads=8)
    return O.qcqp_fwd_batch(d["P"].numpy(), d["q"].numpy(), d["l_n"].numpy(), d["mu"].numpy(), eps, max_iter, nthreads=8)


def oracle_bwd(O, kind, d, x):
    if kind == "qp":
        return O.qp_bwd_batch(d["P"].numpy(), d["q"].numpy(), x, d["grad_x"].numpy(), nthreads=8)
    return O.qcqp_bwd_batch(d["P"].numpy(), d["q"].numpy(), d["l_n"].numpy(), d["mu"].numpy(), x, d["grad_x"].numpy(), nthreads=8)


def hip_fwd(ops, kind, g, layout=0, eps=1e-7, max_iter=1000):
    if kind == "qp":
        return ops.qp_forward(g["P"], g["q"], eps, max_iter, layout=layout, return_iters=True)
    return ops.qcqp_forward(g["P"], g["q"], g["l_n"], g["mu"], eps, max_iter, layout=layout, return_iters=True)


def hip_bwd(ops, kind, g, x, layout=0):
    if kind == "qp":
        gP, gq, st = ops.qp_backward(g["P"], g["q"], x, g["grad_x"], layout=layout, return_steps=True)
        return [gP, gq], st
    gP, gq, gl, gm, st = ops.qcqp_backward(g["P"], g["q"], g["l_n"], g["mu"], x, g["grad_x"], layout=layout,
                                           return_steps=True)
    return [gP, gq, gl, gm], st


def check_forward(xh, ith, xo, ito, min_match=0.999):
    diff = np.abs(npy(xh) - xo)
    assert np.isfinite(npy(xh)).all()
    assert diff.max() <= X_TOL, "max |x_hip - x_oracle| = %g" % diff.max()
    match = (npy(ith) == ito).mean()
    assert match >= min_match, "iteration counts differ on %.3f%% of the problems" % (100 * (1 - match))
    assert np.median(diff.max(axis=(1, 2))) < 1e-11


def check_end_to_end(grads, steps, ref, max_flip=0.08, tol=1e-6):
    """x came from the HIP forward: compare where the refinement exit agrees (see module docstring)."""
    *gref, sref = ref
    same = npy(steps) == sref
    assert 1.0 - same.mean() <= max_flip, "refinement exit differs on %.1f%% of the problems" % (100 * (1 - same.mean()))
    for a, b in zip(grads, gref):
        a, b = npy(a)[same], b[same]
        if a.size:
            scale = np.maximum(1.0, np.abs(b).reshape(b.shape[0], -1).max(1)).reshape((-1,) + (1,) * (b.ndim - 1))
            assert (np.abs(a - b) / scale).max() <= tol


def check_backward_exact(grads, steps, ref, exact=True, rtol=1e-9):
    *gref, sref = ref
    assert np.array_equal(npy(steps), sref), "refinement step counts differ"
    for a, b in zip(grads, gref):
        a = npy(a)
        if exact:
            assert np.array_equal(a, b), "max diff %g" % np.abs(a - b).max()
        else:
            assert np.allclose(a, b, rtol=rtol, atol=rtol * max(1.0, np.abs(b).max()))


# Evaluation-order noise of the reference's OWN backward formulas (K = A A^T + 1e-7 I, cond ~ 1e9 and beyond): an
# OpenBLAS / LAPACK-ordered evaluation (tools/independent_order_check.py) differs from the oracle, on the batches of
# test_qcqp_backward_wave_kernel_16_to_32 and where the refinement exits agree, by up to 1.1e-7 (grad_q), 3.9e-6
# (grad_l_n) and 8.6e-6 (grad_mu); QP systems: 1e-8.  The tolerances of a re-associating kernel follow from that.
REASSOC_TOL = {"qp": (1e-7, 1e-7), "qcqp": (1e-6, 1e-6, 2e-5, 2e-5)}


def check_backward_reassociated(oracle, kind, d, xo, grads, steps, ref, min_same=0.9):
    """Kernels that evaluate the backward's sums in another order than the reference (matrix cores: QP and QCQP, 16 < N <= 64): gradients within REASSOC_TOL where the refinement exit agrees; where it does not -- the exit
    test compares rounding noise with 1e-10, Solver.cpp:30-39 -- against the reference formula run for the kernel's
    own number of bodies (orc_set_force_ir_steps).  Every problem is checked."""
    tols = REASSOC_TOL[kind]
    *gref, sref = ref
    sth = npy(steps)
    same = sth == sref
    B = sth.shape[0]
    assert same.mean() >= min_same, "refinement exit differs on %.1f%%" % (100 * (1 - same.mean()))

    def rel(a, b):
        scale = np.maximum(1.0, np.abs(b).reshape(b.shape[0], -1).max(1)).reshape((-1,) + (1,) * (b.ndim - 1))
        return float((np.abs(a - b) / scale).max()) if b.size else 0.0
    for a, b, tol in zip(grads, gref, tols):
        a = npy(a)
        assert np.isfinite(a).all()
        assert rel(a[same], b[same]) <= tol, "off by %.2e (tolerance %.0e)" % (rel(a[same], b[same]), tol)
    for st in np.unique(sth[~same]):
        sel = np.nonzero((~same) & (sth == st))[0]
        oracle.set_force_ir_steps(int(st))
        try:
            forced = oracle_bwd(oracle, kind, {k: torch.as_tensor(v)[torch.as_tensor(sel)] for k, v in d.items()}, xo[sel])
        finally:
            oracle.set_force_ir_steps(0)
        for a, b, tol in zip(grads, forced[:-1], tols):
            assert rel(npy(a)[sel], b) <= 10 * tol
    assert B == same.shape[0]


def reassociating(kind, N):
    """Which default routes of the general (dense P) backward evaluate their sums on the matrix cores."""
    return kind in ("qp", "qcqp") and 16 < N <= 64


def check_dense_backward(oracle, kind, N, d, xo, grads, steps, ref, min_same=0.75):
    if reassociating(kind, N):
        check_backward_reassociated(oracle, kind, d, xo, grads, steps, ref, min_same=min_same)
    else:
        check_backward_exact(grads, steps, ref, exact=False)


# ---------------------------------------------------------------- diagonal fast path
@pytest.mark.parametrize("kind", ["qp", "qcqp"])
@pytest.mark.parametrize("N,B", [(2, 777), (4, 500), (8, 2051), (16, 301), (32, 131), (64, 37)])
def test_diag_fast_path_matches_oracle(oracle, ops, kind, N, B):
    d = make_problem(kind, B, N, 100 + N)
    g = dev(d)
    xo, ito = oracle_fwd(oracle, kind, d)
    xh, ith = hip_fwd(ops, kind, g)
    check_forward(xh, ith, xo, ito)
    # backward on IDENTICAL x: bit-exact, including the 1-vs-3 refinement step decision
    grads, st = hip_bwd(ops, kind, g, torch.from_numpy(xo).cuda())
    check_backward_exact(grads, st, oracle_bwd(oracle, kind, d, xo), exact=True)
    # end to end
    grads2, st2 = hip_bwd(ops, kind, g, xh)
    check_end_to_end(grads2, st2, oracle_bwd(oracle, kind, d, xo), max_flip=0.0 if kind == "qp" else 0.08)


@pytest.mark.parametrize("kind", ["qp", "qcqp"])
@pytest.mark.parametrize("N,lpps", [(8, (1, 2, 4)), (16, (2, 4, 8)), (32, (4, 8, 16)), (64, (8, 16, 32)), (4, (1, 2))])
def test_every_lanes_per_problem_variant(oracle, ops, kind, N, lpps):
    from diffqcqp_amd import _capi
    d = make_problem(kind, 333, N, 200 + N)
    g = dev(d)
    xo, ito = oracle_fwd(oracle, kind, d)
    for wpb in (1, 4):
        for lpp in lpps:
            knob("fwd_lpp", lpp)
            knob("wpb", wpb)
            xh, ith = hip_fwd(ops, kind, g)
            check_forward(xh, ith, xo, ito)
    knob("fwd_lpp", 0)
    knob("wpb", 0)
    grads, st = hip_bwd(ops, kind, g, torch.from_numpy(xo).cuda())  # wpb = default after the wpb=1 sweep
    check_backward_exact(grads, st, oracle_bwd(oracle, kind, d, xo))


@pytest.mark.parametrize("kind", ["qp", "qcqp"])
@pytest.mark.parametrize("B", [0, 1, 7, 8, 9, 31, 32, 33, 63, 64, 65, 255, 257])
def test_ragged_and_empty_batches(oracle, ops, kind, B):
    N = 8
    if B == 0:
        e = lambda *s: torch.empty(s, dtype=torch.float64, device="cuda")
        if kind == "qp":
            assert ops.qp_forward(e(0, N, N), e(0, N, 1), 1e-7, 100).shape == (0, N, 1)
        else:
            assert ops.qcqp_forward(e(0, N, N), e(0, N, 1), e(0, 4, 1), e(0, 4, 1), 1e-7, 100).shape == (0, N, 1)
        return
    d = make_problem(kind, B, N, 300 + B)
    g = dev(d)
    # guard bands: the kernels must not write past the batch
    xbuf = torch.full((B + 4, N, 1), 7.0, dtype=torch.float64, device="cuda")
    xo, ito = oracle_fwd(oracle, kind, d)
    if kind == "qp":
        ops.qp_forward(g["P"], g["q"], 1e-7, 1000, out=xbuf[:B])
    else:
        ops.qcqp_forward(g["P"], g["q"], g["l_n"], g["mu"], 1e-7, 1000, out=xbuf[:B])
    assert (xbuf[B:] == 7.0).all()
    assert np.abs(npy(xbuf[:B]) - xo).max() <= X_TOL
    grads, st = hip_bwd(ops, kind, g, torch.from_numpy(xo).cuda())
    check_backward_exact(grads, st, oracle_bwd(oracle, kind, d, xo))


@pytest.mark.parametrize("kind", ["qp", "qcqp"])
@pytest.mark.parametrize("N,B,structure", [(8, 2051, "diag"), (8, 700, "mixed"), (32, 131, "diag"), (32, 61, "mixed"),
                                           (2, 777, "diag"), (64, 20, "diag"), (8, 20011, "mixed"), (4, 20011, "mixed"),
                                           (2, 20011, "mixed"), (8, 20011, "sparse")])
def test_backward_with_the_forwards_diagonal_cache_is_identical(oracle, ops, kind, N, B, structure):
    """The forward can leave the verified diagonal of P for the backward of the same problems (which then
    skips the P stream); non-diagonal tiles are flagged and still read P.  Results must be bit-identical."""
    if structure == "sparse":   # one non-diagonal problem in 97
        d, dd = make_problem(kind, B, N, 420 + N, "diag"), make_problem(kind, B, N, 421 + N, "dense")
        d["P"][5::97] = dd["P"][5::97]
    else:
        d = make_problem(kind, B, N, 420 + N, structure)
    g = dev(d)
    cache = ops.diag_cache(g["q"])
    cache[1].fill_(1)  # stale flags must be overwritten by the forward, not trusted
    if kind == "qp":
        x = ops.qp_forward(g["P"], g["q"], 1e-7, 1000, cache=cache)
        a = ops.qp_backward(g["P"], g["q"], x, g["grad_x"])
        b = ops.qp_backward(g["P"], g["q"], x, g["grad_x"], cache=cache)
    else:
        x = ops.qcqp_forward(g["P"], g["q"], g["l_n"], g["mu"], 1e-7, 1000, cache=cache)
        a = ops.qcqp_backward(g["P"], g["q"], g["l_n"], g["mu"], x, g["grad_x"])
        b = ops.qcqp_backward(g["P"], g["q"], g["l_n"], g["mu"], x, g["grad_x"], cache=cache)
    for u, v in zip(a, b):
        assert torch.equal(u, v)
    flags = cache[1].cpu().numpy()
    offdiag = (d["P"] - torch.diag_embed(torch.diagonal(d["P"], dim1=1, dim2=2))).abs().amax((1, 2)).numpy() > 0
    assert (flags[offdiag] == 2).all(), "a non-diagonal problem must be flagged 2 (seen, not diagonal)"
    assert np.isin(flags, (1, 2)).all(), "the forward examines every problem"
    if N <= 8:   # problem by problem: a diagonal problem next to a non-diagonal one is still handed over as diagonal, and the
        # backward (B large enough for the work-list route) queues only the flag-2 problems of such a tile
        assert (flags[~offdiag] == 1).all()
        assert torch.equal(cache[0][torch.from_numpy(~offdiag).cuda()],
                           torch.diagonal(g["P"], dim1=1, dim2=2)[torch.from_numpy(~offdiag).cuda()])
    for ws in ops._workspaces.values():
        assert header_is_clean(ws)
    if structure == "diag":
        assert flags.all()
        assert torch.equal(cache[0], torch.diagonal(g["P"], dim1=1, dim2=2))


def test_compact_diagonal_layout_extension(oracle, ops):
    """DQQ_P_DIAG: P given as (B,N) -- same numbers as the (B,N,N) layout."""
    from diffqcqp_amd import _capi
    d = make_problem("qcqp", 1000, 8, 401)
    g = dev(d)
    pdiag = torch.diagonal(g["P"], dim1=1, dim2=2).contiguous()
    x0 = ops.qcqp_forward(g["P"], g["q"], g["l_n"], g["mu"], 1e-7, 1000)
    x1 = ops.qcqp_forward(pdiag, g["q"], g["l_n"], g["mu"], 1e-7, 1000, layout=_capi.P_DIAG)
    assert torch.equal(x0, x1)
    g0 = ops.qcqp_backward(g["P"], g["q"], g["l_n"], g["mu"], x0, g["grad_x"])
    g1 = ops.qcqp_backward(pdiag, g["q"], g["l_n"], g["mu"], x0, g["grad_x"], layout=_capi.P_DIAG)
    assert torch.equal(torch.diagonal(g0[0], dim1=1, dim2=2), g1[0])
    for a, b in zip(g0[1:], g1[1:]):
        assert torch.equal(a, b)


# ---------------------------------------------------------------- general dense path
@pytest.mark.parametrize("kind", ["qp", "qcqp"])
@pytest.mark.parametrize("N,B,structure", [(8, 200, "dense"), (8, 100, "diag"), (4, 64, "dense"), (12, 40, "dense"),
                                           (16, 40, "dense"), (32, 12, "dense"), (6, 50, "dense")])
def test_dense_kernel_matches_oracle(oracle, ops, kind, N, B, structure):
    from diffqcqp_amd import _capi
    d = make_problem(kind, B, N, 500 + N, structure)
    g = dev(d)
    xo, ito = oracle_fwd(oracle, kind, d)
    xh, ith = hip_fwd(ops, kind, g, layout=_capi.P_DENSE)
    check_forward(xh, ith, xo, ito, min_match=0.99)
    grads, st = hip_bwd(ops, kind, g, torch.from_numpy(xo).cuda(), layout=_capi.P_DENSE)
    check_dense_backward(oracle, kind, N, {k: v.numpy() for k, v in d.items()}, xo, grads, st,
                         oracle_bwd(oracle, kind, d, xo))


@pytest.mark.parametrize("kind", ["qp", "qcqp"])
@pytest.mark.parametrize("N,B", [(32, 40), (64, 12)])
def test_register_and_lds_wave_dense_forward_agree(oracle, ops, kind, N, B):
    """N = 32 / 64 forward has two general kernels: the register-resident wave-per-problem kernel on the matrix cores
    (default) and the LDS wave kernel in the reference's summation order ("dense_wave64" = 0).  Both follow the oracle's
    trajectory; they differ only in summation order."""
    from diffqcqp_amd import _capi
    d = make_problem(kind, B, N, 650 + N, "dense")
    g = dev(d)
    xo, ito = oracle_fwd(oracle, kind, d)
    out = {}
    try:
        for reg in (1, 0):
            knob("dense_wave64", reg)
            out[reg] = hip_fwd(ops, kind, g, layout=_capi.P_DENSE)
            check_forward(out[reg][0], out[reg][1], xo, ito, min_match=0.9)
    finally:
        knob("dense_wave64", 1)
    assert (out[0][0] - out[1][0]).abs().max() < 1e-8


@pytest.mark.parametrize("N,B", [(32, 40), (64, 12), (64, 300)])
def test_register_and_lds_wave_dense_backward_agree(oracle, ops, N, B):
    """N = 32 / 64 QP backward has two general kernels: register-resident block Cholesky on the matrix cores (default) and
    the LDS wave kernel in the reference's operation order ("dense_wave64" = 0).  Both are checked against the oracle."""
    from diffqcqp_amd import _capi
    d = make_problem("qp", B, N, 690 + N, "dense")
    g = dev(d)
    xo, _ = oracle_fwd(oracle, "qp", d)
    ref = oracle_bwd(oracle, "qp", d, xo)
    out = {}
    try:
        for reg in (1, 0):
            knob("dense_wave64", reg)
            out[reg] = hip_bwd(ops, "qp", g, torch.from_numpy(xo).cuda(), layout=_capi.P_DENSE)
            check_backward_exact(out[reg][0], out[reg][1], ref, exact=False)
    finally:
        knob("dense_wave64", 1)
    for a, b in zip(out[0][0], out[1][0]):
        assert (a - b).abs().max() <= 1e-9 * max(1.0, float(b.abs().max()))


@pytest.mark.parametrize("kind", ["qp", "qcqp"])
@pytest.mark.parametrize("N,B", [(8, 333), (4, 100), (6, 77), (12, 50), (16, 41), (32, 9)])
def test_dense_backward_teams_agree_with_one_problem_per_wave(oracle, ops, kind, N, B):
    """The general backward packs 64/T problems per wave for small N; same arithmetic as T = 64."""
    from diffqcqp_amd import _capi
    d = make_problem(kind, B, N, 680 + N, "dense")
    g = dev(d)
    xo, _ = oracle_fwd(oracle, kind, d)
    ref = oracle_bwd(oracle, kind, d, xo)
    out = {}
    knob("dense_wave64", 0)   # the LDS kernels (the matrix-core kernels take 16 < N by default)
    knob("wave_qcqp_bwd", 0)
    try:
        for teams in (1, 0):
            knob("dense_teams", teams)
            out[teams] = hip_bwd(ops, kind, g, torch.from_numpy(xo).cuda(), layout=_capi.P_DENSE)
            check_backward_exact(out[teams][0], out[teams][1], ref, exact=False)
    finally:
        knob("dense_teams", 1)
        knob("dense_wave64", 1)
        knob("wave_qcqp_bwd", 1)
    for a, b in zip(out[0][0], out[1][0]):
        assert torch.equal(a, b)  # identical operation order -> identical bits


@pytest.mark.parametrize("kind", ["qp", "qcqp"])
@pytest.mark.parametrize("N,B", [(8, 1500), (6, 300), (4, 400), (2, 130), (10, 90), (12, 120), (16, 70)])
def test_small_dense_backward_is_bit_exact_and_matches_the_wave_kernel(oracle, ops, kind, N, B):
    """Even N <= 16, dense P: the statically sized team kernel (bwd_small.hip, default) keeps the reference's
    operation order -- inactive multipliers are decoupled zero slots -- so it reproduces the oracle bit for bit
    on identical x, like the run-time sized kernel behind it."""
    from diffqcqp_amd import _capi
    d = make_problem(kind, B, N, 700 + N, "dense")
    g = dev(d)
    xo, _ = oracle_fwd(oracle, kind, d)
    ref = oracle_bwd(oracle, kind, d, xo)
    out = {}
    for opt in (1, 0):
        knob("small_bwd", opt)
        out[opt] = hip_bwd(ops, kind, g, torch.from_numpy(xo).cuda(), layout=_capi.P_DENSE)
    knob("small_bwd", 1)
    check_backward_exact(out[1][0], out[1][1], ref, exact=True)
    check_backward_exact(out[0][0], out[0][1], ref, exact=False)


@pytest.mark.parametrize("kind", ["qp", "qcqp"])
@pytest.mark.parametrize("N,B", [(8, 24576 + 77), (8, 30000), (6, 16384), (4, 16500), (2, 16384 + 1)])
def test_lane_per_problem_backward_is_the_team_kernel_bit_for_bit(oracle, ops, kind, N, B):
    """Dense P declared dense, B >= 16384 (N = 8: 24576): one lane per problem (bwd_lane_dense.hip) -- triangular loops, structural
    zeros left out, K in LDS, the factor and the explicit inverse in registers.  Same operation order as the team kernel
    (bwd_small.hip) and the oracle: every output and every refinement step count identical, ragged last wave included;
    bit-exact against the oracle on the oracle's x."""
    from diffqcqp_amd import _capi
    d = make_problem(kind, B, N, 760 + N, "dense")
    g = dev(d)
    n = 3000
    xo, _ = oracle_fwd(oracle, kind, {k: v[:n] for k, v in d.items()})
    x = hip_fwd(ops, kind, g, layout=_capi.P_DENSE)[0]
    x[:n] = torch.from_numpy(xo).cuda()          # the first n problems: the oracle's x, the rest: the kernel's own
    out = {}
    try:
        for opt in (1, 0):
            knob("lane_bwd", opt)
            out[opt] = hip_bwd(ops, kind, g, x, layout=_capi.P_DENSE)
    finally:
        knob("lane_bwd", 1)
    for a, b in zip(out[1][0], out[0][0]):
        assert torch.equal(a, b)
    assert torch.equal(out[1][1], out[0][1])
    ref = oracle_bwd(oracle, kind, {k: v[:n] for k, v in d.items()}, xo)
    check_backward_exact([t[:n] for t in out[1][0]], out[1][1][:n], ref, exact=True)


@pytest.mark.parametrize("kind", ["qp", "qcqp"])
@pytest.mark.parametrize("N,B,ndiag", [(8, 24576 + 10240 + 13, 10240), (4, 16384 + 6144 + 500, 6144)])
def test_feedback_routes_a_long_work_list_to_the_lane_kernel_same_bits(ops, kind, N, B, ndiag):
    """DQQ_P_AUTO, a batch that is (almost) all dense: the drain launch behind the diagonal backward reports the length of its
    work-list to the report word (include/diffqcqp_hip.h: dqq_hint_flags); the next backward of the same kind, N and B drains with the
    lane-per-problem kernel (bwd_lane_dense.hip, LIST) instead of the team kernel.  Same bits, whichever drains; the
    work-list header is left clean (a third call); a stale word (the list has become short, or empty) costs time only."""
    from diffqcqp_amd import _capi
    d = make_problem(kind, B, N, 790 + N, "dense")
    # ndiag: a diagonal stretch (whole tiles) in the middle -- those take the fast path; more than a quarter of the batch, so
    # that the list stays the route (from three quarters non-diagonal on, twice running, the lane kernel takes the batch whole)
    for i in range(N):
        for j in range(N):
            if i != j:
                d["P"][3008:3008 + ndiag, i, j] = 0.0
    g = dev(d)
    x = hip_fwd(ops, kind, g, layout=_capi.P_AUTO)[0]
    slot = (0 if kind == "qp" else 1) * 4 + N // 2 - 1
    was_on = _capi._feedback is not None
    _capi.enable_feedback(True)
    knob("lane_list_drains", 0)
    try:
        _capi._feedback.zero_()
        team = hip_bwd(ops, kind, g, x)             # nothing known yet: the team kernel drains, and reports
        torch.cuda.synchronize()
        assert _capi.get_option("lane_list_drains") == 0
        assert _capi.feedback_words()[slot] == (B, B - ndiag)
        lane = hip_bwd(ops, kind, g, x)             # the word says "long": the lane kernel drains
        lane2 = hip_bwd(ops, kind, g, x)            # ... and left the list's header clean
        torch.cuda.synchronize()
        assert _capi.get_option("lane_list_drains") == 2
        assert _capi.feedback_words()[slot] == (B, B - ndiag)
        for other in (lane, lane2):
            for a, b in zip(team[0], other[0]):
                assert torch.equal(a, b)
            assert torch.equal(team[1], other[1])
        # the same B, now diagonal but for 70 problems: the stale word sends the lane kernel after a short list
        d2 = make_problem(kind, B, N, 791 + N, "diag")
        d2["P"][128:198] = d["P"][128:198]           # (tiles are pushed whole: 70 dense problems queue 5 tiles of 16 = 80)
        g2 = dev(d2)
        x2 = hip_fwd(ops, kind, g2, layout=_capi.P_AUTO)[0]
        stale = hip_bwd(ops, kind, g2, x2)
        torch.cuda.synchronize()
        assert _capi.get_option("lane_list_drains") == 3
        assert _capi.feedback_words()[slot][0] == B and 70 <= _capi.feedback_words()[slot][1] <= 70 + 64
        fresh = hip_bwd(ops, kind, g2, x2)          # corrected: the team kernel again
        torch.cuda.synchronize()
        assert _capi.get_option("lane_list_drains") == 3
        for a, b in zip(stale[0], fresh[0]):
            assert torch.equal(a, b)
        assert torch.equal(stale[1], fresh[1])
        # a stale "long" in front of an EMPTY list (a diagonal batch): every wave of the lane kernel leaves at once
        _capi._feedback[slot] = (B << 32) | B
        d3 = make_problem(kind, B, N, 792 + N, "diag")
        g3 = dev(d3)
        x3 = hip_fwd(ops, kind, g3, layout=_capi.P_AUTO)[0]
        e1 = hip_bwd(ops, kind, g3, x3)
        torch.cuda.synchronize()
        assert _capi.get_option("lane_list_drains") == 4 and _capi.feedback_words()[slot] == (B, 0)
        e2 = hip_bwd(ops, kind, g3, x3)
        for a, b in zip(e1[0], e2[0]):
            assert torch.equal(a, b)
        # without the buffer: the team kernel, whatever came before
        _capi.enable_feedback(False)
        knob("lane_list_drains", 0)
        off = hip_bwd(ops, kind, g, x)
        torch.cuda.synchronize()
        assert _capi.get_option("lane_list_drains") == 0
        for a, b in zip(team[0], off[0]):
            assert torch.equal(a, b)
    finally:
        _capi.enable_feedback(was_on)


@pytest.mark.parametrize("kind", ["qp", "qcqp"])
@pytest.mark.parametrize("N,B", [(8, 24576 + 3 * 1024 + 5), (4, 16384 + 777), (2, 16384 + 64)])
def test_feedback_sends_an_all_dense_auto_batch_to_the_lane_kernel_whole(ops, kind, N, B):
    """DQQ_P_AUTO, every problem non-diagonal, the same kind / N / B step after step: the third backward skips the diagonal fast
    path's launch and its work-list altogether -- ONE launch of the lane-per-problem kernel over the batch, which recounts
    the non-diagonal problems for the call after it (bwd_lane_dense.hip REPORT).  Same bits as the two-launch route; the
    workspace header is left clean; when the batch then stops being all non-diagonal -- a third of it diagonal, all of it
    diagonal -- the stale hint still gives the right bits (a diagonal problem gets the same bits from the general routine
    as from the fast path) and is corrected by that very launch."""
    from diffqcqp_amd import _capi
    d = make_problem(kind, B, N, 797 + N, "dense")
    g = dev(d)
    x = hip_fwd(ops, kind, g)[0]
    slot = (0 if kind == "qp" else 1) * 4 + N // 2 - 1
    was_on = _capi._feedback is not None
    _capi.enable_feedback(False)
    ref = hip_bwd(ops, kind, g, x)                              # no hint: fast path's launch + team drain
    _capi.enable_feedback(True)
    same = lambda a, b: all(torch.equal(u, v) for u, v in zip(a[0], b[0])) and torch.equal(a[1], b[1])
    try:
        _capi._feedback.zero_()
        knob("bwd_whole_batches", 0)
        for expect_whole, expect_streak in ((0, 0), (0, 1), (1, 2), (2, 3), (3, 3)):
            out = hip_bwd(ops, kind, g, x)
            torch.cuda.synchronize()
            assert same(ref, out)
            assert _capi.get_option("bwd_whole_batches") == expect_whole
            assert _capi.feedback_words()[slot] == (B, B) and _capi.feedback_streaks()[slot] == expect_streak
            for ws in ops._workspaces.values():
                assert header_is_clean(ws)
        # a third of the batch turns diagonal (whole tiles): the hint is stale for one call
        d3 = make_problem(kind, B, N, 798 + N, "diag")
        third = (B // 3) // 64 * 64
        d["P"][512:512 + third] = d3["P"][512:512 + third]
        g = dev(d)
        x = hip_fwd(ops, kind, g)[0]
        _capi.enable_feedback(False)
        ref = hip_bwd(ops, kind, g, x)
        _capi.enable_feedback(True)
        _capi._feedback[slot] = (1 << 62) | (B << 32) | B        # (a fresh buffer: what the calls above had left in the old one)
        out = hip_bwd(ops, kind, g, x)
        torch.cuda.synchronize()
        assert same(ref, out) and _capi.get_option("bwd_whole_batches") == 4
        assert _capi.feedback_words()[slot] == (B, B - third) and _capi.feedback_streaks()[slot] == 0
        out = hip_bwd(ops, kind, g, x)                          # corrected: the fast path's launch again
        torch.cuda.synchronize()
        assert same(ref, out) and _capi.get_option("bwd_whole_batches") == 4
        # ... and a wholly diagonal batch behind a stale "all non-diagonal, twice running"
        g = dev(d3)
        x = hip_fwd(ops, kind, g)[0]
        _capi.enable_feedback(False)
        ref = hip_bwd(ops, kind, g, x)
        _capi.enable_feedback(True)
        _capi._feedback[slot] = ((2 << 62) | (B << 32) | B) - (1 << 64)   # (as a signed 64-bit integer)
        out = hip_bwd(ops, kind, g, x)
        torch.cuda.synchronize()
        assert same(ref, out) and _capi.get_option("bwd_whole_batches") == 5
        assert _capi.feedback_words()[slot] == (B, 0)
        knob("bwd_skip_classify", 0)                # the option keeps the fast path's launch in front
        _capi._feedback[slot] = ((2 << 62) | (B << 32) | B) - (1 << 64)   # (as a signed 64-bit integer)
        out = hip_bwd(ops, kind, g, x)
        torch.cuda.synchronize()
        assert same(ref, out) and _capi.get_option("bwd_whole_batches") == 5
    finally:
        knob("bwd_skip_classify", 1)
        _capi.enable_feedback(was_on)


@pytest.mark.parametrize("kind", ["qp", "qcqp"])
def test_feedback_moves_a_mostly_dense_forward_to_one_lane_same_bits(ops, kind):
    """DQQ_P_AUTO, N = 8, a batch large enough for two lanes per problem: once the backward's drain has reported that half of
    the batch or more is non-diagonal, the next forward of that kind, N and B runs on ONE lane per problem (a problem's whole
    matrix in its lane's registers).  x and the iteration counts are the same bits on one, two and four lanes, for the
    non-diagonal problems and for the diagonal ones (a third of this batch, NOT aligned to any tile: which routine solves
    a problem depends on the problem alone)."""
    from diffqcqp_amd import _capi
    N, B = 8, 57344 + 4096 + 21
    d = make_problem(kind, B, N, 795, "dense")
    d3 = make_problem(kind, B, N, 796, "diag")
    third = B // 3
    d["P"][1021:1021 + third] = d3["P"][1021:1021 + third]    # (whole 16-problem tiles of the backward: 1024 .. 1024 + 16 k)
    g = dev(d)
    slot = (0 if kind == "qp" else 1) * 4 + N // 2 - 1
    was_on = _capi._feedback is not None
    _capi.enable_feedback(True)
    try:
        _capi._feedback.zero_()
        knob("fwd_feedback_routes", 0)
        x2, it2 = hip_fwd(ops, kind, g)                       # nothing known: two lanes per problem
        assert _capi.get_option("fwd_feedback_routes") == 0
        hip_bwd(ops, kind, g, x2)
        torch.cuda.synchronize()
        fb_B, fb_n = _capi.feedback_words()[slot]
        assert fb_B == B and B - third <= fb_n <= B - third + 32   # (the backward queues whole tiles of 16)
        x4, it4 = hip_fwd(ops, kind, g)                       # most of the batch was non-diagonal: one lane per problem
        assert _capi.get_option("fwd_feedback_routes") == 1
        assert torch.equal(x2, x4) and torch.equal(it2, it4)
        for lpp in (1, 4):                                    # (what the hint selects are the instantiations of option fwd_lpp)
            knob("fwd_lpp", lpp)
            xf, itf = hip_fwd(ops, kind, g)
            assert torch.equal(x4, xf) and torch.equal(it4, itf)
        knob("fwd_lpp", 0)
        knob("fwd_feedback", 0)
        hip_fwd(ops, kind, g)
        assert _capi.get_option("fwd_feedback_routes") == 1   # the option keeps the forward off the word
        knob("fwd_feedback", 1)
        _capi._feedback[slot] = (B << 32) | (B // 2 - 1)      # fewer than half of the batch last time: two lanes stay
        x1, it1 = hip_fwd(ops, kind, g)
        assert _capi.get_option("fwd_feedback_routes") == 1 and torch.equal(x1, x2) and torch.equal(it1, it2)
    finally:
        knob("fwd_lpp", 0)
        knob("fwd_feedback", 1)
        _capi.enable_feedback(was_on)


@pytest.mark.parametrize("kind", ["qp", "qcqp"])
@pytest.mark.parametrize("N,B", [(8, 30000), (4, 20000), (2, 17000)])
def test_a_diagonal_problem_gets_the_fast_paths_bits_from_the_general_backward(ops, kind, N, B):
    """bwd_diag.hip queues whole tiles, so diagonal problems do land in the general kernels; and a DQQ_P_AUTO batch that was
    (almost) all non-diagonal twice running is sent to the lane-per-problem kernel WHOLE, without a look at P (launch.h:
    feedback).  Both rest on this: the general backward kernels -- team and lane per problem -- give a diagonal problem the
    very bits of the diagonal fast path.  Checked on the ill-conditioned diagonals of the reference's figure workload
    (exp(U(-10, 10))) with singular coordinates mixed in."""
    from diffqcqp_amd import _capi
    gen = torch.Generator().manual_seed(5 + N)
    p = torch.exp(torch.rand(B, N, generator=gen, dtype=torch.float64) * 20 - 10)
    p[::7, 0] = 0.0
    r = lambda *s: torch.rand(*s, generator=gen, dtype=torch.float64)
    g = dev({"P": torch.diag_embed(p).contiguous(), "q": 2 * r(B, N, 1) - 1, "l_n": r(B, N // 2, 1), "mu": r(B, N // 2, 1),
             "grad_x": torch.randn(B, N, 1, generator=gen, dtype=torch.float64)})
    x = hip_fwd(ops, kind, g)[0]
    fast = hip_bwd(ops, kind, g, x, layout=_capi.P_AUTO)
    bits = lambda t: t.view(torch.int64) if t.dtype is torch.float64 else t
    try:
        for lane in (1, 0):
            knob("lane_bwd", lane)
            general = hip_bwd(ops, kind, g, x, layout=_capi.P_DENSE)
            for a, b in zip(fast[0] + [fast[1]], general[0] + [general[1]]):
                assert torch.equal(bits(a), bits(b))
    finally:
        knob("lane_bwd", 1)


@pytest.mark.parametrize("kind", ["qp", "qcqp"])
@pytest.mark.parametrize("N", [8, 4, 2])
def test_a_problems_result_does_not_depend_on_its_neighbours_or_the_batch_size(ops, kind, N):
    """DQQ_P_AUTO, N <= 8: which routine solves a problem -- the diagonal arithmetic or the in-kernel general solve -- is decided
    problem by problem, and neither depends on the lane layout; the backward's general kernels give a diagonal problem the bits
    of its fast path.  So the same problems, shuffled among each other and cut into batches of other sizes (other layouts:
    four lanes per problem below 57344 problems at N = 8, two above), give the same x, iteration counts, gradients and
    refinement step counts, bit for bit."""
    B = 60000
    dense, diag = make_problem(kind, B, N, 821, "dense"), make_problem(kind, B, N, 822, "diag")
    gen = torch.Generator().manual_seed(823)
    pick = torch.rand(B, generator=gen) < 0.07                  # 7 % non-diagonal, scattered
    d = {k: torch.where(pick.view(-1, *([1] * (v.dim() - 1))), dense[k], diag[k]) for k, v in diag.items()}
    perm = torch.randperm(B, generator=gen)
    bits = lambda t: t.view(torch.int64) if t.dtype is torch.float64 else t

    def run(t):
        g = dev(t)
        x, it = hip_fwd(ops, kind, g)
        gr, st = hip_bwd(ops, kind, g, x)
        return [x, it] + gr + [st]

    ref = run(d)
    shuffled = run({k: v[perm].contiguous() for k, v in d.items()})
    for a, b in zip(ref, shuffled):
        assert torch.equal(bits(a[perm]), bits(b))
    for lo, hi in ((0, 777), (777, 777 + 20011), (30000, 60000)):
        part = run({k: v[lo:hi].contiguous() for k, v in d.items()})
        for a, b in zip(ref, part):
            assert torch.equal(bits(a[lo:hi]), bits(b))


@pytest.mark.parametrize("kind", ["qp", "qcqp"])
@pytest.mark.parametrize("every,one_lane", [(8, True), (100, False)])
def test_forward_hint_with_the_hand_off_counts_blocks_not_problems(ops, kind, every, one_lane):
    """With the forward's hand-off the backward queues single problems, and the word says so (bit 31): the forward's hint --
    which pays per 16-problem block with a non-diagonal problem, not per problem -- estimates the blocks from the count.
    One problem in 8, scattered: nearly every block has one, one lane per problem; one in 100: two lanes stay.  Same bits."""
    from diffqcqp_amd import _capi
    N, B = 8, 57344 + 2048 + 5
    d, dd = make_problem(kind, B, N, 841, "diag"), make_problem(kind, B, N, 842, "dense")
    d["P"][3::every] = dd["P"][3::every]
    g = dev(d)
    ndense = len(range(3, B, every))
    slot = (0 if kind == "qp" else 1) * 4 + N // 2 - 1
    was_on = _capi._feedback is not None
    _capi.enable_feedback(True)
    try:
        _capi._feedback.zero_()
        knob("fwd_feedback_routes", 0)
        cache = ops.diag_cache(g["q"])
        if kind == "qp":
            fwd = lambda: ops.qp_forward(g["P"], g["q"], 1e-7, 1000, cache=cache, return_iters=True)
            bwd = lambda x: ops.qp_backward(g["P"], g["q"], x, g["grad_x"], cache=cache)
        else:
            fwd = lambda: ops.qcqp_forward(g["P"], g["q"], g["l_n"], g["mu"], 1e-7, 1000, cache=cache, return_iters=True)
            bwd = lambda x: ops.qcqp_backward(g["P"], g["q"], g["l_n"], g["mu"], x, g["grad_x"], cache=cache)
        x2, it2 = fwd()
        b2 = bwd(x2)
        torch.cuda.synchronize()
        assert _capi.get_option("fwd_feedback_routes") == 0
        assert _capi.feedback_words()[slot] == (B, ndense)                      # single problems ...
        assert (int(_capi._feedback[slot]) >> 31) & 1 == 1                      # ... and the word says so
        x1, it1 = fwd()
        b1 = bwd(x1)
        torch.cuda.synchronize()
        assert _capi.get_option("fwd_feedback_routes") == (1 if one_lane else 0)
        assert torch.equal(x1, x2) and torch.equal(it1, it2)
        for u, v in zip(b1, b2):
            assert torch.equal(u, v)
    finally:
        _capi.enable_feedback(was_on)


@pytest.mark.parametrize("kind", ["qp", "qcqp"])
def test_racing_hints_never_change_a_result(ops, kind):
    """Two streams, each with its own workspace, alternate dense, mixed and diagonal batches of ONE (kind, N, B) without ever
    waiting for each other: the feedback word is written and read in every order the hardware produces, most hints are
    stale or belong to the other stream's batch.  Every forward and backward gives the bits of the hint-free run."""
    from diffqcqp_amd import _capi
    N, B = 8, 57344 + 2048 + 7
    dense, diag = make_problem(kind, B, N, 811, "dense"), make_problem(kind, B, N, 812, "diag")
    mixed = {k: v.clone() for k, v in diag.items()}
    mixed["P"][5000:5003] = dense["P"][5000:5003]
    mostly = {k: v.clone() for k, v in dense.items()}
    mostly["P"][64 * 100:64 * 160] = diag["P"][64 * 100:64 * 160]
    batches = [dev(b) for b in (dense, diag, mixed, mostly)]
    was_on = _capi._feedback is not None
    _capi.enable_feedback(False)
    ref = []
    for g in batches:
        x, it = hip_fwd(ops, kind, g)
        ref.append((x, it, hip_bwd(ops, kind, g, x)))
    torch.cuda.synchronize()
    _capi.enable_feedback(True)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    got = []
    try:
        order = [0, 0, 0, 1, 3, 3, 3, 2, 0, 1, 1, 0, 3, 2, 2, 0, 0, 0, 0, 1]
        for i, b in enumerate(order):
            with torch.cuda.stream(streams[i % 2]):
                g = batches[b if i % 2 == 0 else (b + 1) % 4]
                x, it = hip_fwd(ops, kind, g)
                got.append((b if i % 2 == 0 else (b + 1) % 4, x, it, hip_bwd(ops, kind, g, ref[b if i % 2 == 0 else (b + 1) % 4][0])))
        torch.cuda.synchronize()
        for b, x, it, bw in got:
            assert torch.equal(x, ref[b][0]) and torch.equal(it, ref[b][1])
            for u, v in zip(bw[0] + [bw[1]], ref[b][2][0] + [ref[b][2][1]]):
                assert torch.equal(u, v)
    finally:
        _capi.enable_feedback(was_on)


def test_two_host_threads_two_streams_same_bits(ops):
    """The entry points are re-entrant (include/diffqcqp_hip.h): two host threads, each on its own stream with its own
    workspace, run QP and QCQP forward + backward on dense, mixed and diagonal batches at the same time, the feedback word
    shared between them.  Every result is the single-threaded, hint-free run's, bit for bit."""
    import threading
    from diffqcqp_amd import _capi
    N, B = 8, 57344 + 1024 + 11
    data = {}
    for kind in ("qp", "qcqp"):
        dense, diag = make_problem(kind, B, N, 831, "dense"), make_problem(kind, B, N, 832, "diag")
        mixed = {k: v.clone() for k, v in diag.items()}
        mixed["P"][777:790] = dense["P"][777:790]
        data[kind] = [dev(b) for b in (dense, diag, mixed)]
    was_on = _capi._feedback is not None
    _capi.enable_feedback(False)
    ref = {}
    for kind, batches in data.items():
        for i, g in enumerate(batches):
            x, it = hip_fwd(ops, kind, g)
            ref[kind, i] = (x, it, hip_bwd(ops, kind, g, x))
    torch.cuda.synchronize()
    _capi.enable_feedback(True)
    got, errors = [], []

    def worker(kind, order):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream()):
                for i in order:
                    g = data[kind][i]
                    x, it = hip_fwd(ops, kind, g)
                    got.append((kind, i, x, it, hip_bwd(ops, kind, g, ref[kind, i][0])))
                torch.cuda.current_stream().synchronize()
        except Exception as e:  # noqa: BLE001  (reported by the main thread)
            errors.append(e)

    try:
        threads = [threading.Thread(target=worker, args=("qp", [0, 0, 0, 1, 2, 0, 1, 1, 2, 0, 0, 0])),
                   threading.Thread(target=worker, args=("qcqp", [1, 0, 0, 0, 2, 2, 1, 0, 0, 0, 1, 2]))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        torch.cuda.synchronize()
        assert not errors, errors
        assert len(got) == 24
        for kind, i, x, it, bw in got:
            assert torch.equal(x, ref[kind, i][0]) and torch.equal(it, ref[kind, i][1])
            for u, v in zip(bw[0] + [bw[1]], ref[kind, i][2][0] + [ref[kind, i][2][1]]):
                assert torch.equal(u, v)
    finally:
        _capi.enable_feedback(was_on)


@pytest.mark.parametrize("kind,N,B", [("qcqp", 64, 40), ("qcqp", 50, 24), ("qcqp", 44, 24), ("box", 32, 48), ("box", 22, 30)])
def test_reference_order_backward_beyond_the_wave_kernel(oracle, ops, kind, N, B):
    """The global-memory workgroup kernel in the reference's operation order -- the default for box 21 < N <= 32, and for
    QCQP 42 < N <= 64 behind wave_qcqp_bwd = 0 (the matrix-core kernels take 16 < N <= 64 since round 3): same 1e-9 bar
    (and identical refinement step counts) as the LDS wave kernel below those sizes."""
    from diffqcqp_amd import _capi
    d = make_problem(kind, B, N, 730 + N, "dense")
    g = dev(d)
    if kind == "qcqp":
        xo, _ = oracle_fwd(oracle, kind, d)
        knob("wave_qcqp_bwd", 0)
        try:
            grads, st = hip_bwd(ops, kind, g, torch.from_numpy(xo).cuda(), layout=_capi.P_DENSE)
        finally:
            knob("wave_qcqp_bwd", 1)
        check_backward_exact(grads, st, oracle_bwd(oracle, kind, d, xo), exact=False)
    else:
        xo = oracle.boxqp_fwd_batch(d["P"].numpy(), d["q"].numpy(), d["l_min"].numpy(), d["l_max"].numpy(), 1e-7, 1000,
                                    nthreads=8)[0]
        ref, out, duals = _box_bwd(oracle, ops, d, xo, layout=_capi.P_DENSE)
        check_backward_exact(list(out[:4]), out[4], ref[:4] + (ref[5],), exact=False)
        assert np.allclose(npy(duals[0]), ref[4], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("kind", ["qp", "qcqp", "box", "sbox"])
@pytest.mark.parametrize("N,B", [(18, 70), (20, 90), (24, 64), (31, 40), (32, 130), (34, 40), (40, 50), (47, 30), (48, 60),
                                 (50, 30), (56, 40), (63, 20)])
def test_wave_per_problem_forward_every_size(oracle, ops, kind, N, B):
    """16 < N <= 64, dense P: one wave per problem with the matrix in registers (dense_wave64.hip), N padded with the
    identity to the next multiple of 16.  Same trajectory as the oracle; also against the LDS wave kernel."""
    from diffqcqp_amd import _capi
    if kind == "qcqp" and N % 2:
        pytest.skip("QCQP needs an even N")
    d = make_problem(kind, B, N, 5100 + N, "dense")
    g = dev(d)
    if kind in ("qp", "qcqp"):
        xo, ito = oracle_fwd(oracle, kind, d)
        xh, ith = hip_fwd(ops, kind, g, layout=_capi.P_DENSE)
    else:
        xo, ito, xh, ith = _box_fwd(oracle, ops, kind, d, layout=_capi.P_DENSE)
    check_forward(xh, ith, xo, ito, min_match=0.97)
    knob("dense_wave64", 0)
    try:
        if kind in ("qp", "qcqp"):
            xw, itw = hip_fwd(ops, kind, g, layout=_capi.P_DENSE)
        else:
            _, _, xw, itw = _box_fwd(oracle, ops, kind, d, layout=_capi.P_DENSE)
    finally:
        knob("dense_wave64", 1)
    assert (xw - xh).abs().max() < 1e-8 and (itw == ith).float().mean() >= 0.97
    if kind == "qp":
        # QP backward: every 16 < N <= 64 runs on the same register-resident design (K on the matrix cores);
        # same 1e-9 / identical-refinement-steps bar as the reference-order wave kernel
        grads, st = hip_bwd(ops, kind, g, torch.from_numpy(xo).cuda(), layout=_capi.P_DENSE)
        check_backward_exact(grads, st, oracle_bwd(oracle, kind, d, xo), exact=False)
        gq_only = ops.qp_backward(g["P"], g["q"], torch.from_numpy(xo).cuda(), g["grad_x"], need_P=False,
                                  layout=_capi.P_DENSE)
        assert gq_only[0] is None and torch.equal(gq_only[1], grads[1])


@pytest.mark.parametrize("N", [3, 5, 7])
def test_dense_kernel_odd_n_qp(oracle, ops, N):
    d = make_problem("qp", 60, N, 600 + N, "dense")
    g = dev(d)
    xo, ito = oracle_fwd(oracle, "qp", d)
    xh, ith = hip_fwd(ops, "qp", g)  # AUTO: no fast path for odd N -> dense kernel
    check_forward(xh, ith, xo, ito, min_match=0.99)
    grads, st = hip_bwd(ops, "qp", g, torch.from_numpy(xo).cuda())
    check_backward_exact(grads, st, oracle_bwd(oracle, "qp", d, xo), exact=False)


def test_dense_n64_config5_shape(oracle, ops):
    """BASELINE.json configs[4] at a reduced batch: N=64 dense-P QP, P = S S^T/64 + 0.1 I."""
    d = make_problem("qp", 24, 64, 1005, "dense")
    g = dev(d)
    xo, ito = oracle_fwd(oracle, "qp", d)
    xh, ith = hip_fwd(ops, "qp", g)
    check_forward(xh, ith, xo, ito, min_match=0.9)
    grads, st = hip_bwd(ops, "qp", g, torch.from_numpy(xo).cuda())
    check_backward_exact(grads, st, oracle_bwd(oracle, "qp", d, xo), exact=False)


@pytest.mark.parametrize("kind", ["qp", "qcqp"])
@pytest.mark.parametrize("N", [8, 32])
def test_auto_layout_mixed_batch_uses_fallback(oracle, ops, kind, N):
    """Diagonal and dense problems interleaved: tiles with a non-zero off-diagonal go through the
    work-list to the dense kernel, the rest through the fast path; the work-list is left zeroed."""
    from diffqcqp_amd import _capi
    B = 500 if N == 8 else 61
    d = make_problem(kind, B, N, 700 + N, "mixed")
    g = dev(d)
    xo, ito = oracle_fwd(oracle, kind, d)
    # fuse_fallback: 1 = non-diagonal tiles solved inside the fast kernel (small N), 0 = work-list + dense
    # kernel, -1 = built-in choice.  Twice each: the second call relies on the work-list being re-zeroed.
    for fuse in (0, 0, 1, -1):
        knob("fuse_fallback", fuse)
        xh, ith = hip_fwd(ops, kind, g)
        check_forward(xh, ith, xo, ito, min_match=0.99)
        grads, st = hip_bwd(ops, kind, g, torch.from_numpy(xo).cuda())
        if kind == "qcqp":   # (QP: this well-conditioned family never leaves the loop at another body)
            check_dense_backward(oracle, kind, N, {k: v.numpy() for k, v in d.items()}, xo, grads, st,
                                 oracle_bwd(oracle, kind, d, xo))
        else:
            check_backward_exact(grads, st, oracle_bwd(oracle, kind, d, xo), exact=False)
    for ws in ops._workspaces.values():
        assert header_is_clean(ws)  # work-list count, exit tickets, pick-up index: left zeroed


# csrc/launch.h kWsEntries: count, ticket, next, 32 sub-tickets and the 32 segment counters of the N >= 32 list, a cache
# line apart each
WS_HEADER_INTS = 32 + 2 * 32 * 32


def header_is_clean(ws):
    """Count, exit tickets, pick-up index, segment counters: zero after every launch.  Ints 4..7 (kWsFbShadow) are not part
    of the list: the last word this workspace's drain launches sent to the report word (dqq_hint_flags), and where; int 10
    (kWsFbSkips) counts the unchanged reports not sent since (at most 63)."""
    h = ws[:WS_HEADER_INTS].clone()
    assert 0 <= int(h[10]) <= 63
    h[4:8] = 0
    h[10] = 0
    return int(h.abs().sum()) == 0


@pytest.mark.parametrize("kind", ["qp", "qcqp"])
@pytest.mark.parametrize("N,B", [(8, 20000), (8, 70001), (16, 9000), (32, 3000), (64, 700), (32, 8195), (64, 2051)])
def test_worklist_header_is_rezeroed_by_large_drains(ops, kind, N, B):
    """The participants of a drain launch draw exit tickets in two levels (32 sub-tickets, then one) and the last one
    re-zeroes the header.  Dense batches through the work-list at sizes where every sub-ticket has several members,
    three calls in a row on the same workspace, forward and backward: identical to the DQQ_P_DENSE results each
    time, header zero afterwards; then a diagonal batch on the same workspaces finds an empty list."""
    from diffqcqp_amd import _capi
    g = dev(make_problem(kind, B, N, 4100 + N, "dense"))
    xd, itd = hip_fwd(ops, kind, g, layout=1)
    gd, std = hip_bwd(ops, kind, g, xd, layout=1)
    knob("fuse_fallback", 0)
    try:
        first = None
        for _ in range(3):
            xa, ita = hip_fwd(ops, kind, g)
            ga, sta = hip_bwd(ops, kind, g, xd)
            torch.cuda.synchronize()
            for ws in ops._workspaces.values():
                assert header_is_clean(ws)
            if first is None:
                first = (xa, ita, ga, sta)
                # the general kernels behind the work-list may differ from the ones DQQ_P_DENSE picks for this size
                assert float((ita == itd).double().mean()) >= 0.999
                same = (ita == itd)
                assert float((xa - xd)[same].abs().max()) < 1e-9
                agree = (sta == std).reshape(sta.shape[0], -1).all(1)
                assert float(agree.double().mean()) > 0.85
                for u, v in zip(ga, gd):
                    scale = float(v.abs().max()) + 1e-300
                    assert float((u - v)[agree].abs().max()) / scale < 1e-6
            else:       # the same launches again on the same workspace: the same bits
                assert torch.equal(xa, first[0]) and torch.equal(ita, first[1]) and torch.equal(sta, first[3])
                for u, v in zip(ga, first[2]):
                    assert torch.equal(u, v)
        gdiag = dev(make_problem(kind, B, N, 4200 + N))
        x1, it1 = hip_fwd(ops, kind, gdiag)
        assert torch.isfinite(x1).all() and int(it1.min()) >= 1
        torch.cuda.synchronize()
        for ws in ops._workspaces.values():
            assert header_is_clean(ws)
    finally:
        knob("fuse_fallback", -1)


def test_single_nonzero_offdiagonal_is_detected(oracle, ops):
    """One tiny off-diagonal entry anywhere in a tile must route that tile to the dense kernel."""
    N, B = 8, 256
    d = make_problem("qp", B, N, 801)
    for (b, i, j) in [(0, 0, 1), (77, 7, 0), (130, 3, 4), (255, 6, 7)]:
        d["P"][b, i, j] = 1e-3
        d["P"][b, j, i] = 1e-3
    d["P"][200, 2, 5] = -0.0  # a negative zero is still zero
    g = dev(d)
    xo, ito = oracle_fwd(oracle, "qp", d)
    xh, ith = hip_fwd(ops, "qp", g)
    check_forward(xh, ith, xo, ito, min_match=0.99)
    # and the detection matters: the diagonal-only answer differs on the touched problems
    pd = torch.diagonal(g["P"], dim1=1, dim2=2).contiguous()
    from diffqcqp_amd import _capi
    xd = ops.qp_forward(pd, g["q"], 1e-7, 1000, layout=_capi.P_DIAG)
    assert (xd[0] - xh[0]).abs().max() > 1e-6


def test_bad_inputs_terminate_and_poison_only_their_own_problem(oracle, ops):
    """The reference signals nothing on bad input (LLT success is never checked, Solver.cpp:76): the result
    is NaN.  Here: every launch terminates (max_iter bounds the loop), a bad problem yields NaN, and its
    neighbours in the same tile are unaffected."""
    from diffqcqp_amd import _capi
    N, B = 8, 300
    d = make_problem("qcqp", B, N, 990)
    xo, _ = oracle_fwd(oracle, "qcqp", d)
    bad = {5: "zero", 70: "negative", 131: "nan", 299: "inf"}
    for b, what in bad.items():
        if what == "zero":
            d["P"][b] = 0.0
        elif what == "negative":
            d["P"][b, 3, 3] = -50.0
        elif what == "nan":
            d["P"][b, 2, 2] = float("nan")
        else:
            d["P"][b, 1, 1] = float("inf")
    g = dev(d)
    for layout in (_capi.P_AUTO, _capi.P_DENSE):
        for kind in ("qp", "qcqp"):
            x, it = hip_fwd(ops, kind, g, layout=layout, max_iter=200)
            torch.cuda.synchronize()
            assert int(it.max()) <= 200
            xn = npy(x)
            for b in bad:
                assert not np.isfinite(xn[b]).all(), (layout, kind, b, xn[b].ravel())
            good = np.array([b not in bad for b in range(B)])
            assert np.isfinite(xn[good]).all()
            if kind == "qcqp":
                assert np.abs(xn[good] - xo[good]).max() <= X_TOL
    # the backward must terminate on non-finite input as well
    grads, st = hip_bwd(ops, "qcqp", g, torch.from_numpy(xo).cuda())
    torch.cuda.synchronize()
    assert int(st.max()) <= 10


# ---------------------------------------------------------------- box QP / signed box QP
def _box_fwd(O, ops, kind, d, layout=0, eps=1e-7, max_iter=1000):
    v = d.get("v")
    xo, ito = O.boxqp_fwd_batch(d["P"].numpy(), d["q"].numpy(), d["l_min"].numpy(), d["l_max"].numpy(), eps, max_iter,
                                v=None if v is None else v.numpy(), nthreads=8)
    g = dev(d)
    xh, ith = ops.boxqp_forward(g["P"], g["q"], g["l_min"], g["l_max"], eps, max_iter, v=g.get("v"), layout=layout,
                                return_iters=True)
    return xo, ito, xh, ith


@pytest.mark.parametrize("kind", ["box", "sbox"])
@pytest.mark.parametrize("N,B", [(2, 130), (4, 500), (8, 2051), (16, 301), (32, 131), (64, 37)])
def test_box_forward_diag_fast_path(oracle, ops, kind, N, B):
    d = make_problem(kind, B, N, 800 + N)
    xo, ito, xh, ith = _box_fwd(oracle, ops, kind, d)
    check_forward(xh, ith, xo, ito)
    lo, hi = d["l_min"].numpy(), d["l_max"].numpy()
    assert (npy(xh) >= lo).all() and (npy(xh) <= hi).all()
    if kind == "sbox":
        assert (np.sign(d["v"].numpy()) * npy(xh) <= 0).all()


@pytest.mark.parametrize("kind", ["box", "sbox"])
@pytest.mark.parametrize("N,B,structure", [(8, 200, "dense"), (5, 60, "dense"), (16, 40, "dense"), (32, 12, "dense"),
                                           (64, 6, "dense"), (6, 70, "dense"), (8, 300, "mixed"), (32, 40, "mixed")])
def test_box_forward_dense_and_mixed(oracle, ops, kind, N, B, structure):
    from diffqcqp_amd import _capi
    d = make_problem(kind, B, N, 820 + N, structure)
    for layout in ((_capi.P_AUTO,) if structure == "mixed" else (_capi.P_AUTO, _capi.P_DENSE)):
        xo, ito, xh, ith = _box_fwd(oracle, ops, kind, d, layout=layout)
        check_forward(xh, ith, xo, ito, min_match=0.99 if N < 32 else 0.9)
    if structure == "dense" and N in (8, 32):  # the wave kernel behind the lane / workgroup kernels agrees
        knob("lane_dense", 0)
        knob("dense_wave64", 0)
        try:
            xo, ito, xw, itw = _box_fwd(oracle, ops, kind, d, layout=_capi.P_DENSE)
        finally:
            knob("lane_dense", 1)
            knob("dense_wave64", 1)
        check_forward(xw, itw, xo, ito, min_match=0.99)
        assert (xw - xh).abs().max() < 1e-8


def test_box_forward_reduces_to_qp(oracle, ops):
    """l_min = 0, l_max = huge: the box kernel follows the QP kernel's trajectory."""
    d = make_problem("qp", 777, 8, 840)
    g = dev(d)
    x0, it0 = ops.qp_forward(g["P"], g["q"], 1e-7, 1000, return_iters=True)
    x1, it1 = ops.boxqp_forward(g["P"], g["q"], torch.zeros_like(g["q"]), torch.full_like(g["q"], 1e300), 1e-7, 1000,
                                return_iters=True)
    assert torch.equal(it0, it1) and (x0 - x1).abs().max() < 1e-12


def _box_bwd(O, ops, d, x, layout=0):
    ref = O.boxqp_bwd_batch(d["P"].numpy(), d["q"].numpy(), d["l_min"].numpy(), d["l_max"].numpy(), x,
                            d["grad_x"].numpy(), nthreads=8)
    g = dev(d)
    B, N = d["q"].shape[0], d["q"].shape[1]
    duals = (torch.empty(B, 2 * N, dtype=torch.float64, device="cuda"), torch.empty(B, 2 * N, dtype=torch.float64, device="cuda"))
    out = ops.boxqp_backward(g["P"], g["q"], g["l_min"], g["l_max"], torch.from_numpy(x).cuda(), g["grad_x"],
                             layout=layout, return_steps=True, duals=duals)
    return ref, out, duals


@pytest.mark.parametrize("N,B", [(2, 130), (4, 500), (8, 2051), (16, 301), (32, 131), (64, 37)])
def test_box_backward_diag_fast_path_is_bit_exact(oracle, ops, N, B):
    """Diagonal P, identical x: the per-coordinate blocks reproduce the oracle's dense arithmetic bit for bit
    (gradients, multipliers, both refinement step counts)."""
    d = make_problem("box", B, N, 850 + N)
    xo, _ = oracle.boxqp_fwd_batch(d["P"].numpy(), d["q"].numpy(), d["l_min"].numpy(), d["l_max"].numpy(), 1e-7, 1000,
                                   nthreads=8)
    # make some coordinates sit exactly on a bound and pin a few (l_min == l_max): 3x3 blocks
    d["l_max"][::7, 0, 0] = d["l_min"][::7, 0, 0]
    xo[::7, 0, 0] = d["l_min"].numpy()[::7, 0, 0]
    ref, out, duals = _box_bwd(oracle, ops, d, xo)
    gP, gq, glo, ghi, gam, st = ref
    assert np.array_equal(npy(out[4]), st)
    for a, b in zip(out[:4], (gP, gq, glo, ghi)):
        assert np.array_equal(npy(a), b), "max diff %g" % np.abs(npy(a) - b).max()
    assert np.array_equal(npy(duals[0]), gam)
    active = (gam > 0).reshape(B, 2, N).any(axis=1).mean()
    assert 0.2 < active < 0.95  # the batch exercises active and inactive coordinates


@pytest.mark.parametrize("N,B,structure", [(8, 200, "dense"), (4, 64, "dense"), (5, 60, "dense"), (16, 24, "dense"),
                                           (21, 8, "dense"), (8, 300, "mixed"), (16, 90, "mixed")])
def test_box_backward_dense_and_mixed(oracle, ops, N, B, structure):
    from diffqcqp_amd import _capi
    d = make_problem("box", B, N, 870 + N, structure)
    xo, _ = oracle.boxqp_fwd_batch(d["P"].numpy(), d["q"].numpy(), d["l_min"].numpy(), d["l_max"].numpy(), 1e-7, 1000,
                                   nthreads=8)
    for layout in ((_capi.P_AUTO,) if structure == "mixed" or N in (5, 21) else (_capi.P_AUTO, _capi.P_DENSE)):
        ref, out, duals = _box_bwd(oracle, ops, d, xo, layout=layout)
        check_backward_exact(list(out[:4]), out[4], ref[:4] + (ref[5],), exact=False)
        assert np.allclose(npy(duals[0]), ref[4], rtol=1e-9, atol=1e-12)


def test_box_end_to_end_and_large_n(oracle, ops):
    """HIP forward -> HIP backward against the oracle chain; general-P box backward beyond the wave kernel's
    N <= 21: the global-memory kernel (default, reference order) -- including N = 64 through DQQ_P_AUTO, the
    BoxQPFn2 default, where the fast path queues its non-diagonal tiles for it."""
    from diffqcqp_amd import _capi
    d = make_problem("box", 1000, 8, 890)
    xo, ito, xh, ith = _box_fwd(oracle, ops, "box", d)
    check_forward(xh, ith, xo, ito)
    ref, out, _ = _box_bwd(oracle, ops, d, npy(xh))
    check_end_to_end(list(out[:4]), out[4][:, 1], (ref[0], ref[1], ref[2], ref[3], ref[5][:, 1]))
    assert _capi.lib().dqq_max_n(3, 0) == 21
    for N, B, structure, layout in ((40, 6, "dense", _capi.P_DENSE), (64, 70, "dense", _capi.P_AUTO),
                                    (64, 70, "mixed", _capi.P_AUTO), (26, 9, "dense", _capi.P_DENSE)):
        dd = make_problem("box", B, N, 891 + N, structure)
        xd = oracle.boxqp_fwd_batch(dd["P"].numpy(), dd["q"].numpy(), dd["l_min"].numpy(), dd["l_max"].numpy(), 1e-7,
                                    1000, nthreads=8)[0]
        ref, out, duals = _box_bwd(oracle, ops, dd, xd, layout=layout)
        check_backward_exact(list(out[:4]), out[4], ref[:4] + (ref[5],), exact=False)
        assert np.allclose(npy(duals[0]), ref[4], rtol=1e-9, atol=1e-12)
    # the work-list of the stream is empty again after the mixed AUTO calls
    assert int(ops._workspace(torch.device("cuda", 0), 70)[:2].abs().sum()) == 0


@pytest.mark.parametrize("kind,N,B", [("qp", 65, 5), ("qp", 96, 4), ("qcqp", 70, 4), ("qcqp", 128, 3), ("box", 80, 3),
                                      ("sbox", 72, 3), ("qp", 131, 2)])
def test_no_size_limit(oracle, ops, kind, N, B):
    """The reference solves any n (Solver.cpp:61); beyond what the register / LDS kernels hold (dqq_max_n) the
    global-memory kernels take over: same trajectory, backward in the reference's operation order."""
    d = make_problem(kind, B, N, 7000 + N, "dense")
    g = dev(d)
    if kind in ("qp", "qcqp"):
        xo, ito = oracle_fwd(oracle, kind, d)
        xh, ith = hip_fwd(ops, kind, g)
        check_forward(xh, ith, xo, ito, min_match=0.99)
        grads, st = hip_bwd(ops, kind, g, torch.from_numpy(xo).cuda())
        check_backward_exact(grads, st, oracle_bwd(oracle, kind, d, xo), exact=False)
    else:
        xo, ito, xh, ith = _box_fwd(oracle, ops, kind, d)
        check_forward(xh, ith, xo, ito, min_match=0.99)
        if kind == "box":
            ref, out, _ = _box_bwd(oracle, ops, d, xo)
            check_backward_exact(list(out[:4]), out[4], ref[:4] + (ref[5],), exact=False)


def test_box_autograd_functions_and_module_level_api(oracle, ops):
    """BoxQPFn2 / SignedBoxQPFn2 keep the reference's signatures (qcqp.py:54-137); gradients flow to
    P, q, l_min, l_max; the module-level numpy functions return the reference's shapes."""
    from diffqcqp_amd.qcqp import BoxQPFn2, SignedBoxQPFn2
    from diffqcqp_amd import diffqcqp as M
    d = make_problem("sbox", 64, 8, 895)
    g = {k: v.cuda().requires_grad_(k in ("P", "q", "l_min", "l_max")) for k, v in d.items()}
    ws = torch.zeros_like(g["q"])
    x = BoxQPFn2.apply(g["P"], g["q"], g["l_min"], g["l_max"], ws, 1e-7, 1000)
    assert x.shape == (64, 8, 1)
    (x * g["grad_x"]).sum().backward()
    xo, _ = oracle.boxqp_fwd_batch(d["P"].numpy(), d["q"].numpy(), d["l_min"].numpy(), d["l_max"].numpy(), 1e-7, 1000)
    ref = oracle.boxqp_bwd_batch(d["P"].numpy(), d["q"].numpy(), d["l_min"].numpy(), d["l_max"].numpy(), npy(x),
                                 d["grad_x"].numpy())
    assert np.abs(npy(x) - xo).max() <= X_TOL
    for name, r in zip(("P", "q", "l_min", "l_max"), ref[:4]):
        assert np.array_equal(npy(g[name].grad), r), name  # diagonal P, identical x: bit-exact
    # CPU tensors are staged and come back on the CPU
    xc = BoxQPFn2.apply(d["P"], d["q"], d["l_min"], d["l_max"], torch.zeros_like(d["q"]), 1e-7, 1000)
    assert not xc.is_cuda and torch.equal(xc, x.detach().cpu())
    # signed box: forward only
    xs = SignedBoxQPFn2.apply(g["P"], g["q"], g["l_min"], g["l_max"], g["v"], ws, 1e-7, 1000)
    xso, _ = oracle.boxqp_fwd_batch(d["P"].numpy(), d["q"].numpy(), d["l_min"].numpy(), d["l_max"].numpy(), 1e-7, 1000,
                                    v=d["v"].numpy())
    assert np.abs(npy(xs) - xso).max() <= X_TOL
    with pytest.raises(NotImplementedError):
        xs.sum().backward()
    # module-level numpy API (pybindings.cpp:77-81)
    P0, q0, lo0, hi0, v0 = (d[k][3].numpy() for k in ("P", "q", "l_min", "l_max", "v"))
    x1 = M.solveBoxQP(P0, q0, lo0, hi0, np.zeros(8), 1e-7, 1e-7, 1000)
    assert x1.shape == (8,) and np.abs(x1 - xo[3, :, 0]).max() <= X_TOL
    x2 = M.solveSignedBoxQP(P0, q0, lo0, hi0, v0, np.zeros(8), 1e-7, 1e-7, 1000)
    assert np.abs(x2 - xso[3, :, 0]).max() <= X_TOL
    blg, gam = M.solveDerivativesBoxQP(P0, q0, lo0, hi0, xo[3], d["grad_x"][3].numpy())
    blo, gamo = oracle.solveDerivativesBoxQP(P0, q0, lo0, hi0, xo[3], d["grad_x"][3].numpy())
    assert blg.shape == (24,) and gam.shape == (16,)
    assert np.array_equal(blg, blo) and np.array_equal(gam, gamo)


# ---------------------------------------------------------------- golden fixtures
@pytest.mark.parametrize("path", GENERIC_FIXTURES, ids=os.path.basename)
def test_hip_reproduces_golden(ops, path):
    d = np.load(path)
    eps, mi = float(d["eps"]), int(d["max_iter"])
    if "l_min" in d.files:  # box QP / signed box QP (forward only)
        g = {k: torch.from_numpy(d[k]).cuda() for k in ("P", "q", "grad_x", "l_min", "l_max", "v") if k in d.files}
        xh, ith = ops.boxqp_forward(g["P"], g["q"], g["l_min"], g["l_max"], eps, mi, v=g.get("v"), return_iters=True)
        assert np.abs(npy(xh) - d["x"]).max() <= X_TOL
        assert (npy(ith) == d["iters"]).mean() >= 0.95
        if "v" not in d.files:
            out = ops.boxqp_backward(g["P"], g["q"], g["l_min"], g["l_max"], torch.from_numpy(d["x"]).cuda(), g["grad_x"],
                                     return_steps=True)
            assert np.array_equal(npy(out[4]), d["ir_steps"])
            for a, n in zip(out[:4], ("grad_P", "grad_q", "grad_l_min", "grad_l_max")):
                assert np.allclose(npy(a), d[n], rtol=1e-9, atol=1e-9 * max(1.0, np.abs(d[n]).max())), n
        return
    kind = "qcqp" if "l_n" in d.files else "qp"
    g = {k: torch.from_numpy(d[k]).cuda() for k in ("P", "q", "grad_x", "l_n", "mu") if k in d.files}
    xh, ith = hip_fwd(ops, kind, g, eps=eps, max_iter=mi)
    assert np.abs(npy(xh) - d["x"]).max() <= X_TOL
    assert (npy(ith) == d["iters"]).mean() >= 0.95
    grads, st = hip_bwd(ops, kind, g, torch.from_numpy(d["x"]).cuda())
    names = ["grad_P", "grad_q", "grad_l_n", "grad_mu"][: len(grads)]
    assert np.array_equal(npy(st), d["ir_steps"])
    for a, n in zip(grads, names):
        assert np.allclose(npy(a), d[n], rtol=1e-9, atol=1e-9 * max(1.0, np.abs(d[n]).max()))


# ---------------------------------------------------------------- autograd drop-in surface
def test_autograd_functions_match_reference_contract(oracle, ops):
    from diffqcqp_amd.qcqp import QPFn2, QCQPFn2
    d = make_problem("qcqp", 64, 8, 901)
    P, q = d["P"].cuda().requires_grad_(True), d["q"].cuda().requires_grad_(True)
    l_n, mu = d["l_n"].cuda().requires_grad_(True), d["mu"].cuda().requires_grad_(True)
    ws = torch.zeros_like(q)
    x = QCQPFn2.apply(P, q, l_n, mu, ws, 1e-7, 1000)
    assert x.shape == (64, 8, 1) and x.is_cuda
    (x * d["grad_x"].cuda()).sum().backward()
    xo, _ = oracle_fwd(oracle, "qcqp", d)
    ref = oracle_bwd(oracle, "qcqp", d, xo)
    for t, r in zip((P, q, l_n, mu), ref[:-1]):
        assert t.grad.shape == r.shape
    # the autograd path is the ops path: same numbers as calling the backward op on the saved x
    direct = ops.qcqp_backward(P.detach(), q.detach(), l_n.detach(), mu.detach(), x.detach(), d["grad_x"].cuda())
    for t, r in zip((P, q, l_n, mu), direct):
        assert torch.equal(t.grad, r)
    _, _, _, _, st = ops.qcqp_backward(P.detach(), q.detach(), l_n.detach(), mu.detach(), x.detach(),
                                       d["grad_x"].cuda(), return_steps=True)
    check_end_to_end([P.grad, q.grad, l_n.grad, mu.grad], st, ref)
    # QP, only q requires grad, warm_start has no effect and gets no grad; CPU tensors are staged via the GPU
    dq = make_problem("qp", 10, 8, 902)
    Pc, qc = dq["P"], dq["q"].clone().requires_grad_(True)
    w1 = torch.zeros(10, 8, 1, requires_grad=True)
    x1 = QPFn2.apply(Pc, qc, w1, 1e-7, 1000)
    x2 = QPFn2.apply(Pc, qc, torch.randn(10, 8, 1), 1e-7, 1000)
    assert not x1.is_cuda and torch.equal(x1, x2)
    x1.sum().backward()
    assert w1.grad is None and qc.grad is not None and qc.grad.shape == (10, 8, 1)
    cache = ops.diag_cache(qc.detach().cuda())
    cache[1].zero_()  # no verified diagonal: the backward reads P
    out = QPFn2.backward(type("C", (), {"saved_tensors": (Pc.cuda(), qc.detach().cuda(), x1.detach().cuda()) + cache,
                                        "needs_input_grad": (False, True, False, False, False, False),
                                        "home": torch.device("cpu"), "layout": 0})(), torch.ones(10, 8, 1))
    assert len(out) == 6 and out[0] is None and out[2:] == (None, None, None, None)


def test_unbatched_twins_and_module_level_api(oracle, ops):
    """qcqp_no_batch.py shapes and the `diffqcqp` module-level numpy functions
    (same names / kwargs / return shapes as pybindings.cpp:74-83), checked against the oracle's twins."""
    from diffqcqp_amd import diffqcqp as M
    from diffqcqp_amd import qcqp_no_batch as NB
    for structure in ("diag", "dense"):
        d = make_problem("qcqp", 3, 8, 950, structure)
        for b in range(3):
            P, q, g = d["P"][b].numpy(), d["q"][b].numpy(), d["grad_x"][b, :, 0].numpy()
            ln, mu = d["l_n"][b].numpy(), d["mu"][b].numpy()
            x = M.solveQP(P, q, np.zeros(8), 1e-7, 1e-7, 1000, True)
            xo = oracle.solveQP(P, q, None, 1e-7, 1e-7, 1000, True)
            assert x.shape == (8,) and np.abs(x - xo).max() < 1e-9
            bl = M.solveDerivativesQP(P, q, xo, g)
            assert bl.shape == (8,) and np.allclose(bl, oracle.solveDerivativesQP(P, q, xo, g), rtol=1e-9, atol=1e-12)
            assert np.allclose(M.solveDerivativesQP(P, q, xo, g, epsilon=1e-3), oracle.solveDerivativesQP(P, q, xo, g, 1e-3),
                               rtol=1e-9, atol=1e-12)
            xq = M.solveQCQP(P, q, ln, mu, np.zeros(8), epsilon=1e-7, max_iter=1000)
            xqo = oracle.solveQCQP(P, q, ln, mu, None, 1e-7, 1e-7, 1000)
            assert np.abs(xq - xqo).max() < 1e-9
            E1, E2, blg = M.solveDerivativesQCQP(P, q, ln, mu, xqo, g)
            E1o, E2o, blgo = oracle.solveDerivativesQCQP(P, q, ln, mu, xqo, g)
            assert E1.shape == (4, 4) and blg.shape == (12,)
            assert np.allclose(E1, E1o, rtol=1e-9, atol=1e-13) and np.allclose(E2, E2o, rtol=1e-9, atol=1e-13)
            assert np.allclose(blg, blgo, rtol=1e-9, atol=1e-12)
    # unbatched autograd twins
    d = make_problem("qcqp", 1, 8, 951, "dense")
    P = d["P"][0].clone().requires_grad_(True)
    q = d["q"][0].clone().requires_grad_(True)
    l_n, mu = d["l_n"][0].clone().requires_grad_(True), d["mu"][0].clone().requires_grad_(True)
    x = NB.QCQPFn2.apply(P, q, l_n, mu, torch.zeros(8, 1), 1e-7, 1000)
    assert x.shape == (8,)
    (x * d["grad_x"][0, :, 0]).sum().backward()
    assert P.grad.shape == (8, 8) and q.grad.shape == (8, 1) and l_n.grad.shape == (4, 1) and mu.grad.shape == (4, 1)
    xo, _ = oracle_fwd(oracle, "qcqp", d)
    assert np.abs(x.detach().numpy() - xo[0, :, 0]).max() < 1e-9
    xq = NB.QPFn2.apply(d["P"][0], q, torch.zeros(8, 1), 1e-7, 1000)
    assert xq.shape == (8,)


def test_runs_on_a_non_default_stream(oracle, ops):
    d = make_problem("qp", 300, 8, 903)
    g = dev(d)
    xo, _ = oracle_fwd(oracle, "qp", d)
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        x = ops.qp_forward(g["P"], g["q"], 1e-7, 1000)
    s.synchronize()
    assert np.abs(npy(x) - xo).max() <= X_TOL


# ---------------------------------------------------------------- full BASELINE sizes: properties
@pytest.mark.parametrize("kind", ["qp", "qcqp"])
def test_full_size_b65536_n8_properties(oracle, ops, kind):
    """configs[1]/[2] sizes.  Size-independent properties + the oracle on a 4096-problem sample."""
    B, N = 65536, 8
    d = make_problem(kind, B, N, 1002 if kind == "qp" else 1003)
    g = dev(d)
    x, it = hip_fwd(ops, kind, g)
    assert torch.isfinite(x).all() and int(it.max()) < 200
    p = torch.diagonal(g["P"], dim1=1, dim2=2)
    if kind == "qp":
        cf = torch.clamp(-g["q"][:, :, 0] / p, min=0)
        err = (x[:, :, 0] - cf).abs().amax(1)
        assert float(err.median()) < 1e-6 and float(err.max()) < 1e-3 and float(x.min()) >= 0.0
    else:
        r = (g["l_n"] * g["mu"])[:, :, 0]
        nrm = torch.sqrt(x[:, 0::2, 0] ** 2 + x[:, 1::2, 0] ** 2)
        assert float((nrm - r).max()) < 1e-12
    # permutation equivariance: problems are independent, tiles must not leak into each other
    perm = torch.randperm(B, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    gp = {k: v[perm].contiguous() for k, v in g.items()}
    xp, _ = hip_fwd(ops, kind, gp)
    assert torch.equal(xp, x[perm])
    grads, st = hip_bwd(ops, kind, g, x)
    gradsp, _ = hip_bwd(ops, kind, gp, xp)
    for a, b in zip(grads, gradsp):
        assert torch.equal(b, a[perm])
    # grad_P is the outer product of grad_q and x (qcqp.py:49-51): exact by construction
    assert torch.equal(grads[0], grads[1] * x.transpose(1, 2))
    # oracle on a sample
    idx = torch.arange(0, B, 16)
    ds = {k: v[idx].contiguous() for k, v in d.items()}
    xo, ito = oracle_fwd(oracle, kind, ds)
    check_forward(x[idx.cuda()], it[idx.cuda()], xo, ito)
    ref = oracle_bwd(oracle, kind, ds, xo)
    gs, sts = hip_bwd(ops, kind, dev(ds), torch.from_numpy(xo).cuda())
    check_backward_exact(gs, sts, ref)


def test_full_size_config4_per_gpu_shard(oracle, ops):
    """configs[3]: B=262144 N=32 over 8 GPUs = 32768 problems per GPU; one shard here."""
    B, N = 32768, 32
    d = make_problem("qp", B, N, 1004)
    g = dev(d)
    x, it = hip_fwd(ops, "qp", g)
    p = torch.diagonal(g["P"], dim1=1, dim2=2)
    cf = torch.clamp(-g["q"][:, :, 0] / p, min=0)
    assert float((x[:, :, 0] - cf).abs().amax(1).median()) < 1e-6
    grads, st = hip_bwd(ops, "qp", g, x)
    assert torch.equal(grads[0], grads[1] * x.transpose(1, 2))
    idx = torch.arange(0, B, 64)
    ds = {k: v[idx].contiguous() for k, v in d.items()}
    xo, ito = oracle_fwd(oracle, "qp", ds)
    check_forward(x[idx.cuda()], it[idx.cuda()], xo, ito)
    gs, sts = hip_bwd(ops, "qp", dev(ds), torch.from_numpy(xo).cuda())
    check_backward_exact(gs, sts, oracle_bwd(oracle, "qp", ds, xo))


def test_full_size_config4_whole_batch(oracle, ops):
    """configs[3] at its FULL size on one GPU: B=262144, N=32, diagonal P in the (B,32,32) layout (2.1 GB of P,
    generated on the device), QP forward + backward.  Size-independent properties over the whole batch: closed form
    x* = max(-q/p, 0) within the solver's accuracy, feasibility, iteration counts inside the family's range, the
    backward identity dl_i = p_i g_i / (p_i^2 + 1e-7) on the inactive set, grad_P = grad_q (x) x bit for bit,
    equivariance under a permutation of the batch; plus the oracle on a sample spread over the batch."""
    B, N = 262144, 32
    gen = torch.Generator(device="cuda").manual_seed(1004)
    p = torch.rand(B, N, generator=gen, dtype=torch.float64, device="cuda") + 0.1
    P = torch.diag_embed(p).contiguous()
    q = 2 * torch.rand(B, N, 1, generator=gen, dtype=torch.float64, device="cuda") - 1
    gx = torch.randn(B, N, 1, generator=gen, dtype=torch.float64, device="cuda")
    cache = ops.diag_cache(q)
    x, it = ops.qp_forward(P, q, 1e-7, 1000, return_iters=True, cache=cache)
    assert bool(cache[1].all())                                    # every tile verified diagonal
    assert bool((x >= 0).all()) and 10 <= int(it.min()) and int(it.max()) < 60
    cf = torch.clamp(-q[:, :, 0] / p, min=0)
    err = (x[:, :, 0] - cf).abs().amax(1)
    assert float(err.median()) < 1e-6 and float(err.max()) < 1e-3   # the reference's own accuracy
    gP, gq, st = ops.qp_backward(P, q, x, gx, return_steps=True, cache=cache)
    assert int(st.max()) == 1 and int(st.min()) == 1
    assert torch.equal(gP, gq * x.transpose(1, 2))                  # qcqp.py:48-51
    inactive = (x[:, :, 0] > 1e-10) | (-(p * x[:, :, 0] + q[:, :, 0]) >= -1e-10)
    dl = torch.where(inactive, p * gx[:, :, 0] / (p * p + 1e-7), torch.zeros_like(p))
    assert float((gq[:, :, 0] + dl).abs().max()) < 1e-9
    # permutation equivariance on a slice (problems are independent: no cross-problem term anywhere)
    perm = torch.randperm(4096, generator=torch.Generator().manual_seed(3)).cuda()
    sl = slice(100000, 104096)
    xp, itp = ops.qp_forward(P[sl][perm].contiguous(), q[sl][perm].contiguous(), 1e-7, 1000, return_iters=True)
    # (the smaller batch runs with fewer lanes per problem: same trajectory, tree sums associated differently)
    assert torch.equal(itp, it[sl][perm]) and float((xp - x[sl][perm]).abs().max()) < 1e-12
    idx = torch.arange(0, B, 1024, device="cuda")
    d = {"P": P[idx].cpu(), "q": q[idx].cpu(), "grad_x": gx[idx].cpu()}
    xo, ito = oracle_fwd(oracle, "qp", d)
    check_forward(x[idx], it[idx], xo, ito)
    gs, sts = hip_bwd(ops, "qp", dev(d), torch.from_numpy(xo).cuda())
    check_backward_exact(gs, sts, oracle_bwd(oracle, "qp", d, xo))


def test_full_size_config5_dense_n64(oracle, ops):
    """configs[4] at its full size: B=65536, N=64, dense P = S S^T/64 + 0.1 I (generated on the device), QP
    forward + backward.  Size-independent properties: feasibility, the natural KKT residual |min(x, Px+q)|,
    grad_P = grad_q (x) x; plus the oracle on a sample of the batch."""
    B, N = 65536, 64
    gen = torch.Generator(device="cuda").manual_seed(1005)
    S = torch.rand(B, N, N, generator=gen, dtype=torch.float64, device="cuda")
    P = torch.bmm(S, S.transpose(1, 2)) / N
    del S
    P.diagonal(dim1=1, dim2=2).add_(0.1)
    q = 2 * torch.rand(B, N, 1, generator=gen, dtype=torch.float64, device="cuda") - 1
    gx = torch.randn(B, N, 1, generator=gen, dtype=torch.float64, device="cuda")
    x, it = ops.qp_forward(P, q, 1e-7, 1000, return_iters=True)
    assert int(it.max()) < 1000 and bool((x >= 0).all())
    r = torch.bmm(P, x) + q
    # Reference quirk (Solver.cpp:88): the loop stops on the dual residual rho*|l_2 - l_2_pred| alone, so an
    # iterate that the projection maps to the same point twice in a row (typically x = 0) is returned whatever
    # its primal residual is.  A handful of problems in 65536 end that way; they must be exactly what the
    # oracle returns, and every other problem must satisfy the KKT conditions.
    nat = torch.minimum(x, r).abs().amax(dim=(1, 2))               # complementarity / dual feasibility
    odd = torch.nonzero(nat > 1e-5)[:, 0]
    assert odd.numel() <= B // 1000
    if odd.numel():
        dq = {"P": P[odd].cpu(), "q": q[odd].cpu()}
        xq, itq = oracle_fwd(oracle, "qp", dq)
        assert np.array_equal(npy(it[odd]), itq) and np.abs(npy(x[odd]) - xq).max() <= 1e-9
    gP, gq, st = ops.qp_backward(P, q, x, gx, return_steps=True)
    assert torch.equal(gP, gq * x.transpose(1, 2))                  # qcqp.py:48-51
    assert int(st.max()) <= 10 and bool(torch.isfinite(gq).all())
    idx = torch.arange(0, B, 256, device="cuda")                    # 256 problems of the batch against the oracle
    d = {"P": P[idx].cpu(), "q": q[idx].cpu(), "grad_x": gx[idx].cpu()}
    xo, ito = oracle_fwd(oracle, "qp", d)
    # the matrix-core kernel re-associates the sums of the inverse (block Gauss-Jordan instead of LLT + substitutions):
    # x within 1e-6 everywhere, the iteration count equal to the oracle's on >= 99 % of the sample (measured: 1024 of 1024 equal,
    # max |dx| 1.3e-14: profiles/r07d_cfg5_match.txt) and never more than 2 apart
    check_forward(x[idx], it[idx], xo, ito, min_match=0.99)
    assert np.abs(npy(it[idx]).astype(np.int64) - ito).max() <= 2
    gs, sts = hip_bwd(ops, "qp", dev(d), torch.from_numpy(xo).cuda())
    check_backward_exact(gs, sts, oracle_bwd(oracle, "qp", d, xo), exact=False)


@pytest.mark.parametrize("kind", ["qp", "qcqp"])
@pytest.mark.parametrize("structure", ["dense", "one_in_1000"])
def test_full_size_b65536_n8_dense_p_through_auto(oracle, ops, kind, structure):
    """A real contact problem's P is dense.  At the bench's batch size, through DQQ_P_AUTO (what QPFn2 / QCQPFn2 pass):
    a fully dense batch and a diagonal batch with one dense problem in 1000, forward + backward.  The non-diagonal
    tiles go through the work-list to the kernels DQQ_P_DENSE launches directly, so both routes must return the same
    numbers for them; round 2 solved such tiles one problem per wave inside the fused backward for exactly
    32 Ki <= B <= 128 Ki (10-100x slower).  A sample is checked against the oracle."""
    B, N = 65536, 8
    dd = make_problem(kind, B, N, 6100, "dense")
    d = dict(dd)
    if structure == "one_in_1000":
        dg = make_problem(kind, B, N, 6100, "diag")
        sel = (torch.arange(B) % 1000 == 1).view(B, 1, 1)
        d["P"] = torch.where(sel, dd["P"], dg["P"]).contiguous()
    g = dev(d)
    xa, ita = hip_fwd(ops, kind, g, layout=0)
    xd, itd = hip_fwd(ops, kind, g, layout=1)
    nd = torch.arange(B) % 1000 == 1 if structure == "one_in_1000" else torch.ones(B, dtype=torch.bool)
    assert np.abs(npy(xa) - npy(xd)).max() <= 1e-9 and (npy(ita) == npy(itd)).mean() >= 0.9999
    ga, sa = hip_bwd(ops, kind, g, xd, layout=0)
    gd, sd = hip_bwd(ops, kind, g, xd, layout=1)
    assert np.array_equal(npy(sa), npy(sd))
    for a, b in zip(ga, gd):
        a, b = npy(a), npy(b)
        assert np.array_equal(a[nd.numpy()], b[nd.numpy()]), "non-diagonal problems: AUTO and DENSE differ"
        assert np.allclose(a, b, rtol=1e-9, atol=1e-12)
    idx = np.concatenate([np.nonzero(nd.numpy())[0][:256], np.arange(0, B, 257)[:256]])
    sub = {k: v[idx] for k, v in d.items()}
    xo, ito = oracle_fwd(oracle, kind, sub)
    assert np.abs(npy(xa)[idx] - xo).max() <= X_TOL and (npy(ita)[idx] == ito).mean() >= 0.99
    ref = oracle_bwd(oracle, kind, sub, npy(xd)[idx])
    assert np.array_equal(npy(sa)[idx], ref[-1])
    for a, b in zip(ga, ref[:-1]):
        assert np.allclose(npy(a)[idx], b, rtol=1e-9, atol=1e-9 * max(1.0, np.abs(b).max()))


@pytest.mark.parametrize("kind", ["qp", "qcqp"])
@pytest.mark.parametrize("N,B,pattern", [(32, 6000, "sparse"), (64, 3001, "sparse"), (32, 4099, "one_segment"),
                                         (64, 2500, "one_segment"), (64, 4097, "dense"), (32, 3, "dense")])
def test_segmented_worklist_uneven_segments(ops, kind, N, B, pattern):
    """N >= 32: the fast kernels queue non-diagonal tiles on a SEGMENTED list (workgroup i -> segment i mod 32, one
    counter per segment, csrc/launch.h) that the general kernels read through a scan of the 32 counters.  Batches whose
    dense problems leave the segments uneven, most of them empty ("one_segment": only workgroups of one residue class
    see a dense problem) or all full; forward + backward through DQQ_P_AUTO, twice on the same workspace, against the
    same batch declared DQQ_P_DENSE (the same general kernels without a list).  Both fast-kernel shapes: four waves
    per workgroup (one aggregated push) and one (a push per wave)."""
    from diffqcqp_amd import _capi
    dd, dg = make_problem(kind, B, N, 6700 + N, "dense"), make_problem(kind, B, N, 6700 + N, "diag")
    per_wg_fwd = (64 // (N // 2)) * 4           # problems per workgroup of the forward (LPP = N/2, four waves)
    idx = torch.arange(B)
    if pattern == "sparse":
        sel = (idx * 2654435761 % 4294967296) % 23 == 0
    elif pattern == "one_segment":
        sel = ((idx // per_wg_fwd) % 32 == 5) & (idx % 3 == 0)
    else:
        sel = torch.ones(B, dtype=torch.bool)
    d = dict(dd)
    d["P"] = torch.where(sel.view(B, 1, 1), dd["P"], dg["P"]).contiguous()
    g = dev(d)
    xd, itd = hip_fwd(ops, kind, g, layout=1)
    gd, sd = hip_bwd(ops, kind, g, xd, layout=1)
    nd = sel.numpy()
    try:
        for wpb in (0, 1, 0):
            knob("wpb", wpb)
            xa, ita = hip_fwd(ops, kind, g, layout=0)
            ga, sa = hip_bwd(ops, kind, g, xd, layout=0)
            torch.cuda.synchronize()
            for ws in ops._workspaces.values():
                assert header_is_clean(ws)
            # queued problems: the very kernel DQQ_P_DENSE launches -> the same bits
            assert np.array_equal(npy(xa)[nd], npy(xd)[nd]) and np.array_equal(npy(ita)[nd], npy(itd)[nd])
            assert np.array_equal(npy(sa)[nd], npy(sd)[nd])
            for a, b in zip(ga, gd):
                assert np.array_equal(npy(a)[nd], npy(b)[nd])
            # the rest took the diagonal arithmetic (or, as neighbours of a dense problem in its tile, the list)
            assert np.abs(npy(xa) - npy(xd)).max() <= 1e-9 and (npy(ita) == npy(itd)).mean() >= 0.99
            same = (npy(sa) == npy(sd)).reshape(B, -1).all(1)
            assert same.mean() > 0.9
            for a, b in zip(ga, gd):
                a, b = npy(a), npy(b)
                assert np.abs(a - b)[same].max() <= 1e-6 * max(1.0, np.abs(b).max())
    finally:
        knob("wpb", 0)


@pytest.mark.parametrize("N,B", [(18, 300), (20, 129), (24, 500), (26, 64), (30, 257), (32, 1024), (34, 96), (40, 200),
                                 (46, 64), (48, 257), (50, 64), (56, 130), (62, 33), (64, 300)])
def test_qcqp_backward_wave_kernel_16_to_32(oracle, ops, N, B):
    """QCQP backward for a dense P, 16 < N <= 64: one wave per problem, the factor of the (N/2 + N)-unknown system in
    registers (N <= 32: bwd_wave_qcqp.hip, the system matrix too; beyond: bwd_wave_qcqp_big.hip, the matrix streamed;
    block Cholesky).  Against the oracle on the oracle's x (tolerances: REASSOC_TOL, the
    reference's own evaluation-order noise) and against the LDS wave kernel in the reference's summation order
    (wave_qcqp_bwd = 0); refinement exits that differ are checked against the reference formula at the kernel's own
    exit (orc_set_force_ir_steps)."""
    from diffqcqp_amd import _capi
    d = make_problem("qcqp", B, N, 6400 + N, "dense")
    g = dev(d)
    xo, _ = oracle_fwd(oracle, "qcqp", d)
    xs = torch.from_numpy(xo).cuda()
    ref = oracle_bwd(oracle, "qcqp", d, xo)
    for layout in (_capi.P_DENSE, _capi.P_AUTO):
        duals = (torch.empty(B, N // 2, 1, device="cuda", dtype=torch.float64),
                 torch.empty(B, N // 2, 1, device="cuda", dtype=torch.float64))
        gP, gq, gl, gm, st = ops.qcqp_backward(g["P"], g["q"], g["l_n"], g["mu"], xs, g["grad_x"], layout=layout,
                                               return_steps=True, duals=duals)
        check_backward_reassociated(oracle, "qcqp", d, xo, (gP, gq, gl, gm), st, ref)
        assert torch.isfinite(duals[0]).all() and torch.isfinite(duals[1]).all()
    # the kernel in the reference's summation order agrees (bit-exact with the oracle on identical x)
    knob("wave_qcqp_bwd", 0)
    try:
        grads, st2 = hip_bwd(ops, "qcqp", g, xs, layout=_capi.P_DENSE)
        check_backward_exact(grads, st2, ref, exact=False)
    finally:
        knob("wave_qcqp_bwd", 1)
