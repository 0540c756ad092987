"""Re-spreading the tail of an N = 8 forward tile onto more lanes per problem (csrc/admm_core.h) must not change a single bit.

The problems of a tile stop at different iterations; once few are left they move onto four, then eight lanes per problem.
Where a problem runs does not enter its arithmetic, so x and the iteration counts must equal those of the kernel that never
moves anything -- for full and ragged batches, for iteration budgets that end before / at / after a move, for problems that
fail (NaN), and next to tiles that are not diagonal.  (The thresholds are knobs of the developer build, csrc/tuning.h: these
tests skip on the shipped library, whose thresholds are compile-time constants.)"""
import numpy as np
import pytest
import torch

from conftest import make_problem, knob

pytestmark = pytest.mark.gpu


def _run(kind, d, eps, max_iter, _unused=0, layout=0):
    from diffqcqp_amd import _capi, ops
    if True:
        B, N = d["q"].shape[0], d["q"].shape[1]
        x = torch.full((B, N, 1), float("nan"), dtype=torch.float64, device="cuda")
        P = d["P"] if layout != 2 else torch.diagonal(d["P"], dim1=1, dim2=2).contiguous()
        if kind == "qp":
            _, it = ops.qp_forward(P, d["q"], eps, max_iter, layout=layout, out=x, return_iters=True)
        else:
            _, it = ops.qcqp_forward(P, d["q"], d["l_n"], d["mu"], eps, max_iter, layout=layout, out=x,
                                     return_iters=True)
        torch.cuda.synchronize()
        return x.cpu().numpy(), it.cpu().numpy()


def _run_respread(kind, d, eps, max_iter, at, lpp=2, at2=0, from2=0):
    """from2: the iteration from which the second move applies ("fwd_respread2_from"; shipped: 48 -- these well-conditioned
    problems are done by then, so the tests set 0 to drive the eight-lane stage, and 20 / 48 for the gate itself)."""
    from diffqcqp_amd import _capi
    knob("fwd_respread", at)
    knob("fwd_respread2", at2)
    knob("fwd_respread2_from", from2)
    knob("fwd_lpp", lpp)
    try:
        return _run(kind, d, eps, max_iter, 0)
    finally:
        knob("fwd_respread", 16)  # the defaults
        knob("fwd_respread2", 8)
        knob("fwd_respread2_from", 48)
        knob("fwd_lpp", 0)


@pytest.mark.parametrize("kind", ["qp", "qcqp"])
@pytest.mark.parametrize("B", [1, 17, 31, 33, 1000, 4099, 65536, 70001])
def test_respread_bit_identical(kind, B):
    """N = 8 on two lanes per problem: once at most `fwd_respread` problems of a wave are still iterating they move
    onto four lanes per problem (csrc/admm_core.h admm_fwd_diag_respread).  Neither the moment of the move nor the
    move itself may change a bit of x or an iteration count (the reference solves every problem on its own,
    pybindings.cpp:43-61) -- for every threshold, for iteration budgets that end before / at / after the move, for
    problems that fail (NaN), and next to tiles that are not diagonal."""
    d = {k: v.cuda() for k, v in make_problem(kind, B, 8, 777 + B).items()}
    d["P"][3::41] *= -1.0  # non-convex problems: the failure signalling must travel with the problem
    for eps, max_iter in ((1e-7, 1000), (1e-7, 1), (1e-7, 2), (1e-7, 16), (1e-7, 17), (1e-7, 22), (1e-12, 1000)):
        xa, ia = _run_respread(kind, d, eps, max_iter, 0)
        assert (ia >= 1).all() and (ia <= max_iter).all()
        for at in (1, 7, 16):
            xb, ib = _run_respread(kind, d, eps, max_iter, at)
            assert np.array_equal(ia, ib), (eps, max_iter, at)
            assert np.array_equal(xa, xb, equal_nan=True), (eps, max_iter, at)
        # the second move: the last survivors onto eight lanes per problem, one coordinate per lane ("fwd_respread2")
        for at, at2 in ((16, 8), (16, 1), (16, 3), (7, 7), (2, 8), (12, 5)):
            xb, ib = _run_respread(kind, d, eps, max_iter, at, at2=at2)
            assert np.array_equal(ia, ib), (eps, max_iter, at, at2)
            assert np.array_equal(xa, xb, equal_nan=True), (eps, max_iter, at, at2)
        # ... gated by the iteration count ("fwd_respread2_from": the shipped 48, and a gate that falls inside these solves)
        for from2 in (20, 48):
            xb, ib = _run_respread(kind, d, eps, max_iter, 16, at2=8, from2=from2)
            assert np.array_equal(ia, ib) and np.array_equal(xa, xb, equal_nan=True), (eps, max_iter, from2)
    assert np.isnan(xa[3::41]).all() and np.isfinite(np.delete(xa, np.s_[3::41], axis=0)).all()


@pytest.mark.parametrize("kind", ["qp", "qcqp"])
def test_respread_against_oracle_and_with_dense_tiles(kind):
    from oracle import oracle
    B = 3000
    d0 = make_problem(kind, B, 8, 31, structure="dense")
    P = torch.diag_embed(torch.diagonal(d0["P"], dim1=1, dim2=2)).contiguous()
    P[7::500] = d0["P"][7::500]  # a few tiles are not diagonal: solved by the general routine, untouched by the move
    d0["P"] = P
    d = {k: v.cuda() for k, v in d0.items()}
    xa, ia = _run_respread(kind, d, 1e-7, 1000, 0)
    xb, ib = _run_respread(kind, d, 1e-7, 1000, 16, at2=8)
    assert np.array_equal(ia, ib) and np.array_equal(xa, xb)
    n = 600
    Pn, q = d0["P"][:n].numpy(), d0["q"][:n].numpy()
    if kind == "qp":
        xo, io = oracle.qp_fwd_batch(Pn, q, 1e-7, 1000)
    else:
        xo, io = oracle.qcqp_fwd_batch(Pn, q, d0["l_n"][:n].numpy(), d0["mu"][:n].numpy(), 1e-7, 1000)
    assert (io == ib[:n]).mean() >= 0.999
    same = io == ib[:n]
    assert np.abs(xo - xb[:n])[same].max() < 1e-6
