"""The in-kernel general solve of the fused fast path (csrc/group_dense.h): a non-diagonal tile met by
fwd_diag_kernel through DQQ_P_AUTO is solved by the lanes that would have taken its diagonal twin -- every lanes-
per-problem choice, all four solvers, dense / mixed / non-symmetric P, ragged batches, tight tolerances -- against
the oracle: x within 1e-6 (median 1e-11), iteration counts equal."""
import numpy as np
import pytest
import torch

from conftest import make_problem, knob
from test_gpu_parity import _box_fwd, check_forward, dev, hip_fwd, npy, oracle_fwd, ops  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.fixture()
def fused(ops):
    from diffqcqp_amd import _capi
    knob("fuse_fallback", 1)
    yield _capi
    knob("fuse_fallback", -1)
    knob("fwd_lpp", 0)


@pytest.mark.parametrize("kind", ["qp", "qcqp"])
@pytest.mark.parametrize("N,lpp", [(2, 1), (4, 1), (4, 2), (8, 2), (8, 4)])
@pytest.mark.parametrize("structure", ["dense", "mixed"])
def test_fused_general_tiles_match_oracle(oracle, ops, fused, kind, N, lpp, structure):
    knob("fwd_lpp", lpp)
    for B in (1, 37, 1029):
        d = make_problem(kind, B, N, 5100 + N + B, structure)
        xo, ito = oracle_fwd(oracle, kind, d)
        xh, ith = hip_fwd(ops, kind, dev(d))
        check_forward(xh, ith, xo, ito, min_match=0.99 if B > 100 else 0.95)


@pytest.mark.parametrize("kind", ["qp", "qcqp"])
def test_fused_general_tiles_tight_tolerance_and_budget(oracle, ops, fused, kind):
    d = make_problem(kind, 300, 8, 5200, "dense")
    g = dev(d)
    for eps, max_iter in ((1e-10, 1000), (1e-7, 7), (1e-7, 1)):
        xo, ito = oracle_fwd(oracle, kind, d, eps=eps, max_iter=max_iter)
        xh, ith = hip_fwd(ops, kind, g, eps=eps, max_iter=max_iter)
        check_forward(xh, ith, xo, ito, min_match=0.98)


@pytest.mark.parametrize("kind", ["qp", "qcqp"])
def test_fused_general_tiles_non_symmetric_p(oracle, ops, fused, kind):
    """llt() reads the lower triangle only (Solver.cpp:76), the power iteration the full matrix (:51)."""
    B, N = 200, 8
    d = make_problem(kind, B, N, 5300, "dense")
    gen = torch.Generator().manual_seed(5300)
    d["P"] = (d["P"] + torch.triu(torch.rand(B, N, N, generator=gen, dtype=torch.float64), diagonal=1) * 0.05).contiguous()
    xo, ito = oracle_fwd(oracle, kind, d)
    xh, ith = hip_fwd(ops, kind, dev(d))
    check_forward(xh, ith, xo, ito, min_match=0.98)


@pytest.mark.parametrize("kind", ["box", "sbox"])
@pytest.mark.parametrize("N,lpp", [(2, 1), (4, 1), (4, 2), (8, 4)])
def test_fused_general_tiles_box_solvers(oracle, ops, fused, kind, N, lpp):
    knob("fwd_lpp", lpp)
    d = make_problem(kind, 333, N, 5400 + N, "dense")
    xo, ito, xh, ith = _box_fwd(oracle, ops, kind, d)
    check_forward(xh, ith, xo, ito, min_match=0.98)


def test_fused_general_tile_with_a_singular_problem(oracle, ops, fused):
    """A matrix whose shifted version is not positive definite poisons its own problem only (NaN, as the reference's
    unchecked LLT would, Solver.cpp:76); its tile neighbours are unaffected."""
    B, N = 64, 8
    d = make_problem("qp", B, N, 5500, "dense")
    xo, _ = oracle_fwd(oracle, "qp", d)
    d["P"][5] = -d["P"][5]
    xh, _ = hip_fwd(ops, "qp", dev(d))
    xh = npy(xh)
    assert np.isnan(xh[5]).all()
    keep = np.arange(B) != 5
    assert np.abs(xh[keep] - xo[keep]).max() < 1e-9


def test_dense_layout_small_batches_take_the_group_kernel(oracle, ops):
    """DQQ_P_DENSE at N = 8: same answers from whichever kernel the batch size selects."""
    from diffqcqp_amd import _capi
    for kind in ("qp", "qcqp"):
        for B in (50, 3000, 40000):
            d = make_problem(kind, B, 8, 5600 + B, "dense")
            n = min(B, 2000)
            dn = {k: v[:n] for k, v in d.items()}
            xo, ito = oracle_fwd(oracle, kind, dn)
            xh, ith = hip_fwd(ops, kind, dev(d), layout=_capi.P_DENSE)
            check_forward(xh[:n], ith[:n], xo, ito, min_match=0.99)


@pytest.mark.parametrize("kind", ["qp", "qcqp", "box", "sbox"])
@pytest.mark.parametrize("N,lpp", [(2, 1), (4, 2), (8, 2), (8, 4)])
def test_group_solve_deferred_refactorisation_is_bit_identical(ops, fused, kind, N, lpp):
    """The group solve defers the Gauss-Jordan sweep after a rho update so that one sweep serves every problem of the
    wave that changed rho over several trips (option lane_defer, csrc/group_dense.h).  A problem only sits out
    meanwhile: x and the iteration counts must not depend on the setting -- through DQQ_P_AUTO (tiles of a mixed batch
    met by the fused kernel, the two-pass narrow mapping at N = 8 on two lanes per problem) and through DQQ_P_DENSE
    (the solve's own mapping), budgets that run out mid-solve included."""
    knob("fwd_lpp", lpp)
    B = 1500
    g = dev(make_problem(kind, B, N, 9300 + N, "mixed"))
    gd = dev(make_problem(kind, B, N, 9400 + N, "dense"))

    def fwd(gg, layout, eps, max_iter):
        if kind in ("box", "sbox"):
            return ops.boxqp_forward(gg["P"], gg["q"], gg["l_min"], gg["l_max"], eps, max_iter, v=gg.get("v"), layout=layout,
                                     return_iters=True)
        return hip_fwd(ops, kind, gg, layout=layout, eps=eps, max_iter=max_iter)

    try:
        for gg, layout in ((g, 0), (gd, 0), (gd, 1)):
            for eps, max_iter in ((1e-7, 1000), (1e-10, 1000), (1e-7, 9), (1e-7, 1)):
                knob("lane_defer", 1)
                x1, it1 = fwd(gg, layout, eps, max_iter)
                for defer in (0, 2, 4, 6, 64):
                    knob("lane_defer", defer)
                    xd, itd = fwd(gg, layout, eps, max_iter)
                    assert torch.equal(it1, itd), (layout, eps, max_iter, defer)
                    assert torch.equal(torch.nan_to_num(x1, nan=12345.0), torch.nan_to_num(xd, nan=12345.0))
    finally:
        knob("lane_defer", 0)


@pytest.mark.parametrize("kind", ["qp", "qcqp"])
def test_declared_dense_n8_is_the_same_solve_on_either_side_of_the_batch_size_rule(ops, kind):
    """ADVICE r5.  Which kernel serves a DQQ_P_DENSE N = 8 batch is a function of B (B <= 32768: the group solve inside
    fwd_diag_kernel, four lanes per problem; above: fwd_lane_dense_kernel, a lane per problem) -- two kernels, one algorithm and
    ONE projection rule (|l|^2 > r |r| on the fused squares in both).  The same problems through both: iteration counts equal
    on >= 99.9 %, x within 1e-9 where they are (the x-updates differ in operation order -- explicit inverse in registers
    against factor + substitutions -- so bits may differ; the branch a projection takes at the cone must not)."""
    from diffqcqp_amd import _capi
    small, big = 32768, 40000
    d = make_problem(kind, big, 8, 5900, "dense")
    g = dev(d)
    head = {k: v[:small].contiguous() for k, v in g.items()}
    xa, ia = hip_fwd(ops, kind, head, layout=_capi.P_DENSE)      # group solve
    xb, ib = hip_fwd(ops, kind, g, layout=_capi.P_DENSE)         # lane per problem
    ia, ib = npy(ia), npy(ib)[:small]
    same = ia == ib
    assert same.mean() >= 0.999, same.mean()
    assert np.abs(npy(xa) - npy(xb)[:small])[same].max() < 1e-9
    assert np.abs(ia.astype(int) - ib.astype(int)).max() <= 2
