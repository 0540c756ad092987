// admm_core.h -- per-problem ADMM forward solve for a DIAGONAL P.
//
// Restates Solver::solveQP (reference qcqplib/Solver.cpp:61-123) and
// Solver::solveQCQP (Solver.cpp:521-582, prox_circle :505-519, radius product
// pybindings.cpp:57) for the case where P -- and therefore P + (rho+mu) I and
// its inverse -- is diagonal, so the Cholesky factorisation + explicit inverse
// the reference performs at every rho update (Solver.cpp:76-77, 100-101,
// 114-115) collapses to E reciprocals.  Same update order, same constants,
// same rho / tau / cpt state machine, same stopping tests; the shifted diagonal
// M is ACCUMULATED (M += rho*(tau-1)), not recomputed, as the reference does.
//
// A problem of dimension n is held by a group G of LPP lanes, E = n/LPP
// consecutive coordinates per lane (E even, so a QCQP contact pair never
// straddles two lanes).  All lanes of a group hold bit-identical copies of the
// scalar state (rho, tau, cpt, ...) because G's reductions are symmetric.
//
// Deliberate ulp-level departures from the dense reference arithmetic (all far
// inside the 1e-6 parity tolerance; the trajectory -- rho schedule, iteration
// count -- is unchanged, tests/ check that): closed-form power iteration (below);
// 1/M instead of (1/sqrt(M))/sqrt(M);
// u*(1/rho) instead of u/rho;
// compiler FMA contraction; group (tree) sums when LPP > 1; 1-ulp reciprocal /
// reciprocal-square-root (common.h fast_rcp / fast_rsqrt) in place of IEEE divide
// and sqrt in the power iteration, the disk projection and the rho updates (the E
// reciprocals of a rho update from one reciprocal of their product, 1/rho and
// 1/tau carried along by products: ~3 roundings instead of 1); the QCQP's
// primal stop test compared in squared form (no square root).
// Failure signalling: the reference's LLT of a non-positive shifted diagonal
// yields NaNs (Solver.cpp:76, never checked); here a non-positive M or a
// non-finite rho poisons the output with NaN explicitly.
#pragma once

#include "common.h"

namespace dqq {

// KIND 0 = QP (x >= 0), 1 = QCQP (per-contact disk of radius rad[c]), 2 = box QP (lo <= x <= hi,
// Solver::solveBoxQP, Solver.cpp:198-261), 3 = signed box QP (box and sg o x <= 0, sg = sign(v),
// Solver::solveSignedBoxQP, :374-439).  The box solvers are the QP loop with another projection line
// (:219-220 / :396-398): 10 power steps, both taus damped, dual-residual-only stop.
// p, q: this lane's E coordinates; rad: this lane's E/2 radii (KIND 1); lo, hi, sg: this lane's E bounds and
// signs (KIND 2, 3; may be null otherwise).
// valid = false: the lane only keeps the wave's control flow company.
// Returns the number of ADMM iterations executed (Solver.cpp:79 / :538 loop).
// Reciprocals of E positive numbers from ONE reciprocal (of their product) and 3(E-1) multiplications: 14 instead
// of 20 instructions at E = 4.  Each result carries ~3 roundings instead of 1.  A product that leaves the double
// range (entries beyond ~1e75) or a non-positive factor falls back to one reciprocal per entry.
template <int E>
DQQ_HD void rcp_all(const double (&m)[E], double (&inv)[E])
{
    double pre[E];
    pre[0] = m[0];
#pragma unroll
    for (int e = 1; e < E; ++e) pre[e] = pre[e - 1] * m[e];
    if (pre[E - 1] > 1e-280 && pre[E - 1] < 1e280) {
        double r = fast_rcp(pre[E - 1]);
#pragma unroll
        for (int e = E - 1; e > 0; --e) {
            inv[e] = r * pre[e - 1];
            r = r * m[e];
        }
        inv[0] = r;
    } else {
#pragma unroll
        for (int e = 0; e < E; ++e) inv[e] = fast_rcp(m[e]);
    }
}

template <int KIND, int E, class G>
DQQ_HD int admm_fwd_diag(const double (&p)[E], const double (&q)[E], const double* rad, int n, double eps,
                         double mu, int max_iter, int adaptive, bool valid, double (&x)[E],
                         const double* lo = nullptr, const double* hi = nullptr, const double* sg = nullptr)
{
    constexpr bool QP_LIKE = (KIND != 1);
    static_assert(E % 2 == 0, "E must be even");
#include "admm_diag_prologue.inc"

    // A per-lane loop: a lane leaves it when its problem stops (all lanes of a problem decide alike: the group
    // reductions are symmetric) and keeps its state in place under the execution mask -- as a wave-uniform loop
    // around a `done` flag the same code carried a dozen register copies per iteration.
    int rho_up = 0, cpt = 0, iters = 0;
    DQQ_TL(3);
    if (valid) {
        for (int it = 0; it < max_iter; ++it) {
#define DQQ_ADMM_ON_STOP break
#include "admm_diag_body.inc"
#undef DQQ_ADMM_ON_STOP
        }
    }
    DQQ_TL(4);
    bad = G::max(bad ? 1.0 : 0.0) > 0.0;
#pragma unroll
    for (int e = 0; e < E; ++e) x[e] = bad ? NAN : l2[e];
    return iters;
}

} // namespace dqq
