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
    double M[E], Minv[E], qp[E], l2[E], u[E];

    // ---- power_iteration, Solver.cpp:46-59 (K = 10 steps for QP :71, 100 for QCQP :530).
    // For a diagonal P the K normalised steps from the uniform start vector give v ~ p^K, so the
    // Rayleigh quotient the reference returns is  L = sum p^(2K+1) / sum p^(2K)  (the per-step
    // normalisations only rescale v).  Evaluated by repeated squaring on p scaled by an exact power
    // of two so that nothing overflows; ~1e-15 relative to the iterated value.
    double L;
    {
        double m = 0.0;
#pragma unroll
        for (int e = 0; e < E; ++e) m = fmax(m, fabs(p[e]));
        m = G::max(m);
        if (m > 0.0 && m < 1.79e308) {
            int k;
            (void)frexp(m, &k);
            double num = 0.0, den = 0.0;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const double s1 = ldexp(p[e], -k); // |s1| < 1, the largest is >= 0.5
                const double s2 = s1 * s1, s4 = s2 * s2, s8 = s4 * s4, s16 = s8 * s8;
                double a; // s1^(2K)
                if (QP_LIKE) {
                    a = s16 * s4;                                   // ^20
                } else {
                    const double s32 = s16 * s16, s64 = s32 * s32, s128 = s64 * s64;
                    a = (s128 * s64) * s8;                          // ^200
                }
                den += a;
                num += a * s1;
            }
            num = G::sum(num);
            den = G::sum(den);
            L = ldexp(num / den, k);
        } else {
            L = (m > 0.0) ? NAN : 0.0; // P == 0 -> L = 0 (rho = 0 -> NaN, as in the reference); inf/NaN -> NaN
        }
    }

    // ---- Solver.cpp:72-77 / 531-536
    double p40, p15;
    G::pow_pair(L / mu, p40, p15);
    double rho = sqrt(mu * L) * p40;
    double tau_inc = p15, tau_dec = tau_inc;
    double inv_rho = fast_rcp(rho);
    bool bad = !(rho > 0.0) || !(rho < 1.79e308);
    // Every M[e] receives the same sequence of additions and rounding is monotone, so the smallest M[e] is always
    // the one of the smallest p[e]: positivity of the shifted diagonal after a rho update is ONE comparison.
    double Mmin = p[0];
#pragma unroll
    for (int e = 1; e < E; ++e) Mmin = fmin(Mmin, p[e]);
    Mmin = Mmin + (rho + mu);
#pragma unroll
    for (int e = 0; e < E; ++e) {
        M[e] = p[e] + (rho + mu);
        bad = bad || !(M[e] > 0.0);   // (also catches a NaN entry, which fmin above would drop)
        qp[e] = q[e];
        l2[e] = 0.0;
        u[e] = 0.0;
    }
    rcp_all<E>(M, Minv);

    // A per-lane loop: a lane leaves it when its problem stops (all lanes of a problem decide alike: the group
    // reductions are symmetric) and keeps its state in place under the execution mask -- as a wave-uniform loop
    // around a `done` flag the same code carried a dozen register copies per iteration.
    int rho_up = 0, cpt = 0, iters = 0;
    if (valid) {
        for (int it = 0; it < max_iter; ++it) {
            double rd = 0.0, rp = 0.0;
            double w[E], z[E], lv[(KIND == 1) ? E : 1];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const double l = Minv[e] * (rho * l2[e] - u[e] - qp[e]);      // :80 / :539
                qp[e] = q[e] - mu * l;                                        // :81 / :540
                w[e] = kAlpha * l + (1 - kAlpha) * l2[e];                     // alpha*l + (1-alpha)*l2_pred
                z[e] = w[e] + u[e] * inv_rho;                                 // :82 / :541
                if (KIND == 1) lv[e] = l;
            }
            if (KIND == 0) {
#pragma unroll
                for (int e = 0; e < E; ++e) z[e] = fmax(z[e], 0.0);           // cwiseMax(0), :82
            } else if (KIND == 2 || KIND == 3) {
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    double t = z[e];
                    t = t < lo[e] ? lo[e] : t;                                // cwiseMax(l_min), :219 / :396
                    t = hi[e] < t ? hi[e] : t;                                // cwiseMin(l_max), :220 / :397
                    if (KIND == 3) {                                          // v o min(v o l_2, 0), :398
                        double m = sg[e] * t;
                        m = 0 < m ? 0 : m;
                        t = sg[e] * m;
                    }
                    z[e] = t;
                }
            } else {
#pragma unroll
                for (int c = 0; c < E / 2; ++c) {                             // prox_circle, :505-519
                    const double a = z[2 * c], b = z[2 * c + 1];
                    const double n2 = a * a + b * b;
                    const double rn = fast_rsqrt(n2);      // n2 == 0: inf -> nrm NaN -> no scaling
                    const double nrm = n2 * rn;
                    const double sc = (nrm > rad[c]) ? rad[c] * rn : 1.0; // one select; * 1.0 is exact
                    z[2 * c] = a * sc;
                    z[2 * c + 1] = b * sc;
                }
            }
            double dw[E], dz[E];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                dw[e] = w[e] - z[e];
                dz[e] = z[e] - l2[e];
                u[e] += rho * dw[e];                                          // :83 / :543
                l2[e] = z[e];                                                 // :87 / :547
            }
            rd = max_abs2(dz[0], dz[1]);                                      // :84-85 / :544-545
            rp = max_abs2(dw[0], dw[1]);                                      // :86 / :546
#pragma unroll
            for (int e = 2; e < E; ++e) {
                rd = max_abs1(rd, dz[e]);
                rp = max_abs1(rp, dw[e]);
            }
            rd = G::max(rd);
            rp = G::max(rp);
            const double res_dual = rho * rd;   // max|rho*d| == rho*max|d| for rho > 0
            const double res_prim = rp;
            iters = it + 1;
            bool stop = res_dual < eps;                                       // :88
            if (KIND == 1) {
                if (stop) {
                    // res_prim < eps + 1e-4 |l|_2 (:548), only evaluated once the dual test passes, and without
                    // the square root: t = res_prim - eps < 0, or t^2 < 1e-8 |l|^2
                    double nl = 0.0;
#pragma unroll
                    for (int e = 0; e < E; ++e) nl += lv[e] * lv[e];
                    nl = G::sum(nl);
                    const double t = res_prim - eps;
                    stop = t < 0.0 || t * t < (kEpsRel * kEpsRel) * nl;
                }
            }
            if (stop) break;
            if (adaptive) {
                // rho adaptation, Solver.cpp:90-120 / 550-580: increase when the primal residual dominates,
                // decrease when the dual one does, at most once every 5 imbalanced iterations.  Same state
                // machine as common.h RhoSchedule (used by the general kernels), kept inline here: through the
                // struct this kernel measured 0.7 us (2 %) slower at the bench shape.
                const bool inc = res_prim > kMuThresh * res_dual;             // :92 / :552
                const bool dec = !inc && (res_dual > kMuThresh * res_prim);   // :106 / :566
                const bool imb = inc || dec;
                const bool fire = imb && (cpt == 0);                          // cpt % 5 == 0
                cpt = imb ? (cpt == 4 ? 0 : cpt + 1) : cpt;                   // cpt++ (kept mod 5)
                if (fire) {
                    if (rho_up == (inc ? -1 : 1)) {                           // direction flipped: damp tau
                        const double ti = 1 + .8 * (tau_inc - 1), td = 1 + .8 * (tau_dec - 1);
                        if (QP_LIKE) { tau_inc = ti; tau_dec = td; }          // :94-97, :108-111 (QP damps both)
                        else if (inc) tau_inc = ti;                           // :554-556
                        else tau_dec = td;                                    // :568-570
                    }
                    // one reciprocal per update, of the factor in use: rho and 1/rho move by reciprocal factors
                    const double tau = inc ? tau_inc : tau_dec, inv_tau = fast_rcp(tau);
                    const double f = inc ? tau : inv_tau;                     // rho *= tau_inc | rho /= tau_dec
                    const double delta = rho * (f - 1);                       // :98 / :112
                    rho = rho * f;                                            // :99 / :113
                    rho_up = inc ? 1 : -1;
                    inv_rho = inv_rho * (inc ? inv_tau : tau);
                    // llt() + solveInPlace(Identity) of the shifted matrix, diagonal case (:100-101)
                    Mmin += delta;
                    bad = bad || !(Mmin > 0.0);
#pragma unroll
                    for (int e = 0; e < E; ++e) M[e] += delta;
                    rcp_all<E>(M, Minv);
                }
            }
        }
    }
    bad = G::max(bad ? 1.0 : 0.0) > 0.0;
#pragma unroll
    for (int e = 0; e < E; ++e) x[e] = bad ? NAN : l2[e];
    return iters;
}

} // namespace dqq
