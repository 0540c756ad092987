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
// Reciprocals of E positive numbers.  E > 8: from ONE reciprocal (of their product) and 3(E-1) multiplications; each
// result then carries ~3 roundings instead of 1, and a product that leaves the double range (entries beyond ~1e75) or
// a non-positive factor falls back to one reciprocal per entry.  E <= 8 (every instantiation of the fast path): one
// reciprocal per entry -- with the range
// test and its branch the shared reciprocal saved nothing there (11 against 10 VALU instructions per pair), and the
// result must not depend on how a problem's coordinates are grouped into lanes (admm_fwd_diag_respread moves problems
// from E = 4 to E = 2 in mid-solve; round 4: the forward picks among one, two and four lanes per problem at N = 8 by a hint).
template <int E>
DQQ_HD void rcp_all(const double (&m)[E], double (&inv)[E])
{
    if constexpr (E <= 8) {
#pragma unroll
        for (int e = 0; e < E; ++e) inv[e] = fast_rcp(m[e]);
        return;
    }
    // (E > 8 from here on)
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
    {   // (the lean body: see admm_diag_resume)
        const bool adaptive_on = adaptive != 0;
        int badi = bad ? 1 : 0;
        if (valid) {
            for (int it = 0; it < max_iter; ++it) {
#define DQQ_ADMM_LEAN 1
#define DQQ_ADMM_ON_STOP break
#include "admm_diag_body.inc"
#undef DQQ_ADMM_ON_STOP
#undef DQQ_ADMM_LEAN
            }
        }
        bad = badi != 0;
    }
    bad = G::max(bad ? 1.0 : 0.0) > 0.0;
#pragma unroll
    for (int e = 0; e < E; ++e) x[e] = bad ? NAN : l2[e];
    return iters;
}


#if defined(__HIPCC__) // (the solver above also compiles for the host: tests/hostcore)
// The loop of admm_fwd_diag from iteration it0 on, over state that already exists (admm_fwd_diag_respread's later
// phases).  Returns through the state; `iters` is the number of iterations executed in total.  lanes_at > 0: the loop is
// also left -- with more = true on the lanes whose problems still run, and the next iteration in it_next -- once at
// most that many lanes of the wave are still in it and at least exit_from iterations have been executed.
template <int KIND, int E, class G>
DQQ_D void admm_diag_resume(double (&M)[E], double (&Minv)[E], const double (&q)[E], double (&qp)[E], double (&l2)[E],
                            double (&u)[E], const double* rad, double& rho, double& inv_rho, double& tau_inc,
                            double& tau_dec, double& Mmin, int& rho_up, int& cpt, bool& bad, int& iters, int it0,
                            int max_iter, double eps, double mu, int adaptive, bool valid, int lanes_at, bool& more,
                            int& it_next, int exit_from = 0)
{
    constexpr bool QP_LIKE = (KIND != 1);
    static_assert(KIND < 2, "QP / QCQP");
    const double *lo = nullptr, *hi = nullptr, *sg = nullptr;
    (void)lo; (void)hi; (void)sg;
    more = false;
    it_next = it0;
    double itau_inc = fast_rcp(tau_inc), itau_dec = fast_rcp(tau_dec);   // (state of the body; a function of the taus)
    // per-lane loop (a lane leaves it when its problem stops), with the lean body: `bad` as an int in a VGPR (a bool live
    // across the divergent exits costs three scalar instructions at every join), `adaptive` folded into `fire`, the QCQP's
    // second stopping test behind a wave-uniform branch
    if constexpr (E == 1) {
        // The last survivors of a tile, one coordinate per lane: by now most waves of the launch have retired and this one
        // is (nearly) alone on its SIMD -- what an iteration costs it is the length of its dependent chain, not its
        // instruction count, and the plain body schedules better there: the scalar mask bookkeeping of its exits sits in the
        // hazard slots of the residual reductions, and the reference's figure workload (one problem at 25 000 iterations)
        // reads 4.2-4.6 ms with it against 5.2-5.7 with the lean body (A/B on one box, three alternations, with the interleaved
        // residual reductions in both: profiles/r07_ab_tail_body.txt).
        // (the last stage: nothing leaves this loop early -- lanes_at is 0 here)
        if (valid) {
            for (int it = it0; it < max_iter; ++it) {
#define DQQ_ADMM_ON_STOP break
#include "admm_diag_body.inc"
#undef DQQ_ADMM_ON_STOP
            }
        }
    } else {
        const bool adaptive_on = adaptive != 0;
        int badi = bad ? 1 : 0;
        if (valid) {
            for (int it = it0; it < max_iter; ++it) {
#define DQQ_ADMM_LEAN 1
#define DQQ_ADMM_ON_STOP break
#include "admm_diag_body.inc"
#undef DQQ_ADMM_ON_STOP
#undef DQQ_ADMM_LEAN
                if (lanes_at > 0 && it + 1 >= exit_from && __popcll(__ballot(true)) <= lanes_at) {
                    more = it + 1 < max_iter;
                    it_next = it + 1;
                    break;
                }
            }
        }
        bad = badi != 0;
    }
}

// admm_fwd_diag for N = 8 on TWO lanes per problem (E = 4) that re-spreads the tail of its tile: once at most
// `respread_at` (<= 16) of the wave's 32 problems are still iterating, their state moves -- through `lds`, 256
// wave-private doubles -- onto FOUR lanes per problem (E = 2, problem r of the survivors on lanes 4r..4r+3) and
// the loop goes on with half the per-coordinate arithmetic per iteration: a launch lasts as long as its slowest
// problems (8..38 iterations at the bench shape, median 17) and those spent the second half of their iterations
// in a half-empty wave.
// Results are bit-identical to the un-respread solve: the per-coordinate arithmetic is the same code, the group
// maxima are exact, the reciprocals are taken entry by entry (rcp_all) and |l|^2 is summed contact by contact
// in the order of the lane tree -- so a problem's result does not depend on what else is in its tile.
// A problem that moved (moved = true on its two phase-1 lanes) has its x and iteration count stored from here
// (xout / itout point at the tile's first problem); the others return theirs as admm_fwd_diag does.
template <int KIND>
DQQ_D int admm_fwd_diag_respread(const double (&p)[4], const double (&q)[4], const double* rad, double eps, double mu,
                                 int max_iter, int adaptive, bool valid, double (&x)[4], int respread_at, int respread2_at,
                                 double* lds, double* __restrict__ xout, int* __restrict__ itout, bool& moved)
{
    constexpr int E = 4;
    using G = LaneGroup<2>;
    constexpr bool QP_LIKE = (KIND != 1);
    static_assert(KIND < 2, "QP / QCQP");
    const double *lo = nullptr, *hi = nullptr, *sg = nullptr;
    (void)lo; (void)hi; (void)sg;
#include "admm_diag_prologue.inc"

    int rho_up = 0, cpt = 0, iters = 0, it_next = 0;
    bool more = false;
    const int lanes_at = 2 * respread_at;
    {   // (per-lane loop with the lean body: see admm_diag_resume)
        const bool adaptive_on = adaptive != 0;
        int badi = bad ? 1 : 0;
        if (valid) {
            for (int it = 0; it < max_iter; ++it) {
#define DQQ_ADMM_LEAN 1
#define DQQ_ADMM_ON_STOP break
#include "admm_diag_body.inc"
#undef DQQ_ADMM_ON_STOP
#undef DQQ_ADMM_LEAN
                // lanes still in this loop = the execution mask
                if (__popcll(__ballot(true)) <= lanes_at) {
                    more = it + 1 < max_iter;
                    it_next = it + 1;
                    break;
                }
            }
        }
        bad = badi != 0;
    }
    moved = more;
    const unsigned long long mm = __ballot(more);
    if (mm != 0) { // wave-uniform
        constexpr int E2 = 2;
        using G4 = LaneGroup<4>;
        const int lane = threadIdx.x & 63, h = lane & 1;
        const unsigned long long firsts = mm & 0x5555555555555555ull;       // one bit per moving problem
        const int r = __popcll(firsts & ((1ull << (lane & ~1)) - 1));       // its rank among them
        const int nmv = __popcll(firsts);
        const bool valid2 = lane < 4 * nmv;
        // four coordinates of lane (r, h) -> slots 8r+4h .. +3; lane 4r+s of the new layout owns slots 8r+2s, +1 = 2*lane, +1
        double* mine = lds + 8 * r + 4 * h;
        auto move = [&](const double (&a)[4], const double (&b)[4], double (&a2)[2], double (&b2)[2]) {
            wave_lds_fence();
            if (more) {
                *reinterpret_cast<double2*>(mine) = make_double2(a[0], a[1]);
                *reinterpret_cast<double2*>(mine + 2) = make_double2(a[2], a[3]);
                *reinterpret_cast<double2*>(mine + 128) = make_double2(b[0], b[1]);
                *reinterpret_cast<double2*>(mine + 130) = make_double2(b[2], b[3]);
            }
            wave_lds_fence();
            const double2 ta = *reinterpret_cast<const double2*>(lds + 2 * lane);
            const double2 tb = *reinterpret_cast<const double2*>(lds + 128 + 2 * lane);
            a2[0] = ta.x; a2[1] = ta.y;
            b2[0] = tb.x; b2[1] = tb.y;
        };
        double M2[E2], Minv2[E2], q2[E2], qp2[E2], l22[E2], u2[E2], rad2[1];
        move(M, Minv, M2, Minv2);
        move(q, qp, q2, qp2);
        move(l2, u, l22, u2);
        // per-problem scalars (identical on both lanes of a problem: lane h = 0 hands them over), the radii and
        // the problem's position in the tile
        wave_lds_fence();
        int* ldsi = reinterpret_cast<int*>(lds + 128);
        if (more) {
            if (h == 0) {
                lds[r] = rho;
                lds[16 + r] = inv_rho;
                lds[32 + r] = tau_inc;
                lds[48 + r] = tau_dec;
                ldsi[r] = rho_up;
                ldsi[16 + r] = cpt;
                ldsi[32 + r] = lane >> 1;
                ldsi[48 + r] = bad ? 1 : 0;
                ldsi[64] = it_next;  // the same on every lane that moves
            }
            lds[64 + 4 * r + 2 * h] = rad[0];
            lds[64 + 4 * r + 2 * h + 1] = rad[1];
        }
        wave_lds_fence();
        const int g = lane >> 2;
        double rho2 = lds[g], inv_rho2 = lds[16 + g], tau_inc2 = lds[32 + g], tau_dec2 = lds[48 + g];
        int rho_up2 = ldsi[g], cpt2 = ldsi[16 + g];
        const int pl2 = ldsi[32 + g];
        bool bad2 = ldsi[48 + g] != 0;
        const int it0 = ldsi[64];
        rad2[0] = lds[64 + lane];
        wave_lds_fence();
        if (!valid2) { // idle lanes: harmless numbers
            rho2 = inv_rho2 = tau_inc2 = tau_dec2 = 1.0;
            rad2[0] = 1.0;
            bad2 = false;
#pragma unroll
            for (int e = 0; e < E2; ++e) { M2[e] = Minv2[e] = 1.0; q2[e] = qp2[e] = l22[e] = u2[e] = 0.0; }
        }
        double Mmin2 = fmin(M2[0], M2[1]);
        int iters2 = it0, it_next2 = it0;
        bool more2 = false;
        admm_diag_resume<KIND, E2, G4>(M2, Minv2, q2, qp2, l22, u2, rad2, rho2, inv_rho2, tau_inc2, tau_dec2, Mmin2,
                                       rho_up2, cpt2, bad2, iters2, it0, max_iter, eps, mu, adaptive, valid2,
                                       4 * (respread2_at & 0xff), more2, it_next2, respread2_at >> 8);
        const unsigned long long mm2 = __ballot(more2);
        if (!more2) {
            bad2 = G4::max(bad2 ? 1.0 : 0.0) > 0.0;
            if (valid2) {
                *reinterpret_cast<double2*>(xout + pl2 * 8 + 2 * (lane & 3)) =
                    bad2 ? make_double2(NAN, NAN) : make_double2(l22[0], l22[1]);
                if (itout != nullptr && (lane & 3) == 0) itout[pl2] = iters2;
            }
        }
        if (mm2 != 0) { // wave-uniform
            // ---- the last survivors (at most respread2_at <= 8 problems) move onto EIGHT lanes per problem, one coordinate
            // per lane: what such a tail costs is its instruction count per iteration (a lone problem's wave issues one
            // instruction every few cycles whatever its width), and the E = 1 body is about 60 % of the E = 2 body.
            constexpr int E3 = 1;
            using G8 = LaneGroup<8>;
            const int s4 = lane & 3;
            const unsigned long long firsts3 = mm2 & 0x1111111111111111ull;       // one bit per moving problem
            const int r3 = __popcll(firsts3 & ((1ull << (lane & ~3)) - 1));
            const int nmv3 = __popcll(firsts3);
            const bool valid3 = lane < 8 * nmv3;
            double* mine3 = lds + 8 * r3 + 2 * s4;     // coordinates 2 s4, 2 s4 + 1 of problem r3 -> slots 8 r3 + 2 s4, + 1
            auto move3 = [&](const double (&a)[2], const double (&b)[2], double (&a3)[1], double (&b3)[1]) {
                wave_lds_fence();
                if (more2) {
                    *reinterpret_cast<double2*>(mine3) = make_double2(a[0], a[1]);
                    *reinterpret_cast<double2*>(mine3 + 128) = make_double2(b[0], b[1]);
                }
                wave_lds_fence();
                a3[0] = lds[lane];
                b3[0] = lds[128 + lane];
            };
            double M3[E3], Minv3[E3], q3[E3], qp3[E3], l23[E3], u3[E3], rad3[1];
            move3(M2, Minv2, M3, Minv3);
            move3(q2, qp2, q3, qp3);
            move3(l22, u2, l23, u3);
            wave_lds_fence();
            if (more2) {
                if (s4 == 0) {
                    lds[r3] = rho2;
                    lds[16 + r3] = inv_rho2;
                    lds[32 + r3] = tau_inc2;
                    lds[48 + r3] = tau_dec2;
                    ldsi[r3] = rho_up2;
                    ldsi[16 + r3] = cpt2;
                    ldsi[32 + r3] = pl2;
                    ldsi[48 + r3] = bad2 ? 1 : 0;
                    ldsi[64] = it_next2;
                }
                lds[64 + 4 * r3 + s4] = rad2[0];      // lane s4 of the four holds contact s4
            }
            wave_lds_fence();
            const int g3 = lane >> 3;
            double rho3 = lds[g3], inv_rho3 = lds[16 + g3], tau_inc3 = lds[32 + g3], tau_dec3 = lds[48 + g3];
            int rho_up3 = ldsi[g3], cpt3 = ldsi[16 + g3];
            const int pl3 = ldsi[32 + g3];
            bool bad3 = ldsi[48 + g3] != 0;
            const int it03 = ldsi[64];
            rad3[0] = lds[64 + 4 * g3 + ((lane & 7) >> 1)];
            wave_lds_fence();
            if (!valid3) {
                rho3 = inv_rho3 = tau_inc3 = tau_dec3 = 1.0;
                rad3[0] = 1.0;
                bad3 = false;
                M3[0] = Minv3[0] = 1.0;
                q3[0] = qp3[0] = l23[0] = u3[0] = 0.0;
            }
            double Mmin3 = M3[0];
            int iters3 = it03, it_next3 = it03;
            bool more3 = false;
            admm_diag_resume<KIND, E3, G8>(M3, Minv3, q3, qp3, l23, u3, rad3, rho3, inv_rho3, tau_inc3, tau_dec3, Mmin3,
                                           rho_up3, cpt3, bad3, iters3, it03, max_iter, eps, mu, adaptive, valid3, 0, more3,
                                           it_next3);
            bad3 = G8::max(bad3 ? 1.0 : 0.0) > 0.0;
            if (valid3) {
                xout[pl3 * 8 + (lane & 7)] = bad3 ? NAN : l23[0];
                if (itout != nullptr && (lane & 7) == 0) itout[pl3] = iters3;
            }
        }
    }
    bad = G::max(bad ? 1.0 : 0.0) > 0.0;
#pragma unroll
    for (int e = 0; e < E; ++e) x[e] = bad ? NAN : l2[e];
    return iters;
}
#endif // __HIPCC__

} // namespace dqq
