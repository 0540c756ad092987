// group_dense.h -- general (dense P) ADMM forward solve on the lane mapping of the diagonal fast path: the LPP
// lanes that share a problem in fwd_diag_kernel also solve it when its P is NOT diagonal.
//
// Lane s of a group owns rows s*E .. s*E+E-1 (E = N / LPP) of P, of its powers (power iteration) and of
// M^-1 = (P + (rho+mu) I)^-1, all in registers; a mat-vec is one all-gather of the vector over the group (quad_perm
// DPP broadcasts, LPP <= 4) and E*N multiply-adds per lane; the explicit inverse at a rho update is an in-place
// Gauss-Jordan sweep whose pivot row is broadcast from its owner.  A wave therefore advances ALL problems of its
// tile at once (32 at N = 8, LPP = 2) where the per-problem fallback (dense_core.h dense_fwd_problem) took them
// one by one -- 25x on a dense batch through DQQ_P_AUTO.
//
// Algorithm, update order, constants and stopping tests are the reference's (Solver::solveQP / solveQCQP /
// solveBoxQP / solveSignedBoxQP, Solver.cpp:61-123, 521-582, 198-261, 374-439; power_iteration :46-59): the ADMM
// iteration is the very text of the diagonal path (admm_diag_body.inc) with the diagonal scale replaced by the
// mat-vec with M^-1 and the E reciprocals by the inverse.  As in fwd_lane_dense.hip: the factorisation reads the
// LOWER triangle of P only (Eigen's llt(), Solver.cpp:76), the power iteration and nothing else the full matrix;
// the 100 power steps of the QCQP are six matrix squarings and three mat-vecs with exact power-of-two rescaling.
// Ulp-level departures: Gauss-Jordan instead of LLT + substitutions (same pivots d_k > 0 <=> LLT succeeds), group
// (tree) sums for the 2-norms, FMA contraction, 1-ulp reciprocal / reciprocal square root.
#pragma once

#include "admm_core.h"

namespace dqq {

#if defined(__HIPCC__)

// (rows per lane) x N matrix entries per lane: 32 keeps the QP / QCQP kernels under 256 VGPRs (two waves per SIMD,
// as without it); the box kinds carry three more vectors and get half of that
constexpr bool group_dense_supported(int kind, int n, int lpp)
{
    if (kind < 2 && n == 8 && lpp == 1) return true;   // a lane per problem, the whole matrix in its registers: one wave per SIMD (fwd_diag.hip)
    return lpp <= 4 && n % lpp == 0 && (n / lpp) * n <= (kind < 2 ? 32 : 16);
}

template <int N, int LPP>
struct GroupRows {
    static constexpr int E = N / LPP;
    static_assert(LPP == 1 || LPP == 2 || LPP == 4, "quad_perm broadcasts");

    // the value lane J of the group holds
    template <int J>
    static DQQ_D double bcast(double v)
    {
        if constexpr (LPP == 1) return v;
        else if constexpr (LPP == 2) return dpp_f64<J | (J << 2) | ((2 + J) << 4) | ((2 + J) << 6)>(v);
        else return dpp_f64<J | (J << 2) | (J << 4) | (J << 6)>(v);
    }
    static DQQ_D double bcast_from(double v, int j) // j: a constant after unrolling
    {
        if constexpr (LPP == 1) return v;
        else if constexpr (LPP == 2) return j == 0 ? bcast<0>(v) : bcast<1>(v);
        else return j == 0 ? bcast<0>(v) : j == 1 ? bcast<1>(v) : j == 2 ? bcast<2>(v) : bcast<3>(v);
    }
    // every lane's E entries -> the whole vector, in coordinate order
    static DQQ_D void gather(const double (&own)[E], double (&full)[N])
    {
#pragma unroll
        for (int j = 0; j < LPP; ++j)
#pragma unroll
            for (int e = 0; e < E; ++e) full[j * E + e] = bcast_from(own[e], j);
    }
    // y = A x for the lane's rows of A
    static DQQ_D void matvec(const double (&A)[E][N], const double (&own)[E], double (&y)[E])
    {
        double full[N];
        gather(own, full);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            double t = 0.0;
#pragma unroll
            for (int c = 0; c < N; ++c) t = __builtin_fma(A[e][c], full[c], t);   // (written out: see dot() below)
            y[e] = t;
        }
    }
    // sum_i a_i b_i over the problem's N coordinates: every product rounded, then ONE binary tree in coordinate order --
    // within the lane here, across the lanes in G::sum.  Round 4: the general solve runs on four lanes per problem or on
    // one (fwd_diag.hip picks by a hint), and a result must not depend on which; for the same reason every fused
    // multiply-add of this file is written out instead of being left to the compiler's contraction.
    static DQQ_D double dot(const double (&a)[E], const double (&b)[E])
    {
#pragma clang fp contract(off)
        static_assert((E & (E - 1)) == 0, "rows per lane: a power of two");
        double pr[E];
#pragma unroll
        for (int e = 0; e < E; ++e) pr[e] = a[e] * b[e];
#pragma unroll
        for (int w = 1; w < E; w *= 2)
#pragma unroll
            for (int e = 0; e + w < E; e += 2 * w) pr[e] = pr[e] + pr[e + w];
        return LaneGroup<LPP>::sum(pr[0]);
    }
    // rows s*E.. of the matrix as stored.  SC: 8-byte loads (the matrix sits in the LDS tile of stage_tile_lane8, whose
    // row stride of N*N + 1 doubles leaves it 8-byte aligned only)
    template <bool SC = false>
    static DQQ_D void load_rows(const double* __restrict__ Pg, int s, double (&A)[E][N])
    {
#pragma unroll
        for (int e = 0; e < E; ++e)
#pragma unroll
            for (int c = 0; c < N; c += 2) {
                if constexpr (SC) {
                    A[e][c] = Pg[(s * E + e) * N + c];
                    A[e][c + 1] = Pg[(s * E + e) * N + c + 1];
                } else {
                    const double2 t = *reinterpret_cast<const double2*>(Pg + (s * E + e) * N + c);
                    A[e][c] = t.x;
                    A[e][c + 1] = t.y;
                }
            }
    }
    // rows s*E.. of the symmetric matrix the LOWER triangle of P defines (what llt() factorises), diagonal `md`.
    // Entry (g, c) is P[g][c] for c <= g and P[c][g] beyond: both the row g and the column g are loaded, from ONE
    // per-lane pointer each with compile-time offsets, and the side is chosen per entry.  (Indexed as
    // Pg[max(g,c) * N + min(g,c)] every entry had its own per-lane 64-bit address, which the compiler hoisted out of
    // the ADMM loop -- the refactorisation sits inside it -- and kept in ~30 VGPRs for the whole solve.)
    template <bool SC = false>
    static DQQ_D void load_lower_symmetric(const double* __restrict__ Pg, int s, const double (&md)[E],
                                           double (&A)[E][N])
    {
        const double* rowp = Pg + (s * E) * N; // rows s*E .. s*E+E-1, contiguous
        const double* colp = Pg + s * E;       // columns s*E .. : entry [c][g] at colp[c * N + e]
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int g = s * E + e;
#pragma unroll
            for (int c = 0; c < N; c += 2) {
                double2 r;
                if constexpr (SC) r = make_double2(rowp[e * N + c], rowp[e * N + c + 1]);
                else r = *reinterpret_cast<const double2*>(rowp + e * N + c);
                const double c0 = colp[c * N + e], c1 = colp[(c + 1) * N + e];
                A[e][c] = (c <= g) ? r.x : c0;
                A[e][c + 1] = (c + 1 <= g) ? r.y : c1;
            }
#pragma unroll
            for (int j = 0; j < LPP; ++j) A[e][j * E + e] = (s == j) ? md[e] : A[e][j * E + e];
        }
    }
    // A <- A^-1 in place (Gauss-Jordan without pivoting: A symmetric positive definite, else `bad`)
    static DQQ_D void invert(double (&A)[E][N], int s, bool& bad)
    {
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const int j = k / E, ek = k % E;
            double rk[N];
#pragma unroll
            for (int c = 0; c < N; ++c) rk[c] = bcast_from(A[ek][c], j);
            bad = bad || !(rk[k] > 0.0);
            const double pinv = fast_rcp(rk[k]);
#pragma unroll
            for (int c = 0; c < N; ++c) rk[c] = (c == k) ? pinv : rk[c] * pinv;
            const bool owner = (s == j);
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const double f = A[e][k];
#pragma unroll
                for (int c = 0; c < N; ++c) {
                    const double upd = (c == k) ? -(f * pinv) : __builtin_fma(-f, rk[c], A[e][c]);
                    A[e][c] = (e == ek && owner) ? rk[c] : upd;
                }
            }
        }
    }
};

// Pg: this problem's P (N x N, row-major).  q, rad, lo, hi, sg, x, valid, return value: as admm_fwd_diag.
// SC: Pg points into LDS (8-byte aligned): scalar loads.
template <int KIND, int N, int LPP, bool SC = false>
DQQ_D int group_dense_fwd(const double* __restrict__ Pg, const double (&q)[N / LPP], const double* rad, double eps,
                          double mu, int max_iter, int adaptive, bool valid, double (&x)[N / LPP],
                          const double* lo = nullptr, const double* hi = nullptr, const double* sg = nullptr,
                          int defer = 4)
{
    constexpr int E = N / LPP;
    constexpr bool QP_LIKE = (KIND != 1);
    using G = LaneGroup<LPP>;
    using R = GroupRows<N, LPP>;
    const int s = (threadIdx.x & 63) % LPP;

    double A[E][N]; // rows of P, of its powers, then of M^-1
    if (valid) {
        R::template load_rows<SC>(Pg, s, A);
    } else {
#pragma unroll
        for (int e = 0; e < E; ++e)
#pragma unroll
            for (int c = 0; c < N; ++c) A[e][c] = (c % E == e) ? 1.0 : 0.0; // keeps the arithmetic finite
    }
    double md[E]; // the diagonal of P + (rho+mu) I, accumulated as the reference does (:98-100)
#pragma unroll
    for (int e = 0; e < E; ++e) {
        md[e] = A[e][e];
#pragma unroll
        for (int j = 1; j < LPP; ++j) md[e] = (s == j) ? A[e][j * E + e] : md[e];
    }

    // ---- power_iteration, Solver.cpp:46-59 (10 steps for the QP-like solvers :71, 100 for the QCQP :530)
    double L;
    {
        double v[E];
        const double c0 = 1.0 / sqrt((double)N);
        double ss = 0.0;
#pragma unroll
        for (int i = 0; i < N; ++i) ss += c0 * c0;
        const double nn = sqrt(ss);
#pragma unroll
        for (int e = 0; e < E; ++e) v[e] = c0 / nn;
        if constexpr (QP_LIKE) {
#pragma unroll 1
            for (int k = 0; k < 10; ++k) {
                double Av[E];
                R::matvec(A, v, Av);
                const double t = R::dot(Av, Av);
                const double inv = t > 0 ? fast_rsqrt(t) : 1.0; // normalised every step (Solver.cpp:53)
#pragma unroll
                for (int e = 0; e < E; ++e) v[e] = Av[e] * inv;
            }
        } else {
            // P^100 v0 = P^64 (P^32 (P^4 v0)); every squared matrix and vector rescaled by an exact power of two
#pragma unroll 1
            for (int sq = 1; sq <= 6; ++sq) {
                double T2[E][N];
#pragma unroll
                for (int k = 0; k < N; ++k) {
                    double rowk[N];
#pragma unroll
                    for (int c = 0; c < N; ++c) rowk[c] = R::bcast_from(A[k % E][c], k / E);
#pragma unroll
                    for (int e = 0; e < E; ++e)
#pragma unroll
                        for (int c = 0; c < N; ++c) T2[e][c] = (k == 0) ? A[e][0] * rowk[c] : __builtin_fma(A[e][k], rowk[c], T2[e][c]);
                }
                double dmax = 0.0;
#pragma unroll
                for (int e = 0; e < E; ++e)
#pragma unroll
                    for (int c = 0; c < N; ++c) dmax = fmax(dmax, fabs(T2[e][c]));
                dmax = G::max(dmax);
                int ex = 0;
                if (dmax > 0.0 && dmax < 1.79e308) (void)frexp(dmax, &ex);
#pragma unroll
                for (int e = 0; e < E; ++e)
#pragma unroll
                    for (int c = 0; c < N; ++c) A[e][c] = ldexp(T2[e][c], -ex);
                if (sq == 2 || sq == 5 || sq == 6) { // A = P^4, P^32, P^64 up to a power of two
                    double Av[E];
                    R::matvec(A, v, Av);
                    double vm = 0.0;
#pragma unroll
                    for (int e = 0; e < E; ++e) vm = fmax(vm, fabs(Av[e]));
                    vm = G::max(vm);
                    int ev = 0;
                    if (vm > 0.0 && vm < 1.79e308) (void)frexp(vm, &ev);
#pragma unroll
                    for (int e = 0; e < E; ++e) v[e] = ldexp(Av[e], -ev);
                }
            }
            const double t = R::dot(v, v);
            const double inv = t > 0 ? fast_rsqrt(t) : 1.0;
#pragma unroll
            for (int e = 0; e < E; ++e) v[e] = v[e] * inv;
            if (valid) R::template load_rows<SC>(Pg, s, A); // the Rayleigh quotient is taken with P itself
        }
        double Pv[E];
        R::matvec(A, v, Pv);
        L = R::dot(v, Pv);
    }

    // ---- Solver.cpp:72-77 / 531-536
    double p40, p15;
    G::pow_pair(L / mu, p40, p15);
    double rho = sqrt(mu * L) * p40;
    double tau_inc = p15, tau_dec = tau_inc;
    double itau_inc = fast_rcp(tau_inc), itau_dec = itau_inc;   // (admm_diag_body.inc: the reciprocals are state)
    double inv_rho = fast_rcp(rho);
    bool bad = !(rho > 0.0) || !(rho < 1.79e308);
    double qp[E], l2[E], u[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        md[e] += rho + mu;
        qp[e] = q[e];
        l2[e] = 0.0;
        u[e] = 0.0;
    }
    if (valid) R::template load_lower_symmetric<SC>(Pg, s, md, A);
    R::invert(A, s, bad);

#define DQQ_ADMM_GENERAL_P 1
#define DQQ_ADMM_SOLVE(l)                                                                                             \
    do {                                                                                                              \
        double rhs_[E];                                                                                               \
        _Pragma("unroll") for (int e_ = 0; e_ < E; ++e_) rhs_[e_] = __builtin_fma(rho, l2[e_], -u[e_]) - qp[e_];                  \
        R::matvec(A, rhs_, l);                                                                                        \
    } while (0)
    // The refactorisation after a rho update is DEFERRED (as in fwd_lane_dense.hip; knob lane_defer, tuning.h): the group
    // updates rho, 1/rho and the shifted diagonal on the spot and sits out until the wave next runs the sweep -- every
    // `defer`-th trip of the loop, or as soon as no group has anything else to do.  One sweep (~590 instructions, a
    // trip of the iteration is ~150) then serves every group that changed rho since the last one; with 16 problems
    // per wave some group fires on most trips otherwise.  A problem's own arithmetic does not depend on `defer`.
#define DQQ_ADMM_REFACTOR(delta)                                                                                      \
    do {                                                                                                              \
        _Pragma("unroll") for (int e_ = 0; e_ < E; ++e_) md[e_] += (delta);                                           \
        pend = true;                                                                                                  \
    } while (0)
    int rho_up = 0, cpt = 0, iters = 0;
    bool run = valid && max_iter > 0, pend = false;
    int it = 0;
    for (int trip = 0;; ++trip) {
        if (run && !pend) {
            do {
#define DQQ_ADMM_ON_STOP { run = false; break; }
#include "admm_diag_body.inc"
#undef DQQ_ADMM_ON_STOP
            } while (0);
            ++it;
            if (it >= max_iter) run = false;
        }
        if (!__any(run)) break;
        if (__any(pend) && ((trip + 1) % defer == 0 || !__any(run && !pend))) {
            if (pend) {
                R::template load_lower_symmetric<SC>(Pg, s, md, A);
                R::invert(A, s, bad);                                   // llt() + solveInPlace(Identity), :100-101
            }
            pend = false;
        }
    }
#undef DQQ_ADMM_REFACTOR
#undef DQQ_ADMM_SOLVE
#undef DQQ_ADMM_GENERAL_P
    bad = G::max(bad ? 1.0 : 0.0) > 0.0;
#pragma unroll
    for (int e = 0; e < E; ++e) x[e] = bad ? NAN : l2[e];
    return iters;
}

// A non-diagonal tile of the fast path (TILE consecutive problems from `first`, `nvalid` of them real) solved with
// LD = N/2 lanes per problem -- two coordinates, i.e. two rows of every matrix, per lane -- whatever lane mapping the
// diagonal arithmetic of the calling kernel uses: 64/LD problems per pass, TILE / (64/LD) passes.  Why not on the
// caller's mapping (round 2): with E = 4 rows of P and of M^-1 per lane the fused QP / QCQP kernels needed 242 / 250
// VGPRs, two waves per SIMD; the two forwards of a step could not be co-resident and the step paid ~4 us for it
// (DESIGN.md 3.1 (v)).  With two rows per lane the general solve fits the diagonal path's own budget (4 waves per
// SIMD); a dense tile costs two passes of roughly 0.65x the instructions each.  Reads q (and the constraint data)
// and writes x / iters itself, in the general solve's mapping.
template <int KIND, int N, int LD, int TILE, bool STAGED = false>
DQQ_D void group_dense_tile(const double* __restrict__ P, const double* __restrict__ q, const double* __restrict__ l_n,
                            const double* __restrict__ mu_c, const double* __restrict__ v_sign, double* __restrict__ x,
                            int* __restrict__ iters, long first, int nvalid, double eps, double mu, int max_iter,
                            int adaptive, int lane, int defer = 4, unsigned long long dmask = ~0ull, int mask_stride = 0,
                            const double* tile_lds = nullptr)
{
    // STAGED (LD == 1): the tile's matrices sit in LDS (tile_lds, stage_tile_lane8: row stride N*N + 1) -- P is not read from
    // memory here at all
    // dmask / mask_stride: only the problems j of the tile with bit j * mask_stride of dmask set are solved here (the caller's
    // ballot over ITS lanes, mask_stride lanes per problem); mask_stride = 0: all of them
    constexpr int E = N / LD, PPP = 64 / LD; // coordinates per lane, problems per pass
    static_assert(E >= 2 && E % 2 == 0 && TILE % PPP == 0, "whole contacts per lane, whole passes per tile");
    constexpr int EB = (KIND >= 2) ? E : 1;
    const int s = lane % LD;
#pragma unroll 1
    for (int pass = 0; pass < TILE / PPP; ++pass) {
        const int pj = pass * PPP + lane / LD;
        if (pass * PPP >= nvalid) break;            // wave-uniform
        const bool valid = pj < nvalid && (mask_stride == 0 || ((dmask >> (pj * mask_stride)) & 1ull) != 0);
        if (!__any(valid)) continue;                // (none of this pass's problems is non-diagonal)
        const long prob = first + pj;
        double qv[E], xv[E], rad[E / 2], lo[EB], hi[EB], sg[EB];
#pragma unroll
        for (int e = 0; e < E; e += 2) {
            const double2 t = valid ? *reinterpret_cast<const double2*>(q + prob * N + s * E + e) : make_double2(0.0, 0.0);
            qv[e] = t.x; qv[e + 1] = t.y;
        }
#pragma unroll
        for (int c = 0; c < E / 2; ++c) {
            const long co = prob * (N / 2) + s * (E / 2) + c;
            rad[c] = (KIND == 1 && valid) ? l_n[co] * mu_c[co] : (KIND == 1 ? 1.0 : 0.0); // pybindings.cpp:57
        }
        if constexpr (KIND >= 2) {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const long bo = prob * N + s * E + e;
                lo[e] = valid ? l_n[bo] : 0.0;
                hi[e] = valid ? mu_c[bo] : 0.0;
                sg[e] = 0.0;
                if (KIND == 3) { const double c = valid ? v_sign[bo] : 0.0; sg[e] = (double)((c > 0) - (c < 0)); } // :395
            }
        }
        int it;
        if constexpr (STAGED) {
            static_assert(LD == 1, "the staged tile: a lane per problem");
            it = group_dense_fwd<KIND, N, LD, true>(tile_lds + pj * (N * N + 1), qv, rad, eps, mu, max_iter, adaptive, valid,
                                                    xv, lo, hi, sg, defer);
        } else {
            it = group_dense_fwd<KIND, N, LD>(P + prob * (long)(N * N), qv, rad, eps, mu, max_iter, adaptive, valid, xv, lo,
                                              hi, sg, defer);
        }
        if (valid) {
#pragma unroll
            for (int e = 0; e < E; e += 2)
                *reinterpret_cast<double2*>(x + prob * N + s * E + e) = make_double2(xv[e], xv[e + 1]);
            if (iters != nullptr && s == 0) iters[prob] = it;
        }
    }
}

#endif // __HIPCC__

} // namespace dqq
