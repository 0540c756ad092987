// fwd_lane_dense.hip -- general (dense P) forward solve for small N (2, 4, 6, 8): ONE LANE PER
// PROBLEM, everything in VGPRs, no LDS, no barriers.
//
// A contact problem with 4 contacts has a dense 8x8 Delassus matrix P; one wave64 per problem
// (dense_core.h) leaves 56 of 64 lanes idle on it.  Here each lane holds its problem's whole state --
// the 8x8 explicit inverse of P + (rho+mu) I (64 doubles), the lower triangle of P (36), the ADMM
// vectors -- in registers (one wave per SIMD, ~400 VGPRs), every loop is fully unrolled with static
// register indices, and a wave advances 64 problems per instruction.
//
// Algorithm and operation order are the reference's (Solver::solveQP / solveQCQP, Solver.cpp:61-123,
// 521-582; power_iteration :46-59; left-looking LLT and column-wise substitutions for the explicit
// inverse, :76-77): same rho / tau / cpt state machine and stopping tests as admm_core.h, which this
// file mirrors with a dense mat-vec in place of the diagonal scale.  Ulp-level departures as in the
// diagonal fast path: FMA contraction, reciprocal-multiply instead of divide (1-ulp rcp / rsqrt).
// The rho-update branch (refactorisation, ~700 instructions) runs under the exec mask of the lanes
// that fire in that iteration.
#include <algorithm>
#include <atomic>

#include "admm_core.h"
#include "launch.h"

namespace dqq {

// Knob lane_defer (tuning.h): the lane-per-problem kernel runs the refactorisation of the lanes that changed rho every this
// many trips of its loop (1 = in the trip of the change, as rounds 1-2 did; 0 = built-in choice per kind).  Results do
// not depend on it.  The group solve of the fused forward (group_dense.h) takes the same option.
// Dense 8 x 8 (P = S S^T/8 + 0.1 I), QP / QCQP forward, us (tools/probe_lane_defer.py):
//   B = 65536    1: 76.7 / 89.2   2: 65.2 / 80.6   3: 61.0 / 79.6   4: 61.3 / 76.8   6: 58.2 / 78.6   8: 61.0 / 78.4   12: 65.9 / 86.7
//   B = 262144   1: 255 / 275     2: 211 / 246     3: 193 / 236     4: 194 / 234     6: 185 / 247     8: 190 / 248     12: 202 / 270
// (N = 4: 40.7 / 36.1 -> 38.6 / 34.9).  A model of the wave -- 190 instructions per trip, 440 per refactorisation, the
// firing pattern of the reference's rho schedule -- predicts 0.66 / 0.73 of the loop's cost at 4.
// 0 = built-in: 4 for the QCQP, 6 for the QP-like kinds (sweeps above and in tools/probe_group_defer.py; box / signed box
// QP, dense 8 x 8, B = 65536: 1: 79.7 / 80.5   2: 70.7 / 73.4   4: 69.0 / 71.2   6: 65.6 / 68.1   8: 68.0 / 69.9 us)
int lane_defer_for(int kind)
{
    const int v = knob_lane_defer();
    return v > 0 ? (v < 64 ? v : 64) : (kind == kKindQCQP ? 4 : 6);
}

// Explicit inverse of the symmetric matrix whose strict lower triangle is Plow and whose diagonal is
// d: lower Cholesky (left-looking, as Eigen's unblocked LLT), the lower triangular L^-1 by forward substitution
// (column c: L y = e_c, rows >= c), then M^-1 = L^-T L^-1 on its lower half, mirrored (the compiler keeps ONE register
// per symmetric pair).  Round 2 ran a backward substitution per column instead: 408 instead of 240 multiply-adds
// at N = 8 -- and this routine is what a wave of 64 problems executes, under a partial mask, whenever ANY of its
// lanes changes rho: ~25 times per wave on the dense 8 x 8 family of the bench, 60 % of the kernel's instructions.
template <int N>
DQQ_D void lane_chol_inverse(const double (&Plow)[N][N], const double (&d)[N], double (&Minv)[N][N], bool& bad)
{
    double L[N][N], rinv[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < k; ++j) s += L[k][j] * L[k][j];
        const double piv = d[k] - s;
        bad = bad || !(piv > 0.0);
        const double rs = fast_rsqrt(piv);
        L[k][k] = piv * rs;
        rinv[k] = rs;
#pragma unroll
        for (int i = k + 1; i < N; ++i) {
            double t = 0.0;
#pragma unroll
            for (int j = 0; j < k; ++j) t += L[i][j] * L[k][j];
            L[i][k] = (Plow[i][k] - t) * rs;
        }
    }
    double Li[N][N]; // L^-1, lower triangle
#pragma unroll
    for (int c = 0; c < N; ++c) {
#pragma unroll
        for (int i = c; i < N; ++i) {
            double t = (i == c) ? 1.0 : 0.0;
#pragma unroll
            for (int j = c; j < i; ++j) t -= L[i][j] * Li[j][c];
            Li[i][c] = t * rinv[i];
        }
    }
#pragma unroll
    for (int c = 0; c < N; ++c) {
#pragma unroll
        for (int i = c; i < N; ++i) {
            double t = Li[i][i] * Li[i][c];
#pragma unroll
            for (int k = i + 1; k < N; ++k) t += Li[k][i] * Li[k][c];
            Minv[i][c] = t;
            Minv[c][i] = t;
        }
    }
}

// The Cholesky factor alone (lower triangle incl. the diagonal, and the reciprocal pivots): the x-update then runs two
// triangular substitutions instead of a product with the explicit inverse (DQQ_LANE_FWD_FACTORED).  With a lane per problem
// the substitutions are 56 multiply-adds + 16 products against the 64 multiply-adds of the explicit inverse -- no
// cross-lane stages as in the wave-per-problem kernels -- while a refactorisation drops from ~440 to ~150
// instructions, and the wave runs one whenever any of its 64 lanes changed rho (on half of its trips).
template <int N>
DQQ_D void lane_chol(const double (&Plow)[N][N], const double (&d)[N], double (&L)[N][N], double (&rinv)[N], bool& bad)
{
#pragma unroll
    for (int k = 0; k < N; ++k) {
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < k; ++j) s += L[k][j] * L[k][j];
        const double piv = d[k] - s;
        bad = bad || !(piv > 0.0);
        const double rs = fast_rsqrt(piv);
        L[k][k] = piv * rs;
        rinv[k] = rs;
#pragma unroll
        for (int i = k + 1; i < N; ++i) {
            double t = 0.0;
#pragma unroll
            for (int j = 0; j < k; ++j) t += L[i][j] * L[k][j];
            L[i][k] = (Plow[i][k] - t) * rs;
        }
    }
}

#ifndef DQQ_LANE_FWD_FACTORED
#define DQQ_LANE_FWD_FACTORED 1
#endif

// The power-iteration vector is normalised after every step, like the reference (Solver.cpp:53), by a 1-ulp
// reciprocal square root: lambda_max^10 between two normalisations would leave the double range for
// |lambda_max| beyond ~1e15 or below ~1e-15, which the diagonal fast path (exact power-of-two scaling) handles.
template <int KIND, int N>
__global__ __launch_bounds__(64, 1) void fwd_lane_dense_kernel(const double* __restrict__ P,
                                                               const double* __restrict__ q,
                                                               const double* __restrict__ l_n,
                                                               const double* __restrict__ mu_c,
                                                               const double* __restrict__ v_sign, double* __restrict__ x,
                                                               long B, double eps, double mu, int max_iter,
                                                               int adaptive, int* __restrict__ iters,
                                                               int* __restrict__ ws, int use_worklist, int defer)
{
    // KIND 2 / 3 (box / signed box QP, Solver.cpp:198-261 / 374-439): l_n = l_min, mu_c = l_max per coordinate
    static_assert(N % 2 == 0, "even N");
    constexpr bool QP_LIKE = (KIND != 1);
    if (use_worklist && ws[kWsCount] == 0) return;   // an empty list: one scalar load, before the hygiene checks of launch.h
    const long count = use_worklist ? worklist_checked_count(ws, ws + kWsCount, kWsEntryInts(B)) : B;
    const long slot = (long)blockIdx.x * 64 + threadIdx.x;
    const bool valid = slot < count;
    if ((long)blockIdx.x * 64 >= count) { // a wave beyond the end of the work-list: only the reset ticket
        if (use_worklist && threadIdx.x == 0) worklist_release(ws, count, (int)gridDim.x);
        return;
    }
    const long prob = valid ? (use_worklist ? worklist_checked_entry(ws, ws[kWsEntries + slot], B) : slot) : 0;

    // ---- load: P (lower triangle kept, full matrix used by the power iteration), q, radius
    double Pm[N][N];
    {
        const double* Pg = P + prob * (long)(N * N);
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = 0; j < N; j += 2) {
                const double2 t = valid ? *reinterpret_cast<const double2*>(Pg + i * N + j) : make_double2(0.0, 0.0);
                Pm[i][j] = t.x;
                Pm[i][j + 1] = t.y;
            }
        if (!valid) {
#pragma unroll
            for (int i = 0; i < N; ++i) Pm[i][i] = 1.0;
        }
    }
    double qv[N], rad[N / 2];
#pragma unroll
    for (int i = 0; i < N; i += 2) {
        const double2 t = valid ? *reinterpret_cast<const double2*>(q + prob * N + i) : make_double2(0.0, 0.0);
        qv[i] = t.x;
        qv[i + 1] = t.y;
    }
#pragma unroll
    for (int c = 0; c < N / 2; ++c)
        rad[c] = (KIND == 1 && valid) ? l_n[prob * (N / 2) + c] * mu_c[prob * (N / 2) + c] : 1.0;
    constexpr int NB = (KIND >= 2) ? N : 1;
    double blo[NB], bhi[NB], bsg[NB];
    if constexpr (KIND >= 2) {
#pragma unroll
        for (int i = 0; i < N; i += 2) {
            const double2 a = valid ? *reinterpret_cast<const double2*>(l_n + prob * N + i) : make_double2(0.0, 0.0);
            const double2 b = valid ? *reinterpret_cast<const double2*>(mu_c + prob * N + i) : make_double2(0.0, 0.0);
            blo[i] = a.x; blo[i + 1] = a.y;
            bhi[i] = b.x; bhi[i + 1] = b.y;
            bsg[i] = bsg[i + 1] = 0.0;
            if (KIND == 3) {
                const double2 c = valid ? *reinterpret_cast<const double2*>(v_sign + prob * N + i) : make_double2(0.0, 0.0);
                bsg[i] = (double)((c.x > 0) - (c.x < 0));                      // cwiseSign, :395
                bsg[i + 1] = (double)((c.y > 0) - (c.y < 0));
            }
        }
    }

    // ---- power_iteration, Solver.cpp:46-59
    double Lmax;
    {
        double v[N];
        const double c0 = 1.0 / sqrt((double)N);
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < N; ++i) { v[i] = c0; s += c0 * c0; }
        if (s > 0) {
            const double nn = sqrt(s);
#pragma unroll
            for (int i = 0; i < N; ++i) v[i] = v[i] / nn;
        }
        if constexpr (QP_LIKE) {
            for (int k = 0; k < 10; ++k) {
                double Av[N];
                s = 0.0;
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    double t = 0.0;
#pragma unroll
                    for (int j = 0; j < N; ++j) t += Pm[i][j] * v[j];
                    Av[i] = t;
                    s += t * t;
                }
                // normalised every step like the reference (Solver.cpp:53): lambda_max^10 between two
                // normalisations would overflow / underflow for |lambda_max| beyond ~1e15 / below ~1e-15
                const double inv = s > 0 ? fast_rsqrt(s) : 1.0;
#pragma unroll
                for (int i = 0; i < N; ++i) v[i] = Av[i] * inv;
            }
        } else {
            // 100 steps: P^100 v0 = P^64 (P^32 (P^4 v0)) -- six matrix squarings (N mat-vecs' worth each) and three
            // mat-vecs instead of a hundred mat-vecs.  Every squared matrix and vector is rescaled by an exact
            // power of two (taken from the largest diagonal entry, which bounds every entry of a power of a
            // symmetric positive definite matrix), so nothing overflows and no rounding is added by the scaling.
            double Q[N][N];
#pragma unroll
            for (int i = 0; i < N; ++i)
#pragma unroll
                for (int j = 0; j < N; ++j) Q[i][j] = Pm[i][j];
#pragma unroll 1
            for (int sq = 1; sq <= 6; ++sq) {
                double T2[N][N];
                double dmax = 0.0;
#pragma unroll
                for (int i = 0; i < N; ++i) {
#pragma unroll
                    for (int j = 0; j < N; ++j) {
                        double t = 0.0;
#pragma unroll
                        for (int k = 0; k < N; ++k) t += Q[i][k] * Q[k][j];
                        T2[i][j] = t;
                    }
                    dmax = fmax(dmax, fabs(T2[i][i]));
                }
                int e = 0;
                if (dmax > 0.0 && dmax < 1.79e308) (void)frexp(dmax, &e);
#pragma unroll
                for (int i = 0; i < N; ++i)
#pragma unroll
                    for (int j = 0; j < N; ++j) Q[i][j] = ldexp(T2[i][j], -e);
                if (sq == 2 || sq == 5 || sq == 6) { // Q = P^4, P^32, P^64 (up to a power of two)
                    double Av[N];
                    double vm = 0.0;
#pragma unroll
                    for (int i = 0; i < N; ++i) {
                        double t = 0.0;
#pragma unroll
                        for (int j = 0; j < N; ++j) t += Q[i][j] * v[j];
                        Av[i] = t;
                        vm = fmax(vm, fabs(t));
                    }
                    int ev = 0;
                    if (vm > 0.0 && vm < 1.79e308) (void)frexp(vm, &ev);
#pragma unroll
                    for (int i = 0; i < N; ++i) v[i] = ldexp(Av[i], -ev);
                }
            }
            s = 0.0;
#pragma unroll
            for (int i = 0; i < N; ++i) s += v[i] * v[i];
            const double inv = (s > 0) ? fast_rsqrt(s) : 1.0;
#pragma unroll
            for (int i = 0; i < N; ++i) v[i] = v[i] * inv;
        }
        Lmax = 0.0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            double t = 0.0;
#pragma unroll
            for (int j = 0; j < N; ++j) t += Pm[i][j] * v[j];
            Lmax += v[i] * t;
        }
    }

    // ---- Solver.cpp:72-77 / 531-536
    RhoSchedule sched;
    sched.init(Lmax, mu);
    double rho = sched.rho;
    double inv_rho = fast_rcp(rho);
    bool bad = !(rho > 0.0) || !(rho < 1.79e308);
    double md[N];
#if DQQ_LANE_FWD_FACTORED
    double Lf[N][N], rinv[N];
#else
    double Minv[N][N];
#endif
#pragma unroll
    for (int i = 0; i < N; ++i) md[i] = Pm[i][i] + (rho + mu);   // the accumulated shifted diagonal
#if DQQ_LANE_FWD_FACTORED
    lane_chol<N>(Pm, md, Lf, rinv, bad);                           // only the lower triangle of Pm is read
#else
    lane_chol_inverse<N>(Pm, md, Minv, bad);
#endif

    double qp[N], l2[N], u[N];
#pragma unroll
    for (int i = 0; i < N; ++i) { qp[i] = qv[i]; l2[i] = 0.0; u[i] = 0.0; }

    // The refactorisation after a rho update is DEFERRED: a lane that changes rho (rho, 1/rho and the shifted diagonal
    // are updated on the spot) sits out until the wave next runs the refactorisation -- every `defer`-th trip, or as
    // soon as no lane has anything else to do -- so that one pass of those ~450 instructions serves the lanes that fired
    // over several trips (a trip of the loop costs the wave the same ~190 instructions whether 64 lanes take part or 6).
    // On the dense 8 x 8 family of the bench a wave ran the refactorisation on 25 of its 47 trips (QP; QCQP 18 of 36).
    // A lane's own arithmetic, and so its result, does not depend on `defer`.
    int it_done = 0;
    bool done = !valid || max_iter <= 0, pend = false;
    for (int trip = 0;; ++trip) {
        if (!done && !pend) {
            double rhs[N], w[N], z[N];
            double rd = 0.0, rp = 0.0, nl = 0.0;
#pragma unroll
            for (int i = 0; i < N; ++i) rhs[i] = rho * l2[i] - u[i] - qp[i];
#if DQQ_LANE_FWD_FACTORED
            double lsol[N];
#pragma unroll
            for (int i = 0; i < N; ++i) {                                         // L y = rhs
                double t = rhs[i];
#pragma unroll
                for (int j = 0; j < i; ++j) t -= Lf[i][j] * lsol[j];
                lsol[i] = t * rinv[i];
            }
#pragma unroll
            for (int i = N - 1; i >= 0; --i) {                                    // L^T l = y
                double t = lsol[i];
#pragma unroll
                for (int j = i + 1; j < N; ++j) t -= Lf[j][i] * lsol[j];
                lsol[i] = t * rinv[i];
            }
#endif
#pragma unroll
            for (int i = 0; i < N; ++i) {
#if DQQ_LANE_FWD_FACTORED
                const double l = lsol[i];                                         // :80 / :539
#else
                double l = 0.0;
#pragma unroll
                for (int j = 0; j < N; ++j) l += Minv[i][j] * rhs[j];             // :80 / :539
#endif
                qp[i] = qv[i] - mu * l;                                           // :81 / :540
                w[i] = kAlpha * l + (1 - kAlpha) * l2[i];
                z[i] = w[i] + u[i] * inv_rho;                                     // :82 / :541
                if (KIND == 1) nl += l * l;
            }
            if (KIND == 0) {
#pragma unroll
                for (int i = 0; i < N; ++i) z[i] = fmax(z[i], 0.0);
            } else if constexpr (KIND >= 2) {
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    double t = z[i];
                    t = t < blo[i] ? blo[i] : t;                                  // cwiseMax(l_min), :219 / :396
                    t = bhi[i] < t ? bhi[i] : t;                                  // cwiseMin(l_max), :220 / :397
                    if (KIND == 3) {                                              // v o min(v o l_2, 0), :398
                        double m = bsg[i] * t;
                        m = 0 < m ? 0 : m;
                        t = bsg[i] * m;
                    }
                    z[i] = t;
                }
            } else {
#pragma unroll
                for (int c = 0; c < N / 2; ++c) {                                 // prox_circle, :505-519
                    const double a = z[2 * c], b = z[2 * c + 1];
                    const double n2 = __builtin_fma(b, b, a * a);
                    const double rn = fast_rsqrt(n2);
                    // |l_(c)| > r tested on the squares, |l|^2 > r |r|, with the SAME expressions as the diagonal / group body
                    // (admm_diag_body.inc: one prox_circle for every forward that seeds its rsqrt -- ADVICE r5: which kernel
                    // serves a DQQ_P_DENSE N = 8 batch depends on B, the projection rule must not)
                    if (n2 > rad[c] * fabs(rad[c])) {
                        const double sc = rad[c] * rn;
                        z[2 * c] = a * sc;
                        z[2 * c + 1] = b * sc;
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < N; ++i) {
                u[i] += rho * (w[i] - z[i]);                                      // :83 / :543
                rd = fmax(rd, fabs(z[i] - l2[i]));
                rp = fmax(rp, fabs(z[i] - w[i]));
                l2[i] = z[i];
            }
            const double res_dual = rho * rd, res_prim = rp;
            it_done += 1;
            bool stop = res_dual < eps;                                           // :88
            if (KIND == 1) {
                if (stop) stop = res_prim < eps + kEpsRel * sqrt(nl);             // :548
            }
            done = stop || it_done >= max_iter;
            if (!done && adaptive) {
                double delta;
                if (sched.template update<QP_LIKE, true>(res_prim, res_dual, delta)) { // Solver.cpp:90-120 / 550-580
                    rho = sched.rho;
                    inv_rho = fast_rcp(rho);
#pragma unroll
                    for (int i = 0; i < N; ++i) md[i] += delta;
                    pend = true;
                }
            }
        }
        if (__all(done)) break;
        if (__any(pend) && ((trip + 1) % defer == 0 || !__any(!done && !pend))) {
#if DQQ_LANE_FWD_FACTORED
            if (pend) lane_chol<N>(Pm, md, Lf, rinv, bad);                        // llt(); solveInPlace(Identity) is not formed
#else
            if (pend) lane_chol_inverse<N>(Pm, md, Minv, bad);                    // llt() + solveInPlace(Identity)
#endif
            pend = false;
        }
    }

    if (valid) {
#pragma unroll
        for (int i = 0; i < N; i += 2)
            *reinterpret_cast<double2*>(x + prob * N + i) =
                make_double2(bad ? NAN : l2[i], bad ? NAN : l2[i + 1]);
        if (iters != nullptr) iters[prob] = it_done;
    }
    if (use_worklist && threadIdx.x == 0) worklist_release(ws, count, (int)gridDim.x);
}

template <int KIND, int N>
static hipError_t launch_lane(const FwdArgs& a, bool use_worklist, hipStream_t s)
{
    const long nw = (a.B + 63) / 64;
    if (nw == 0) return hipSuccess;
    return launch((fwd_lane_dense_kernel<KIND, N>), dim3((unsigned)nw), dim3(64), 0, s, a.P, a.q, a.l_n, a.mu, a.v, a.x,
                       a.B, a.eps, a.mu_prox, a.max_iter, a.adaptive, a.iters, a.ws, use_worklist ? 1 : 0,
                       lane_defer_for(KIND));
}

bool fwd_lane_dense_supported(int N) { return N == 2 || N == 4 || N == 6 || N == 8; }

hipError_t launch_fwd_lane_dense(int kind, const FwdArgs& a, bool use_worklist, hipStream_t s)
{
#define DQQ_CASE(NN)                                                           \
    if (a.N == NN) {                                                           \
        switch (kind) {                                                        \
        case 0: return launch_lane<0, NN>(a, use_worklist, s);                 \
        case 1: return launch_lane<1, NN>(a, use_worklist, s);                 \
        case 2: return launch_lane<2, NN>(a, use_worklist, s);                 \
        case 3: return launch_lane<3, NN>(a, use_worklist, s);                 \
        default: return hipErrorInvalidValue;                                  \
        }                                                                      \
    }
    DQQ_CASE(2) DQQ_CASE(4) DQQ_CASE(6) DQQ_CASE(8)
#undef DQQ_CASE
    return hipErrorInvalidValue;
}

} // namespace dqq
