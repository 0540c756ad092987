// bwd_lane_dense.hip -- general (dense P) backward for small N (2, 4, 6, 8), P declared dense (DQQ_P_DENSE): ONE LANE PER
// PROBLEM.
//
// A contact problem with 4 contacts has a dense 8 x 8 Delassus matrix, and its backward -- the (N + N/2)-unknown
// Tikhonov system of Solver::solveDerivativesQCQP (Solver.cpp:619-681) through iterative_refinement (:15-44) -- is the
// larger half of its step.  The team kernel (bwd_small.hip: 16 lanes per problem, lane = unknown) spends 554 wave
// instructions per problem on it: 12 of 16 lanes work, every triangular loop runs to its full length on every lane, and
// what a lane needs from its neighbours goes through LDS.  Here a lane owns a whole problem: the triangular loops are
// triangular, structural zeros (the contact-contact block of A A^T is diagonal: contacts couple only through the
// coordinates) cost nothing, and a wave advances 64 problems per instruction -- about 150 wave instructions per problem.
//
// Arithmetic: exactly small_bwd_core.h's, i.e. the reference's dense operation order (pybindings.cpp:62-71 ->
// Solver::dualFromPrimalQCQP :584-617, solveDerivativesQCQP :619-681, iterative_refinement :15-44, getE12QCQP :683-691;
// QP: :125-196): every sum runs sequentially in index order from 0.0, products are rounded before they are added
// (-ffp-contract=off), divisions and square roots are IEEE.  Terms with a structurally zero factor are left out: they are
// +-0, and a running sum that started at +0.0 is never -0.0, so leaving them out changes no bit (finite data).  The
// results are bit-identical to the team kernel's and to the oracle's on identical x (tests/test_gpu_parity.py).
//
// Storage per lane (QCQP, N = 8: M = 12 unknowns): the factor L (72 doubles + 12 reciprocal pivots) and the explicit inverse
// K^-1 (144: its two triangles are computed by different operation sequences and are not bitwise symmetric, so both are
// kept) do not fit 512 registers together, K itself (72) is read again by every refinement body, and a lane owns 336 doubles
// (512 registers + 640 B of LDS at four waves per CU).  Any spill is poison at one wave per SIMD (see lane_ir).  So: K is
// built into LDS (lane-interleaved: element e of lane l at (e * 64 + l) * 8, conflict-free) for the factorisation; while the
// inverse is formed its slots take the first six columns of the inverse; then K is built a SECOND time, column by column of P
// straight from L2, into the registers the dead factor leaves (qcqp_rebuild_K: the same sums in the same order, the same bits).
// A^T b waits in the 8 slots that are left + 4 registers.  The first refinement body (x = 0) multiplies nothing.
#include <utility>

#include "kkt_core.h"
#include "launch.h"

namespace dqq {

namespace {

template <int KIND, int N>
struct LaneSys {
    static constexpr int NC = N / 2;
    static constexpr int M = (KIND == 1) ? N + NC : N;
    // structural zero of K = A A^T + mu I and of its Cholesky factor, i > j: two different contacts (QCQP)
    static constexpr bool kz(int i, int j) { return KIND == 1 && i < NC && j < NC && i != j; }
    // slot of K[i][j], i >= j, in the lane's LDS column (structural zeros take none: a contact row holds its diagonal)
    static constexpr int slot(int i, int j)
    {
        return (KIND == 1) ? (i < NC ? i : NC + i * (i + 1) / 2 - NC * (NC + 1) / 2 + j) : i * (i + 1) / 2 + j;
    }
    static constexpr int SLOTS = slot(M - 1, M - 1) + 1;
    // What does not fit the registers waits in the lane's LDS column.  Four waves per CU (one per SIMD) leave a lane 80
    // slots (640 B); the QCQP at N = 8 takes 160 (TWO waves per CU, see lane_ir): K, then A^T b, then the first PARK
    // columns of the inverse.  A^T b is not touched while the inverse is formed: its LAST AB_LDS entries go to LDS, the
    // first M - AB_LDS stay in registers.
#ifndef DQQ_LANE_REGEN
#define DQQ_LANE_REGEN 1
#endif
    // REGEN (the QCQP at N = 8): K is needed by the factorisation and, much later, by the refinement bodies -- in between, while
    // the inverse is formed, its 72 LDS slots hold the first PARK = 6 columns of the inverse, and K is built a SECOND time
    // (same code, same bits: ~1000 instructions and a re-read of the inputs out of L2) into the registers the dead factor
    // leaves behind.  80 slots per lane = four waves per CU; with K kept (WIDE: 160 slots) only two waves fit a CU.
    static constexpr bool REGEN = (DQQ_LANE_REGEN != 0) && (KIND == 1 && N == 8);
    static constexpr bool WIDE = (KIND == 1 && N == 8) && !REGEN;
    static constexpr int CAP = WIDE ? 160 : 80;
    static constexpr int AB_LDS = (CAP - SLOTS) < M ? (CAP - SLOTS) : M;
    static constexpr int AB_REG = M - AB_LDS;
    static constexpr int PARK = WIDE ? (CAP - SLOTS - AB_LDS) / M : (REGEN ? SLOTS / M : 0);   // columns of K^-1 kept in LDS
    static constexpr int PARK0 = REGEN ? 0 : SLOTS + AB_LDS;              // their first slot (REGEN: over the dead K)
    static constexpr int LDS_SLOTS = REGEN ? SLOTS + AB_LDS : SLOTS + AB_LDS + PARK * M;
};

// K in LDS: element e of this lane
#define DQQ_KL(e) kl[(e) * 64]

// a / b, correctly rounded, from r = RN(1 / b) (the result of an IEEE division): the product a r corrected twice by its
// exact residual.  After the first correction the quotient is within an ulp; the second is Markstein's step, which
// yields the correctly rounded quotient when r is the correctly rounded reciprocal.  5 FP64 instructions where the IEEE
// sequence (scale, seed, Newton steps, fix-ups) takes ~11 -- and the factorisation divides ~280 times per problem by
// only M different pivots.  tools/ubench/div_by_recip.hip: 2^34 pairs incl. quotients next to rounding midpoints, no
// mismatch against a / b.  (Zero results come out as +0 where IEEE gives -0; nothing here depends on a zero's sign.
// Pivots are sqrt(K_kk - ...) with K = A A^T + 1e-7 I: never 0 / inf on finite data; on non-finite data both forms
// give non-finite results.)
#ifndef DQQ_LANE_FASTDIV
#define DQQ_LANE_FASTDIV 1
#endif
static DQQ_D double div_by(double a, double b, double r)
{
#if DQQ_LANE_FASTDIV
    double q = a * r;
    double e = __builtin_fma(-b, q, a);
    q = __builtin_fma(e, r, q);
    e = __builtin_fma(-b, q, a);
    return __builtin_fma(e, r, q);
#else
    (void)r;
    return a / b;
#endif
}

#ifndef DQQ_LANE_PAIR
#define DQQ_LANE_PAIR 2
#endif
// Columns C0 .. C0 + W - 1 of K^-1 by solveInPlace(Identity) (Solver.cpp:22-23): forward, then backward substitution, the
// W columns side by side -- a triangular solve is one long dependent chain (66 adds + 12 quotients per substitution at
// M = 12), and with one wave per SIMD a second, independent chain is what fills the FP64 pipeline's latency.  C0 is a
// template argument: as a loop over the columns the compiler left this rolled and put the factor in scratch memory.
// Finished columns go to LDS (the first S::PARK) or to Kinv.
template <typename S, int C>
static DQQ_D __attribute__((always_inline)) void inv_fwd_row(const double (&L)[S::M][S::M], const double (&rcp)[S::M],
                                                             double (&y)[S::M], int i)
{
#pragma clang fp contract(off)
    if (i < C) {
        y[i] = 0.0;
    } else if (i == C) {
        y[i] = rcp[C];                                         // 1.0 / L[c][c]
    } else {
        double t = 0.0;
#pragma unroll
        for (int j = C; j < i; ++j)
            if (!S::kz(i, j)) t -= L[i][j] * y[j];
        y[i] = div_by(t, L[i][i], rcp[i]);
    }
}
template <typename S>
static DQQ_D __attribute__((always_inline)) void inv_bwd_row(const double (&L)[S::M][S::M], const double (&rcp)[S::M],
                                                             double (&y)[S::M], int i)
{
#pragma clang fp contract(off)
    double t = y[i];
#pragma unroll
    for (int j = i + 1; j < S::M; ++j)
        if (!S::kz(j, i)) t -= L[j][i] * y[j];
    y[i] = div_by(t, L[i][i], rcp[i]);
}
template <typename S, int C>
static DQQ_D __attribute__((always_inline)) void inv_keep(const double (&y)[S::M], double (&Kinv)[S::M][S::M], double* kl)
{
#pragma unroll
    for (int i = 0; i < S::M; ++i) {
        if (C < S::PARK) kl[(S::PARK0 + C * S::M + i) * 64] = y[i];
        else Kinv[i][C] = y[i];
    }
}
template <typename S, int C0>
static DQQ_D __attribute__((always_inline)) void lane_inverse_cols(const double (&L)[S::M][S::M], const double (&rcp)[S::M],
                                                                   double (&Kinv)[S::M][S::M], double* kl)
{
    constexpr int M = S::M;
    constexpr bool TWO = (DQQ_LANE_PAIR == 2) && (C0 + 1 < M);
    constexpr int C1 = TWO ? C0 + 1 : C0;
    double y0[M], y1[M];
#pragma unroll
    for (int i = 0; i < M; ++i) {
        inv_fwd_row<S, C0>(L, rcp, y0, i);
        if (TWO) inv_fwd_row<S, C1>(L, rcp, y1, i);
    }
#pragma unroll
    for (int i = M - 1; i >= 0; --i) {
        inv_bwd_row<S>(L, rcp, y0, i);
        if (TWO) inv_bwd_row<S>(L, rcp, y1, i);
    }
    inv_keep<S, C0>(y0, Kinv, kl);
    if (TWO) inv_keep<S, C1>(y1, Kinv, kl);
}
template <typename S, int... I>
static DQQ_D __attribute__((always_inline)) void lane_inverse_all(const double (&L)[S::M][S::M], const double (&rcp)[S::M],
                                                                  double (&Kinv)[S::M][S::M], double* kl,
                                                                  std::integer_sequence<int, I...>)
{
    (lane_inverse_cols<S, I * DQQ_LANE_PAIR>(L, rcp, Kinv, kl), ...);
}

// Solver::iterative_refinement (Solver.cpp:15-44) on K = A_t^T A_t + mu I (lower triangle in the lane's LDS column) and
// Ab = A_t^T b, in the operation order of small_bwd_core.h: team_ir.  Returns xs, the number of bodies in `steps`.
// Ab: entries [0, AB_REG) in abr, the rest in the lane's LDS column behind K.
struct NoRegen {
    template <typename T> DQQ_D void operator()(T&) const {}
};
template <typename S, typename Regen = NoRegen>
static DQQ_D void lane_ir(double* kl, const double (&abr)[S::AB_REG > 0 ? S::AB_REG : 1], double (&xs)[S::M],
                          int& steps, Regen regen = Regen())
{
#pragma clang fp contract(off)
    constexpr int M = S::M;
    // ---- llt(), :23 -- left-looking, column by column
    double L[M][M], rcp[M];   // rcp[k] = 1 / L[k][k] (IEEE): also entry (k, k) of column k of the forward substitution
#pragma unroll
    for (int k = 0; k < M; ++k) {
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < k; ++j)
            if (!S::kz(k, j)) s += L[k][j] * L[k][j];
        double xk = DQQ_KL(S::slot(k, k)) - s;
        xk = sqrt(xk);
        L[k][k] = xk;
        rcp[k] = 1.0 / xk;
#pragma unroll
        for (int i = k + 1; i < M; ++i) {
            if (S::kz(i, k)) continue;   // (0 - 0) / xk
            double t = 0.0;
#pragma unroll
            for (int j = 0; j < k; ++j)
                if (!S::kz(i, j) && !S::kz(k, j)) t += L[i][j] * L[k][j];
            L[i][k] = div_by(DQQ_KL(S::slot(i, k)) - t, xk, rcp[k]);
        }
    }
    // The factor (84 doubles with the reciprocal pivots) and the growing inverse (144) do not fit 512 registers together
    // (M = 12), and what the compiler then spills it reloads in the middle of dependent chains -- with ONE wave per SIMD
    // every such round trip to memory is paid in full: the first version of this kernel, 250 scratch accesses per lane,
    // took 123 us for 65536 problems where the N = 6 instantiation, which fits, takes 25.  The first PARK = 6 columns of the
    // inverse are parked in LDS as they are finished (the refinement bodies read them from there).  Where that LDS comes from:
    // WIDE -- 160 slots per lane, two waves per CU, K stays in LDS: 87 us per 65536; REGEN (default) -- the columns go over
    // the dead K's slots, four waves per CU, and K is rebuilt into registers after the inverse (`regen`): 69 us.
    constexpr int PARK = S::PARK;
    double Kinv[M][M];
    lane_inverse_all<S>(L, rcp, Kinv, kl, std::make_integer_sequence<int, (M + DQQ_LANE_PAIR - 1) / DQQ_LANE_PAIR>{});
    // REGEN: the factor is dead, K comes back -- into registers
    double Kreg[S::REGEN ? S::SLOTS : 1];
    if constexpr (S::REGEN) regen(Kreg);
#define DQQ_KENT(a, b) (S::REGEN ? Kreg[S::REGEN ? S::slot(a, b) : 0] : DQQ_KL(S::slot(a, b)))
    // entry (i, j) of the inverse: the parked columns from LDS
#define DQQ_KINV(i, j) ((j) < PARK ? DQQ_KL(S::PARK0 + (j) * M + (i)) : Kinv[i][j])
    // ---- K^-1 A^T b, :27 (the factor is dead: A^T b comes back from LDS)
    double Ab[M], KinvAb[M];
#pragma unroll
    for (int i = 0; i < M; ++i) Ab[i] = (i < S::AB_REG) ? abr[i < S::AB_REG ? i : 0] : DQQ_KL(S::SLOTS + i - S::AB_REG);
#pragma unroll
    for (int i = 0; i < M; ++i) {
        double t = 0.0;
#pragma unroll
        for (int j = 0; j < M; ++j) t += DQQ_KINV(i, j) * Ab[j];
        KinvAb[i] = t;
    }
    // ---- the refinement loop, :26-41.  Lanes leave it one by one (the wave runs the longest).
#pragma unroll
    for (int i = 0; i < M; ++i) xs[i] = 0.0;
    IrControl ctl;
    ctl.init();
    steps = 0;
    bool done = false;
    for (int it = 0; it < kIrMaxIter; ++it) {
        // K (and the parked columns of the inverse) are read from LDS in EVERY body: without this the loads are hoisted out
        // of the loop as invariants, into registers that are not there
        asm volatile("" : "+v"(kl));
        if (!done) {
            steps = it + 1;
            double xn[M];
            if (it == 0) {
                // x = 0: K^-1 x is a sum of +-0 (:29)
#pragma unroll
                for (int i = 0; i < M; ++i) xn[i] = kMuIr * 0.0 + KinvAb[i];
            } else {
#pragma unroll
                for (int i = 0; i < M; ++i) {
                    double tmp = 0.0;
#pragma unroll
                    for (int j = 0; j < M; ++j) tmp += DQQ_KINV(i, j) * xs[j];
                    xn[i] = kMuIr * tmp + KinvAb[i];
                }
            }
            double ss = 0.0;
#pragma unroll
            for (int i = 0; i < M; ++i) {
                xs[i] = xn[i];
            }
#pragma unroll
            for (int i = 0; i < M; ++i) {
                double d = 0.0;                                                 // :30
#pragma unroll
                for (int j = 0; j < M; ++j) {
                    const int a = i > j ? i : j, b = i > j ? j : i;
                    if (!S::kz(a, b)) d += DQQ_KENT(a, b) * xs[j];
                }
                d = d - Ab[i];
                ss += d * d;                                                    // :31
            }
            if (ctl.update(sqrt(ss))) done = true;                              // :32-41
        }
        if (__all(done)) break;
    }
#undef DQQ_KINV
#undef DQQ_KENT
}

// grad_P = -dl x^T (qcqp.py:49 / :174) of the wave's 64 problems, staged in LDS (the K area is dead) and streamed out with
// coalesced 16-byte stores -- the tile is contiguous in memory, a lane's own matrix is N*N*8 bytes from its neighbour's.
// LIST (the wave's problems come from a work-list): member pp's matrix sits at its own problem index, held by lane pp.
template <int N, bool LIST = false>
static DQQ_D void store_grad_P_tile(double* __restrict__ smem, double* __restrict__ grad_P, long first, int nvalid,
                                    const double (&dl)[N], const double (&xv)[N], int prob32 = 0)
{
#pragma clang fp contract(off)
    constexpr int NN = N * N, TS = NN + 1, CPP = NN / 2;
    __syncthreads();   // every read of K is done
    double* Gl = smem + (int)threadIdx.x * TS;
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) Gl[i * N + j] = -(dl[i] * xv[j]);
    __syncthreads();
    double* Gw = grad_P + first * (long)NN;
#pragma unroll
    for (int k = 0; k < CPP; ++k) {
        const int ch = k * 64 + (int)threadIdx.x;
        const int pp = ch / CPP, w = ch % CPP;
        double* dst = Gw + 2 * (long)ch;
        if constexpr (LIST) dst = grad_P + (long)__shfl(prob32, pp) * NN + 2 * w;
        if (pp < nvalid) {
            // written once, not read again by this launch: non-temporal
            __builtin_nontemporal_store(smem[pp * TS + 2 * w], dst);
            __builtin_nontemporal_store(smem[pp * TS + 2 * w + 1], dst + 1);
        }
    }
}

// A^T b, entry i (a compile-time constant after unrolling): the first AB_REG in registers, the rest in LDS behind K
template <typename S>
static DQQ_D void set_ab(double* __restrict__ kl, double (&abr)[S::AB_REG > 0 ? S::AB_REG : 1], int i, double v)
{
    if (i < S::AB_REG) abr[i < S::AB_REG ? i : 0] = v;
    else DQQ_KL(S::SLOTS + i - S::AB_REG) = v;
}

} // namespace

// dualFromPrimalQCQP (:584-617) and the active set of solveDerivativesQCQP (:622-641), contact by contact;
// row c of A = [[diag(S), diag(gamma) C^T],[C, P + blkdiag(2 gamma_i I2)]] (:643-657): aS at column c, aA / aB at
// the contact's two coordinate columns; an inactive contact's row is zero
template <int N>
static DQQ_D __attribute__((always_inline)) void qcqp_contacts(const double (&xv)[N], const double (&plq)[N],
                                                               const double (&lnv)[N / 2], const double (&mcv)[N / 2],
                                                               double dual_eps, double (&gam)[N / 2], bool (&cact)[N / 2],
                                                               double (&aS)[N / 2], double (&aA)[N / 2], double (&aB)[N / 2])
{
#pragma clang fp contract(off)
    constexpr int NC = N / 2;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const double ln = lnv[c], mc = mcv[c];
        const double r = ln * mc;                                           // pybindings.cpp:65
        const double xa = xv[2 * c], xb = xv[2 * c + 1];
        double g0 = 0.0;
        const double slack = r + -sqrt(xa * xa + xb * xb);
        if (!(slack > dual_eps || r < dual_eps)) {
            const double ca = 2 * xa, cb = 2 * xb;
            const double G = ca * ca + cb * cb;
            const double rhs = ca * plq[2 * c] + cb * plq[2 * c + 1];
            const double Lg = sqrt(G);
            g0 = -((rhs / Lg) / Lg);
        }
        double Sc = -(r * r);
        Sc = Sc + (xa * xa + xb * xb);
        cact[c] = Sc > -kActiveEps && r > kActiveEps;
        gam[c] = g0;
        aS[c] = cact[c] ? Sc : 0.0;
        aA[c] = cact[c] ? g0 * (2 * xa) : 0.0;
        aB[c] = cact[c] ? g0 * (2 * xb) : 0.0;
    }
}

// K of the same system once more (REGEN), straight from memory and COLUMN by column of P: an entry of K is a sum over the
// columns m = 0 .. N-1 in ascending order whichever way the loops nest, so accumulating all entries while the columns stream by
// gives the bits of qcqp_system's row-wise sums -- with 8 doubles of P live instead of 64 (the inverse's registers are full).
template <typename S, int N>
static DQQ_D __attribute__((always_inline)) void qcqp_rebuild_K(const double* __restrict__ Pg, const double (&xv)[N],
                                                                const double (&qv)[N], const double (&lnv)[N / 2],
                                                                const double (&mcv)[N / 2], double dual_eps, double (&gam)[N / 2],
                                                                bool (&cact)[N / 2], double (&Kreg)[S::SLOTS])
{
#pragma clang fp contract(off)
    constexpr int NC = N / 2;
    double plq[N];
#pragma unroll
    for (int i = 0; i < N; ++i) plq[i] = 0.0;
#pragma unroll
    for (int j = 0; j < N; ++j)
#pragma unroll
        for (int i = 0; i < N; ++i) plq[i] += Pg[i * N + j] * xv[j];            // (P l)_i, j ascending, :606
#pragma unroll
    for (int i = 0; i < N; ++i) plq[i] = plq[i] + qv[i];
    double aS[NC], aA[NC], aB[NC], cx[N];
    qcqp_contacts<N>(xv, plq, lnv, mcv, dual_eps, gam, cact, aS, aA, aB);
#pragma unroll
    for (int i = 0; i < N; ++i) cx[i] = cact[i / 2] ? 2 * xv[i] : 0.0;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        double t = 0.0;
        t += aS[c] * aS[c];
        t += aA[c] * aA[c];
        t += aB[c] * aB[c];
        t += kMuIr;
        Kreg[S::slot(c, c)] = t;
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            double t = 0.0;
            if (i / 2 == c) t += cx[i] * aS[c];
            Kreg[S::slot(NC + i, c)] = t;
        }
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            double t = 0.0;
            if (i / 2 == j / 2) t += cx[i] * cx[j];
            Kreg[S::slot(NC + i, NC + j)] = t;
        }
    }
#pragma unroll
    for (int m = 0; m < N; ++m) {
        double col[N];                                                          // column m of D = P + blkdiag(2 gamma_i I2)
#pragma unroll
        for (int i = 0; i < N; ++i) col[i] = (i == m) ? 2 * gam[i / 2] + Pg[i * N + m] : Pg[i * N + m];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            Kreg[S::slot(NC + i, m / 2)] += col[i] * ((m & 1) ? aB[m / 2] : aA[m / 2]);
#pragma unroll
            for (int j = 0; j <= i; ++j) Kreg[S::slot(NC + i, NC + j)] += col[i] * col[j];
        }
    }
#pragma unroll
    for (int i = 0; i < N; ++i) Kreg[S::slot(NC + i, NC + i)] += kMuIr;
}

// The derivative system of solveDerivativesQCQP (Solver.cpp:619-681) from a problem's inputs: the contact duals gam and the
// active set cact, and -- through putK(i, j, v), i >= j, and putAb(i, v) -- K = A A^T + mu I and A^T b.  Pm is overwritten
// (its diagonal becomes that of D).  Called twice in the REGEN instantiation: the same code, the same bits.
template <typename S, int N, typename PutK, typename PutAb>
static DQQ_D __attribute__((always_inline)) void qcqp_system(double (&Pm)[N][N], const double (&xv)[N], const double (&gv)[N],
                                                             const double (&qv)[N], const double (&lnv)[N / 2],
                                                             const double (&mcv)[N / 2], double dual_eps, double (&gam)[N / 2],
                                                             bool (&cact)[N / 2], PutK putK, PutAb putAb)
{
#pragma clang fp contract(off)
    constexpr int NC = N / 2;
    // ---- (P l + q), :606
    double plq[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < N; ++j) s += Pm[i][j] * xv[j];
        plq[i] = s + qv[i];
    }
    double aS[NC], aA[NC], aB[NC];
    qcqp_contacts<N>(xv, plq, lnv, mcv, dual_eps, gam, cact, aS, aA, aB);
    // coordinate row i of A: cx at its contact's column, D = P + blkdiag(2 gamma_i I2) behind it
    double cx[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        cx[i] = cact[i / 2] ? 2 * xv[i] : 0.0;
        Pm[i][i] = 2 * gam[i / 2] + Pm[i][i];
    }
    // ---- A^T b (:19), b = [0; grad_l] (:659-667), and K = A A^T + mu I (:20-21): sums over k = contact columns, then
    // coordinate columns, structural zeros left out
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        double s = 0.0;
        s += aA[c] * gv[2 * c];
        s += aB[c] * gv[2 * c + 1];
        putAb(c, s);
        double t = 0.0;
        t += aS[c] * aS[c];
        t += aA[c] * aA[c];
        t += aB[c] * aB[c];
        t += kMuIr;
        putK(c, c, t);
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < N; ++j) s += Pm[i][j] * gv[j];
        putAb(NC + i, s);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            double t = 0.0;
            if (i / 2 == c) t += cx[i] * aS[c];
            t += Pm[i][2 * c] * aA[c];
            t += Pm[i][2 * c + 1] * aB[c];
            putK(NC + i, c, t);
        }
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            double t = 0.0;
            if (i / 2 == j / 2) t += cx[i] * cx[j];
#pragma unroll
            for (int m = 0; m < N; ++m) t += Pm[i][m] * Pm[j][m];
            if (j == i) t += kMuIr;
            putK(NC + i, NC + j, t);
        }
    }
}

// REPORT mode (below): one call per wave.  A wave's count and its arrival travel in ONE 64-bit atomic (arrivals in the high
// word, problems in the low one), in two levels like the exit tickets of worklist_release -- wave i on the word of cache line
// i mod 32, the last arrival of each group on the top word -- so that no address sees more than 32 atomics and nothing needs a
// fence (with __threadfence() between a count and a ticket, an L2 write-back per wave, the report cost 16 us of a 55 us launch).
// The words are the idle work-list header's (no list is in use by this launch) and are left zero.
static DQQ_D void report_nondiagonal(int* ws, int cnt, unsigned long long* fb, long B)
{
    const int g = (int)(blockIdx.x & 31u);
    const unsigned long long members = (unsigned long long)(((int)gridDim.x - g + 31) >> 5);
    unsigned long long* sub = reinterpret_cast<unsigned long long*>(ws + kWsSubTickets + g * kWsSubStride);
    const unsigned long long one = 1ull << 32;
    const unsigned long long before = atomicAdd(sub, one | (unsigned long long)cnt);
    if ((before >> 32) == members - 1) {
        *sub = 0;
        const unsigned long long gc = (before & 0xffffffffull) + (unsigned long long)cnt;
        unsigned long long* top = reinterpret_cast<unsigned long long*>(ws + kWsRepTop);
        const unsigned long long groups = gridDim.x < 32u ? gridDim.x : 32u;
        const unsigned long long tb = atomicAdd(top, one | gc);
        if ((tb >> 32) == groups - 1) {
            *top = 0;
            worklist_feedback(fb, ws, B, (long)((tb & 0xffffffffull) + gc));
        }
    }
}

// LIST: the drain launch behind the diagonal fast path's backward (DQQ_P_AUTO) -- the problems are the entries of the
// work-list `ws` (launch.h), 64 consecutive entries per wave; `B` is then the batch the list was drawn from.  The launch is
// sized for B entries: a wave beyond the list leaves on one scalar load, the first one reports the list's length to the
// host's feedback word (launch.h worklist_feedback) and the last one out re-zeroes the list's header.
//
// REPORT: a DQQ_P_AUTO batch that the feedback word says was ALL queued last time -- the diagonal fast path's launch is skipped
// altogether and every problem solved here (a diagonal problem gets the same bits from this routine as from the fast path:
// both follow the reference's order; tools/probe_diag_in_general_bits.py, tests).  Nothing classifies the batch then, so this
// launch does, for the next call: each wave counts the problems the fast path WOULD have queued (whole tiles of its
// 128 / N problems, as bwd_diag.hip pushes them) and the last wave to have counted sends the total to the feedback word.
template <int KIND, int N, int MODE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, (KIND == 1 && N > 4) ? 1 : 2))) void bwd_lane_dense_kernel(
    const double* __restrict__ P, const double* __restrict__ q, const double* __restrict__ aux0,
    const double* __restrict__ aux1, const double* __restrict__ x, const double* __restrict__ grad_x,
    double* __restrict__ grad_P, double* __restrict__ grad_q, double* __restrict__ gout0, double* __restrict__ gout1,
    double* __restrict__ gamma_out, double* __restrict__ dgamma_out, long B, double dual_eps, int* __restrict__ ir_steps,
    int* __restrict__ ws, unsigned long long* __restrict__ feedback)
{
#pragma clang fp contract(off)
    using S = LaneSys<KIND, N>;
    constexpr int M = S::M, NC = S::NC;
    constexpr bool LIST = MODE == 1, REPORT = MODE == 2;
    if constexpr (LIST) {   // an empty list: every wave leaves on one scalar load, before the hygiene checks of launch.h
        if (ws[kWsCount] == 0) {
            if (blockIdx.x == 0 && threadIdx.x == 0) worklist_feedback(feedback, ws, B, 0);
            return;
        }
        asm volatile("" ::: "memory");
    }
    const long total = LIST ? worklist_checked_count(ws, ws + kWsCount, kWsEntryInts(B)) : B;   // problems of this launch
    if constexpr (LIST) {
        if (blockIdx.x == 0 && threadIdx.x == 0) worklist_feedback(feedback, ws, B, total);
        if ((long)blockIdx.x * 64 >= total) return;      // (an empty list: every wave, before anything else)
        asm volatile("" ::: "memory");
    }
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* kl = smem + threadIdx.x;
    const long slot = (long)blockIdx.x * 64 + threadIdx.x;
    const bool valid = slot < total;
    // lanes past the end redo the last problem and store nothing
    const long prob = LIST ? worklist_checked_entry(ws, ws[kWsEntries + (valid ? slot : total - 1)], B) : (valid ? slot : total - 1);
    [[maybe_unused]] const int prob32 = (int)prob;

    // ---- load.  A lane's matrix is N*N contiguous doubles, N*N*8 bytes from its neighbour's: read lane by lane, every
    // load instruction touches 64 cache lines for 1 KiB of data, and with the whole chip streaming that way the tiles do
    // not stay in L1 / L2 until their remaining chunks are asked for.  The wave's tile of 64 matrices is contiguous:
    // stream it with coalesced 16-byte loads through LDS (row stride N*N + 1 doubles: conflict-free reads), before K moves in.
    constexpr int NN = N * N, TS = NN + 1, CPP = NN / 2;   // doubles per matrix, LDS row stride, 16-byte chunks per matrix
    const long first = (long)blockIdx.x * 64;
    const int nvalid = (total - first) < 64 ? (int)(total - first) : 64;
    double Pm[N][N], xv[N], gv[N], qv[N];
    {
        const double* Pw = P + first * (long)NN;
        const int last = nvalid * CPP - 1;                     // (a ragged last tile re-reads its last chunk: no branches)
#pragma unroll
        for (int k = 0; k < CPP; ++k) {
            int ch = k * 64 + (int)threadIdx.x;                // chunk of the tile
            ch = ch < last ? ch : last;
            const int pp = ch / CPP, w = ch % CPP;
            const double* src = Pw + 2 * (long)ch;
            if constexpr (LIST) src = P + (long)__shfl(prob32, pp) * NN + 2 * w;   // (runs of a tile's problems stay coalesced)
            const double2 t = *reinterpret_cast<const double2*>(src);
            smem[pp * TS + 2 * w] = t.x;
            smem[pp * TS + 2 * w + 1] = t.y;
        }
        __syncthreads();   // (one wave per workgroup: an LDS fence)
        const double* Pl = smem + (valid ? (int)threadIdx.x : nvalid - 1) * TS;
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = 0; j < N; ++j) Pm[i][j] = Pl[i * N + j];
        __syncthreads();   // the tile is in registers: K may overwrite it
        if constexpr (REPORT) {
            bool nd = false;
#pragma unroll
            for (int i = 0; i < N; ++i)
#pragma unroll
                for (int j = 0; j < N; ++j)
                    if (i != j) nd = nd || (Pm[i][j] != 0.0);
            const unsigned long long mask = __ballot(valid && nd);
            if (threadIdx.x == 0) {
                constexpr int T = 128 / N;   // problems per wave tile of bwd_diag_kernel
                int cnt = 0;
#pragma unroll
                for (int g0 = 0; g0 < 64; g0 += T) {
                    const unsigned long long grp = (T == 64) ? mask : ((mask >> g0) & ((1ull << (T % 64)) - 1));
                    const int members = nvalid - g0 < T ? nvalid - g0 : T;
                    if (grp != 0 && members > 0) cnt += members;
                }
                report_nondiagonal(ws, cnt, feedback, B);
            }
        }
#pragma unroll
        for (int i = 0; i < N; i += 2) {
            const double2 a = *reinterpret_cast<const double2*>(x + prob * N + i);
            const double2 b = *reinterpret_cast<const double2*>(grad_x + prob * N + i);
            const double2 c = *reinterpret_cast<const double2*>(q + prob * N + i);
            xv[i] = a.x; xv[i + 1] = a.y;
            gv[i] = b.x; gv[i + 1] = b.y;
            qv[i] = c.x; qv[i + 1] = c.y;
        }
    }
    double abr[S::AB_REG > 0 ? S::AB_REG : 1], xs[M];
    int steps = 0;

    if constexpr (KIND == 0) {
        // ---- dualFromPrimalQP (:125-134), the active set (:139-147), A = [[diag(l_A), 0],[0, P_II]] in the original
        // coordinate order (small_bwd_core.h), its transpose handed to iterative_refinement (:174)
        bool act[N];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            double gam = 0.0;
#pragma unroll
            for (int j = 0; j < N; ++j) gam += Pm[i][j] * xv[j];
            gam = -(gam + qv[i]);
            if (xv[i] > dual_eps) gam = 0;
            act[i] = gam < -kActiveEps;
        }
        // row i of A; b = [0 on active; grad_l on inactive] (:175-184)
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int k = 0; k < N; ++k) Pm[i][k] = act[i] ? ((k == i) ? xv[i] : 0.0) : (act[k] ? 0.0 : Pm[i][k]);
        double bv[N];
#pragma unroll
        for (int i = 0; i < N; ++i) bv[i] = act[i] ? 0.0 : gv[i];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            double s = 0.0;                                                     // A^T b, :19
#pragma unroll
            for (int k = 0; k < N; ++k) s += Pm[i][k] * bv[k];
            set_ab<S>(kl, abr, i, s);
#pragma unroll
            for (int j = 0; j <= i; ++j) {                                      // A^T A + mu I, :20-21
                double t = 0.0;
#pragma unroll
                for (int k = 0; k < N; ++k) t += Pm[i][k] * Pm[j][k];
                if (j == i) t += kMuIr;
                DQQ_KL(S::slot(i, j)) = t;
            }
        }
        lane_ir<S>(kl, abr, xs, steps);
        double dl[N];
#pragma unroll
        for (int i = 0; i < N; ++i) dl[i] = act[i] ? 0.0 : xs[i];                // :187-191
        if (valid) {
#pragma unroll
            for (int i = 0; i < N; ++i)
                if (grad_q != nullptr) grad_q[prob * N + i] = -dl[i];
            if (ir_steps != nullptr) ir_steps[prob] = steps;
        }
        if (grad_P != nullptr) store_grad_P_tile<N, LIST>(smem, grad_P, first, nvalid, dl, xv, prob32);
    } else {
        double lnv[NC], mcv[NC], gam[NC];
        bool cact[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) { lnv[c] = aux0[prob * NC + c]; mcv[c] = aux1[prob * NC + c]; }
        qcqp_system<S, N>(Pm, xv, gv, qv, lnv, mcv, dual_eps, gam, cact,
                          [&](int i, int j, double v) __attribute__((always_inline)) { DQQ_KL(S::slot(i, j)) = v; },
                          [&](int i, double v) __attribute__((always_inline)) { set_ab<S>(kl, abr, i, v); });
        if constexpr (S::REGEN) {
            // everything but A^T b is rebuilt after the inverse: nothing of the build stays live across the factorisation
            lane_ir<S>(kl, abr, xs, steps, [&](double (&Kreg)[S::SLOTS]) __attribute__((always_inline)) {
                const double* Pg = P + prob * (long)(N * N);
                double x2[N], q2[N], ln2[NC], mc2[NC];
#pragma unroll
                for (int i = 0; i < N; i += 2) {
                    const double2 a = *reinterpret_cast<const double2*>(x + prob * N + i);
                    const double2 c = *reinterpret_cast<const double2*>(q + prob * N + i);
                    x2[i] = a.x; x2[i + 1] = a.y;
                    q2[i] = c.x; q2[i + 1] = c.y;
                }
#pragma unroll
                for (int c = 0; c < NC; ++c) { ln2[c] = aux0[prob * NC + c]; mc2[c] = aux1[prob * NC + c]; }
                qcqp_rebuild_K<S, N>(Pg, x2, q2, ln2, mc2, dual_eps, gam, cact, Kreg);
            });
            // (x, l_n, mu come back from memory for the outputs: nothing but gam / cact lives through the refinement loop)
#pragma unroll
            for (int i = 0; i < N; i += 2) {
                const double2 a = *reinterpret_cast<const double2*>(x + prob * N + i);
                xv[i] = a.x; xv[i + 1] = a.y;
            }
#pragma unroll
            for (int c = 0; c < NC; ++c) { lnv[c] = aux0[prob * NC + c]; mcv[c] = aux1[prob * NC + c]; }
        } else {
            lane_ir<S>(kl, abr, xs, steps);
        }
        if (valid) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const double ln = lnv[c], mc = mcv[c];
                const double dg = cact[c] ? xs[c] : 0.0;                        // :671-674
                if (gout0 != nullptr) gout0[prob * NC + c] = QcqpContact::e2(gam[c], ln, mc) * dg;   // grad_l_n
                if (gout1 != nullptr) gout1[prob * NC + c] = QcqpContact::e1(gam[c], ln, mc) * dg;   // grad_mu
                if (gamma_out != nullptr) gamma_out[prob * NC + c] = gam[c];
                if (dgamma_out != nullptr) dgamma_out[prob * NC + c] = dg;
            }
#pragma unroll
            for (int i = 0; i < N; ++i)
                if (grad_q != nullptr) grad_q[prob * N + i] = -xs[NC + i];
            if (ir_steps != nullptr) ir_steps[prob] = steps;
        }
        if (grad_P != nullptr) {
            double dl[N];
#pragma unroll
            for (int i = 0; i < N; ++i) dl[i] = xs[NC + i];
            store_grad_P_tile<N, LIST>(smem, grad_P, first, nvalid, dl, xv, prob32);
        }
    }
    if constexpr (LIST) {
        if (threadIdx.x == 0) worklist_release(ws, total, (int)((total + 63) / 64));   // the waves that did not leave at the top
    }
}

#undef DQQ_KL

template <int KIND, int N, int MODE>
static hipError_t launch_lane_bwd(const BwdArgs& a, hipStream_t s)
{
    using S = LaneSys<KIND, N>;
    // the lane-interleaved K area (+ A^T b), or the staged tile of P / grad_P (stride N*N + 1), whichever is larger
    const size_t lds = sizeof(double) * 64 * (size_t)(S::LDS_SLOTS > N * N + 1 ? S::LDS_SLOTS : N * N + 1);
    const long grid = (a.B + 63) / 64;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bwd_lane_dense_kernel<KIND, N, MODE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    return launch(bwd_lane_dense_kernel<KIND, N, MODE>, dim3((unsigned)grid), dim3(64), lds, s, a.P, a.q, a.l_n, a.mu, a.x,
                  a.grad_x, a.grad_P, a.grad_q, a.grad_l_n, a.grad_mu, a.gamma, a.dgamma, a.B, a.epsilon, a.ir_steps, a.ws,
                  (MODE != 0 && hint_applies(KIND, N)) ? a.report : nullptr);
}

// P declared dense, QP / QCQP, N = 2, 4, 6, 8, batches that fill the chip: a lane per problem needs 64 problems per wave
// and a wave per SIMD, below that the team kernel's 4 to 8 problems per wave spread a batch better.  Backward, us, team
// / lane (tools/probe_lane_bwd.py --sweep):   QCQP N = 8: B = 16384 34 / 35, 24576 49 / 34, 32768 64 / 42, 65536 117 / 86,
// 131072 221 / 140;  QCQP N = 6: 16384 25 / 20, 65536 79 / 26, 131072 146 / 48;  QP N = 8: 16384 20 / 19, 32768 24 / 14,
// 65536 37 / 22, 131072 68 / 42;  QP N = 6: 65536 21 / 12;  N <= 4: launch-bound up to 65536, 131072: 46 / 25 (QCQP N = 4)
bool bwd_lane_dense_supported(int kind, int N, long B)
{
    const long min_b = (N == 8) ? 24576 : 16384;
    return (kind == kKindQP || kind == kKindQCQP) && (N == 2 || N == 4 || N == 6 || N == 8) && B >= min_b;
}

// mode 0: the whole batch, declared dense; 1: the entries of the work-list; 2: the whole batch of a DQQ_P_AUTO call, reporting
hipError_t launch_bwd_lane_dense(int kind, const BwdArgs& a, int mode, hipStream_t s)
{
    if (a.B == 0) return hipSuccess;
#define DQQ_CASE(NN)                                                                                               \
    if (a.N == NN) {                                                                                               \
        if (mode == 1) return kind == 0 ? launch_lane_bwd<0, NN, 1>(a, s) : launch_lane_bwd<1, NN, 1>(a, s);       \
        if (mode == 2) return kind == 0 ? launch_lane_bwd<0, NN, 2>(a, s) : launch_lane_bwd<1, NN, 2>(a, s);       \
        return kind == 0 ? launch_lane_bwd<0, NN, 0>(a, s) : launch_lane_bwd<1, NN, 0>(a, s);                      \
    }
    DQQ_CASE(2) DQQ_CASE(4) DQQ_CASE(6) DQQ_CASE(8)
#undef DQQ_CASE
    return hipErrorInvalidValue;
}

} // namespace dqq
