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
// Storage per lane (QCQP, N = 8: M = 12 unknowns): the factor L (72 doubles) and the explicit inverse K^-1 (144: its two
// triangles are computed by different operation sequences and are not bitwise symmetric, so both are kept) in registers
// -- one wave per SIMD, 512 registers per lane --, K itself (72, read again by every refinement body) in LDS, lane-
// interleaved (element e of lane l at (e * 64 + l) * 8: conflict-free, 36.9 KB per wave = 4 waves per CU).  K^-1 A^T b is
// accumulated while the columns of the inverse are produced, and the first refinement body (x = 0) multiplies nothing.
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
};

// K in LDS: element e of this lane
#define DQQ_KL(e) kl[(e) * 64]

// Solver::iterative_refinement (Solver.cpp:15-44) on K = A_t^T A_t + mu I (lower triangle in the lane's LDS column) and
// Ab = A_t^T b, in the operation order of small_bwd_core.h: team_ir.  Returns xs, the number of bodies in `steps`.
template <typename S>
static DQQ_D void lane_ir(const double* __restrict__ kl, const double (&Ab)[S::M], double (&xs)[S::M], int& steps)
{
#pragma clang fp contract(off)
    constexpr int M = S::M;
    // ---- llt(), :23 -- left-looking, column by column
    double L[M][M];
#pragma unroll
    for (int k = 0; k < M; ++k) {
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < k; ++j)
            if (!S::kz(k, j)) s += L[k][j] * L[k][j];
        double xk = DQQ_KL(S::slot(k, k)) - s;
        xk = sqrt(xk);
        L[k][k] = xk;
#pragma unroll
        for (int i = k + 1; i < M; ++i) {
            if (S::kz(i, k)) continue;   // (0 - 0) / xk
            double t = 0.0;
#pragma unroll
            for (int j = 0; j < k; ++j)
                if (!S::kz(i, j) && !S::kz(k, j)) t += L[i][j] * L[k][j];
            L[i][k] = (DQQ_KL(S::slot(i, k)) - t) / xk;
        }
    }
    // ---- solveInPlace(Identity), :22-23, column by column; K^-1 A^T b (:27) accumulated as the columns arrive
    double Kinv[M][M], KinvAb[M];
#pragma unroll
    for (int i = 0; i < M; ++i) KinvAb[i] = 0.0;
#pragma unroll
    for (int c = 0; c < M; ++c) {
        double y[M];
#pragma unroll
        for (int i = 0; i < M; ++i) {
            if (i < c) { y[i] = 0.0; continue; }
            double t = (i == c) ? 1.0 : 0.0;
#pragma unroll
            for (int j = c; j < i; ++j)
                if (!S::kz(i, j)) t -= L[i][j] * y[j];
            y[i] = t / L[i][i];
        }
#pragma unroll
        for (int i = M - 1; i >= 0; --i) {
            double t = y[i];
#pragma unroll
            for (int j = i + 1; j < M; ++j)
                if (!S::kz(j, i)) t -= L[j][i] * y[j];
            y[i] = t / L[i][i];
        }
#pragma unroll
        for (int i = 0; i < M; ++i) {
            Kinv[i][c] = y[i];
            KinvAb[i] += y[i] * Ab[c];
        }
    }
    // ---- the refinement loop, :26-41.  Lanes leave it one by one (the wave runs the longest).
#pragma unroll
    for (int i = 0; i < M; ++i) xs[i] = 0.0;
    IrControl ctl;
    ctl.init();
    steps = 0;
    bool done = false;
    for (int it = 0; it < kIrMaxIter; ++it) {
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
                    for (int j = 0; j < M; ++j) tmp += Kinv[i][j] * xs[j];
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
                    if (!S::kz(a, b)) d += DQQ_KL(S::slot(a, b)) * xs[j];
                }
                d = d - Ab[i];
                ss += d * d;                                                    // :31
            }
            if (ctl.update(sqrt(ss))) done = true;                              // :32-41
        }
        if (__all(done)) break;
    }
}

} // namespace

template <int KIND, int N>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, (KIND == 1 && N > 4) ? 1 : 2))) void bwd_lane_dense_kernel(
    const double* __restrict__ P, const double* __restrict__ q, const double* __restrict__ aux0,
    const double* __restrict__ aux1, const double* __restrict__ x, const double* __restrict__ grad_x,
    double* __restrict__ grad_P, double* __restrict__ grad_q, double* __restrict__ gout0, double* __restrict__ gout1,
    double* __restrict__ gamma_out, double* __restrict__ dgamma_out, long B, double dual_eps, int* __restrict__ ir_steps)
{
#pragma clang fp contract(off)
    using S = LaneSys<KIND, N>;
    constexpr int M = S::M, NC = S::NC;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* kl = smem + threadIdx.x;
    const long slot = (long)blockIdx.x * 64 + threadIdx.x;
    const bool valid = slot < B;
    const long prob = valid ? slot : B - 1;   // lanes past the end redo the last problem and store nothing

    double Pm[N][N], xv[N], gv[N], qv[N];
    {
        const double* Pg = P + prob * (long)(N * N);
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = 0; j < N; j += 2) {
                const double2 t = *reinterpret_cast<const double2*>(Pg + i * N + j);
                Pm[i][j] = t.x;
                Pm[i][j + 1] = t.y;
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
    double Ab[M], xs[M];
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
            Ab[i] = s;
#pragma unroll
            for (int j = 0; j <= i; ++j) {                                      // A^T A + mu I, :20-21
                double t = 0.0;
#pragma unroll
                for (int k = 0; k < N; ++k) t += Pm[i][k] * Pm[j][k];
                if (j == i) t += kMuIr;
                DQQ_KL(S::slot(i, j)) = t;
            }
        }
        lane_ir<S>(kl, Ab, xs, steps);
        if (valid) {
            double* Gp = grad_P != nullptr ? grad_P + prob * (long)(N * N) : nullptr;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const double dl = act[i] ? 0.0 : xs[i];                         // :187-191
                if (grad_q != nullptr) grad_q[prob * N + i] = -dl;
                if (Gp != nullptr) {
#pragma unroll
                    for (int j = 0; j < N; j += 2)
                        *reinterpret_cast<double2*>(Gp + i * N + j) = make_double2(-(dl * xv[j]), -(dl * xv[j + 1]));
                }
            }
            if (ir_steps != nullptr) ir_steps[prob] = steps;
        }
    } else {
        // ---- (P l + q), :606
        double plq[N];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            double s = 0.0;
#pragma unroll
            for (int j = 0; j < N; ++j) s += Pm[i][j] * xv[j];
            plq[i] = s + qv[i];
        }
        // ---- dualFromPrimalQCQP (:584-617) and the active set of solveDerivativesQCQP (:622-641), contact by contact;
        // row c of A = [[diag(S), diag(gamma) C^T],[C, P + blkdiag(2 gamma_i I2)]] (:643-657): aS at column c, aA / aB at
        // the contact's two coordinate columns; an inactive contact's row is zero
        double gam[NC], aS[NC], aA[NC], aB[NC];
        bool cact[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const double ln = aux0[prob * NC + c], mc = aux1[prob * NC + c];
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
            Ab[c] = s;
            double t = 0.0;
            t += aS[c] * aS[c];
            t += aA[c] * aA[c];
            t += aB[c] * aB[c];
            t += kMuIr;
            DQQ_KL(S::slot(c, c)) = t;
        }
#pragma unroll
        for (int i = 0; i < N; ++i) {
            double s = 0.0;
#pragma unroll
            for (int j = 0; j < N; ++j) s += Pm[i][j] * gv[j];
            Ab[NC + i] = s;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                double t = 0.0;
                if (i / 2 == c) t += cx[i] * aS[c];
                t += Pm[i][2 * c] * aA[c];
                t += Pm[i][2 * c + 1] * aB[c];
                DQQ_KL(S::slot(NC + i, c)) = t;
            }
#pragma unroll
            for (int j = 0; j <= i; ++j) {
                double t = 0.0;
                if (i / 2 == j / 2) t += cx[i] * cx[j];
#pragma unroll
                for (int m = 0; m < N; ++m) t += Pm[i][m] * Pm[j][m];
                if (j == i) t += kMuIr;
                DQQ_KL(S::slot(NC + i, NC + j)) = t;
            }
        }
        lane_ir<S>(kl, Ab, xs, steps);
        if (valid) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const double ln = aux0[prob * NC + c], mc = aux1[prob * NC + c];
                const double dg = cact[c] ? xs[c] : 0.0;                        // :671-674
                if (gout0 != nullptr) gout0[prob * NC + c] = QcqpContact::e2(gam[c], ln, mc) * dg;   // grad_l_n
                if (gout1 != nullptr) gout1[prob * NC + c] = QcqpContact::e1(gam[c], ln, mc) * dg;   // grad_mu
                if (gamma_out != nullptr) gamma_out[prob * NC + c] = gam[c];
                if (dgamma_out != nullptr) dgamma_out[prob * NC + c] = dg;
            }
            double* Gp = grad_P != nullptr ? grad_P + prob * (long)(N * N) : nullptr;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const double dl = xs[NC + i];
                if (grad_q != nullptr) grad_q[prob * N + i] = -dl;
                if (Gp != nullptr) {
#pragma unroll
                    for (int j = 0; j < N; j += 2)
                        *reinterpret_cast<double2*>(Gp + i * N + j) = make_double2(-(dl * xv[j]), -(dl * xv[j + 1]));
                }
            }
            if (ir_steps != nullptr) ir_steps[prob] = steps;
        }
    }
}

#undef DQQ_KL

template <int KIND, int N>
static hipError_t launch_lane_bwd(const BwdArgs& a, hipStream_t s)
{
    using S = LaneSys<KIND, N>;
    const size_t lds = sizeof(double) * 64 * (size_t)S::SLOTS;
    const long grid = (a.B + 63) / 64;
    return launch(bwd_lane_dense_kernel<KIND, N>, dim3((unsigned)grid), dim3(64), lds, s, a.P, a.q, a.l_n, a.mu, a.x,
                  a.grad_x, a.grad_P, a.grad_q, a.grad_l_n, a.grad_mu, a.gamma, a.dgamma, a.B, a.epsilon, a.ir_steps);
}

// P declared dense, QP / QCQP, N = 2, 4, 6, 8, batches that fill the chip (a lane per problem needs 64 problems per
// wave and a wave per SIMD: below ~16 Ki problems the team kernel's 4 problems per wave spread a small batch better)
bool bwd_lane_dense_supported(int kind, int N, long B)
{
    return (kind == kKindQP || kind == kKindQCQP) && (N == 2 || N == 4 || N == 6 || N == 8) && B >= 16384;
}

hipError_t launch_bwd_lane_dense(int kind, const BwdArgs& a, hipStream_t s)
{
    if (a.B == 0) return hipSuccess;
#define DQQ_CASE(NN)                                              \
    if (a.N == NN) return kind == 0 ? launch_lane_bwd<0, NN>(a, s) : launch_lane_bwd<1, NN>(a, s);
    DQQ_CASE(2) DQQ_CASE(4) DQQ_CASE(6) DQQ_CASE(8)
#undef DQQ_CASE
    return hipErrorInvalidValue;
}

} // namespace dqq
