// bwd_wave_qcqp_big.hip -- QCQP backward for a general (dense) P, 32 < N <= 64: ONE WAVE per problem, the factor of
// the (N + N/2)-unknown system (up to 96 unknowns = 6 x 6 tiles of 16) in registers, the system matrix A STREAMED.
//
// Same composition as bwd_wave_qcqp.hip (pybindings.cpp:62-71 -> Solver.cpp:584-617, :683-691, :619-681, :15-44;
// qcqp.py:173-180), same block-Cholesky core (wave_chol.h).  What is different at this size:
//   * slots: the coordinates come first (slots 0 .. 16 NX - 1, lane l of the "coordinate register" of a vector = entry l),
//     the contacts behind them (slot 16 NX + c = lane c < 32 of its "contact register"): a vector is two registers, a
//     product with a tile row takes its broadcast operand from one or the other.  A symmetric permutation of the
//     reference's (active contacts..., coordinates...) plus empty slots (inactive contacts: mu on the diagonal of K).
//   * A (36 tiles at N = 64) does not fit beside the 21 upper tiles of K.  It never exists as a whole: a TILE COLUMN of
//     A^T (NT tiles: P from L2 + the few contact entries, generated from x, gamma, S and the active mask) is built,
//     used and dropped -- once for P l (dual recovery), once for the Gram matrix K = A A^T (+ the right-hand side A dd
//     from the same tiles), and once per refinement body for the residual, evaluated as A (A^T x) + mu x - A dd
//     (:30; no second copy of K: the first product contracts over the lanes of a 16-lane row, its result is exactly the
//     per-row operand the second product wants).
// Round 2: LDS wave kernel up to N = 42, then the global-memory kernel in the reference's summation order: 8.6 ms per
// 4096 problems at N = 64.
#include "kkt_core.h"
#include "launch.h"
#include "wave_chol.h"

namespace dqq {

template <int NX, int NCT>
struct QcqpSystem {
    static constexpr int NT = NX + NCT;
    const double* Pg;
    int N, g, n;
    unsigned lo;            // n * N + g: per-lane offset of the P tile loads
    double xi;              // coordinate register: l_i in lane i
    double gam, S;          // contact register: gamma_c, S_c in lane c
    unsigned long long am;  // bit c: contact c is active in the derivative system (:637-641)

    // tile column TK of the tile layout of A^T: T[ta][r] of lane (g,n) = A[slot 16 ta + n][slot 16 TK + 4 r + g]
    template <int TK, bool ONLY_P = false>
    DQQ_D void column(v4d (&T)[NT]) const
    {
        const v4d zero = {0.0, 0.0, 0.0, 0.0};
        if constexpr (TK < NX) {      // coordinate columns j = 16 TK + 4 r + g
            double gdiag = 0.0;
            if constexpr (!ONLY_P) gdiag = lane_gather(gam, (16 * TK + n) >> 1); // gamma of the contact of row 16 TK + n
#pragma unroll
            for (int ta = 0; ta < NX; ++ta)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 16 * ta + n, j = 16 * TK + 4 * r + g;
                    double v = (i < N && j < N) ? (Pg + ((16 * ta) * N + 16 * TK + 4 * r))[lo] : ((i == j) ? 1.0 : 0.0);
                    if (!ONLY_P && ta == TK && 4 * r + g == n && i < N) v += 2 * gdiag;   // + 2 gamma_(i/2), :655
                    T[ta][r] = v;
                }
            if constexpr (ONLY_P) {
#pragma unroll
                for (int ta = NX; ta < NT; ++ta) T[ta] = zero;
            } else {
                double xc[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) xc[r] = lane_gather(xi, 16 * TK + 4 * r + g);
#pragma unroll
                for (int ta = NX; ta < NT; ++ta) {   // contact rows c = 16 (ta - NX) + n: gamma_c 2 l_j, j in contact c (:647-650)
                    const int c = 16 * (ta - NX) + n;
                    const double gc = lane_gather(gam, c);
                    const bool actc = (am >> c) & 1ull;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int j = 16 * TK + 4 * r + g;
                        T[ta][r] = (actc && (j >> 1) == c) ? gc * (2 * xc[r]) : 0.0;
                    }
                }
            }
        } else {                      // contact columns c = 16 (TK - NX) + 4 r + g
#pragma unroll
            for (int ta = 0; ta < NX; ++ta) {        // coordinate rows i: 2 l_i, i in contact c (:651-653)
                const int i = 16 * ta + n;
                const double xr = lane_gather(xi, i);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = 16 * (TK - NX) + 4 * r + g;
                    T[ta][r] = (((am >> c) & 1ull) && (i >> 1) == c) ? 2 * xr : 0.0;
                }
            }
#pragma unroll
            for (int ta = NX; ta < NT; ++ta) {       // contact rows: S_c on the diagonal (:644-646)
                const int cr = 16 * (ta - NX) + n;
                const double Sn = lane_gather(S, cr);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = 16 * (TK - NX) + 4 * r + g;
                    T[ta][r] = (ta == TK && c == cr && ((am >> c) & 1ull)) ? Sn : 0.0;
                }
            }
        }
    }
};

// broadcast operand of tile (row) index T for a vector held as (coordinate register, contact register), both already in
// the gathered form lane (g, n') <- entry 4 n' + g
template <int NX, int T>
DQQ_D void tile_dot4_of(double& acc, const v4d& tile, double x0c, double x0k)
{
    if constexpr (T < NX) tile_dot4<4 * T>(acc, tile, x0c);
    else tile_dot4<4 * (T - NX)>(acc, tile, x0k);
}

// (yc, yk) = S (vc, vk) for the symmetric S whose UPPER tiles are in Su
template <int NX, int NCT>
DQQ_D void sym_upper_matvec2(const v4d (&Su)[NX + NCT][NX + NCT], double vc, double vk, int xsrc, int lane, double& yc,
                             double& yk, double* __restrict__ trbuf)
{
    constexpr int NT = NX + NCT;
    const double x0c = dpp_source(lane_gather(vc, xsrc)), x0k = dpp_source(lane_gather(vk, xsrc));
    double a[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0}; // a[0..3]: coordinate blocks, a[4..7]: contact blocks
    static_for<0, NT>([&](auto tjc) __attribute__((always_inline)) {
        constexpr int TJ = decltype(tjc)::value;
        constexpr int AJ = TJ < NX ? TJ : 4 + (TJ - NX);
        static_for<0, TJ + 1>([&](auto tic) __attribute__((always_inline)) {
            constexpr int TI = decltype(tic)::value;
            constexpr int AI = TI < NX ? TI : 4 + (TI - NX);
            tile_dot4_of<NX, TI>(a[AJ], Su[TI][TJ], x0c, x0k);
            if constexpr (TI < TJ) tile_dot4_of<NX, TJ>(a[AI], tile_transpose_lds(Su[TI][TJ], trbuf, lane), x0c, x0k);
        });
    });
    yc = reduce_scatter4(a[0], a[1], a[2], a[3]);
    yk = reduce_scatter4(a[4], a[5], a[6], a[7]);
}

template <int NX, int NCT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 2))) void bwd_wave_qcqp_big_kernel(
    const double* __restrict__ P, const double* __restrict__ q, const double* __restrict__ l_n,
    const double* __restrict__ mu_c, const double* __restrict__ x, const double* __restrict__ grad_x,
    double* __restrict__ grad_P, double* __restrict__ grad_q, double* __restrict__ grad_l_n,
    double* __restrict__ grad_mu, double* __restrict__ gamma_out, double* __restrict__ dgamma_out, long B, int N,
    double dual_eps, int* __restrict__ ir_steps, int* __restrict__ ws, int use_worklist)
{
    constexpr int NT = NX + NCT;
    using Sys = QcqpSystem<NX, NCT>;
    __shared__ __attribute__((aligned(16))) double s_trb[16 * kTrLd]; // tile transposes (one wave per workgroup)
    WorkClaim claim; // (launch.h: direct mode, or dynamic pick-up from the work-list)
    claim.open(ws, use_worklist, N, B);
    const int nc = N / 2;
    for (long w = blockIdx.x;; w += gridDim.x) {
        const long prob = claim.next(ws, B, w); // wave-uniform (SGPRs: P's addressing uses a scalar base)
        if (prob < 0) break;
        claim.ahead_issue(ws); // (work-list: the ticket for this wave's next problem travels while this one is solved)
        int lane = threadIdx.x;
        asm volatile("" : "+v"(lane));
        const int g = lane >> 4, n = lane & 15;
        const int xsrc = 4 * n + g;
        const bool is_coord = lane < N, is_contact = lane < nc;
        const double xi = is_coord ? x[prob * N + lane] : 0.0, gi = is_coord ? grad_x[prob * N + lane] : 0.0;
        const double qi = is_coord ? q[prob * N + lane] : 0.0;
        const double ln = is_contact ? l_n[prob * nc + lane] : 1.0, mc = is_contact ? mu_c[prob * nc + lane] : 1.0;
        Sys sys;
        sys.Pg = P + prob * (long)(N * N);
        sys.N = N; sys.g = g; sys.n = n; sys.lo = n * N + g;
        sys.xi = xi; sys.gam = 0.0; sys.S = 0.0; sys.am = 0ull;

        // ---- dualFromPrimalQCQP, Solver.cpp:584-617: P l + q per coordinate (first pass over P), one contact per lane
        double plq;
        {
            const double x0 = dpp_source(lane_gather(xi, xsrc));
            double acc[4] = {0.0, 0.0, 0.0, 0.0};
            static_for<0, NX>([&](auto tkc) __attribute__((always_inline)) {
                constexpr int TK = decltype(tkc)::value;
                __builtin_amdgcn_sched_barrier(0); // one tile column at a time: NT tiles live, not NT^2
                v4d T[NT];
                sys.template column<TK, true>(T);
#pragma unroll
                for (int ta = 0; ta < NX; ++ta) tile_dot4<4 * TK>(acc[ta], T[ta], x0);
            });
            plq = reduce_scatter4(acc[0], acc[1], acc[2], acc[3]) + qi;
        }
        const int c2 = 2 * (lane & 31);
        const double xa = lane_gather(xi, c2), xb = lane_gather(xi, c2 + 1);
        const double pa = lane_gather(plq, c2), pb = lane_gather(plq, c2 + 1);
        const double rr = ln * mc;                                   // pybindings.cpp:65
        double gamma = 0.0;
        {
            const double slack = rr - sqrt(xa * xa + xb * xb);
            if (is_contact && !(slack > dual_eps || rr < dual_eps)) {
                const double ca = 2 * xa, cb = 2 * xb;
                const double G2 = ca * ca + cb * cb;
                const double rhs = ca * pa + cb * pb;
                const double L = sqrt(G2);
                gamma = -((rhs / L) / L);                            // :605-615
            }
        }
        const double S = (xa * xa + xb * xb) - rr * rr;              // :622-629
        const bool is_act = is_contact && S > -kActiveEps && rr > kActiveEps; // :637-641
        sys.gam = gamma; sys.S = S; sys.am = __ballot(is_act);

        // ---- K = A A^T + mu_ir I (upper tiles) and the right-hand side A dd, dd = [grad_l; 0], one pass over A (:19-21)
        WaveChol<NT> C;
        double Abc, Abk;
        {
            const v4d zero = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ta = 0; ta < NT; ++ta)
#pragma unroll
                for (int tb = ta; tb < NT; ++tb) C.U[ta][tb] = zero;
            const double d0 = dpp_source(lane_gather(gi, xsrc));
            double acc[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            static_for<0, NT>([&](auto tkc) __attribute__((always_inline)) {
                constexpr int TK = decltype(tkc)::value;
                __builtin_amdgcn_sched_barrier(0);
                v4d T[NT];
                sys.template column<TK>(T);
#pragma unroll
                for (int ta = 0; ta < NT; ++ta)
#pragma unroll
                    for (int tb = ta; tb < NT; ++tb) C.U[ta][tb] = tile_xty(C.U[ta][tb], T[ta], T[tb]);
                if constexpr (TK < NX) {   // dd is zero on the contact slots
#pragma unroll
                    for (int ta = 0; ta < NT; ++ta) tile_dot4<4 * TK>(acc[ta < NX ? ta : 4 + (ta - NX)], T[ta], d0);
                }
            });
            Abc = reduce_scatter4(acc[0], acc[1], acc[2], acc[3]);
            Abk = reduce_scatter4(acc[4], acc[5], acc[6], acc[7]);
            const bool on_diag = (n & 3) == g;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) C.U[t][t][r] += (on_diag && (n >> 2) == r) ? kMuIr : 0.0;
        }
        bool bad = false;
        claim.ahead_entry(ws, B);
        C.factor(lane, bad);                                                  // :22
        C.invert_in_place(lane, s_trb);                                              // :23
        double Kc, Kk;                                                        // Kinv * Ab, :27
        sym_upper_matvec2<NX, NCT>(C.U, Abc, Abk, xsrc, lane, Kc, Kk, s_trb);
        double xc = 0.0, xk = 0.0;
        IrControl ctl;
        ctl.init();
        int steps = 0;
        for (int it = 0; it < kIrMaxIter; ++it) {
            steps = it + 1;
            if (it == 0) {
                xc = Kc; xk = Kk;                                             // :29 (the first body multiplies x = 0)
            } else {
                double yc, yk;
                sym_upper_matvec2<NX, NCT>(C.U, xc, xk, xsrc, lane, yc, yk, s_trb);
                xc = Kc + kMuIr * yc; xk = Kk + kMuIr * yk;
            }
            // residual (:30): K x - A dd = A (A^T x) + mu x - A dd, A streamed once more.  (The tiles do not depend on
            // `it`: left alone, the compiler hoists the generation of all NT^2 of them out of this loop -- 288 registers.)
            asm volatile("" : "+v"(sys.lo), "+v"(sys.n), "+v"(sys.g));
            double vrep[NT];
#pragma unroll
            for (int ta = 0; ta < NT; ++ta) vrep[ta] = ta < NX ? lane_gather(xc, 16 * ta + n) : lane_gather(xk, 16 * (ta - NX) + n);
            double acc[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            static_for<0, NT>([&](auto tkc) __attribute__((always_inline)) {
                constexpr int TK = decltype(tkc)::value;
                __builtin_amdgcn_sched_barrier(0);
                v4d T[NT];
                sys.template column<TK>(T);
                double s[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {                                 // (A^T x)[16 TK + 4 r + g]
                    double t = 0.0;
#pragma unroll
                    for (int ta = 0; ta < NT; ++ta) t += T[ta][r] * vrep[ta];
                    s[r] = LaneGroup<16>::sum(t);
                }
#pragma unroll
                for (int ta = 0; ta < NT; ++ta) {
                    double t = acc[ta < NX ? ta : 4 + (ta - NX)];
#pragma unroll
                    for (int r = 0; r < 4; ++r) t += T[ta][r] * s[r];
                    acc[ta < NX ? ta : 4 + (ta - NX)] = t;
                }
            });
            const double dc = reduce_scatter4(acc[0], acc[1], acc[2], acc[3]) + kMuIr * xc - Abc;
            const double dk = reduce_scatter4(acc[4], acc[5], acc[6], acc[7]) + kMuIr * xk - Abk;
            const double res = sqrt(wave_sum64(dc * dc + dk * dk));           // :31
            if (ctl.update(res)) break;                                       // :32-41
        }
        const double bc = bad ? NAN : xc, bk = bad ? NAN : xk;                // blgamma, :670-679
        if (is_contact) {
            const double dg = is_act ? bk : 0.0;
            if (grad_l_n != nullptr) grad_l_n[prob * nc + lane] = QcqpContact::e2(gamma, ln, mc) * dg;   // qcqp.py:178
            if (grad_mu != nullptr) grad_mu[prob * nc + lane] = QcqpContact::e1(gamma, ln, mc) * dg;     // qcqp.py:180
            if (gamma_out != nullptr) gamma_out[prob * nc + lane] = gamma;
            if (dgamma_out != nullptr) dgamma_out[prob * nc + lane] = dg;
        }
        claim.ahead_done(ws, B);
        if (is_coord && grad_q != nullptr) grad_q[prob * N + lane] = -bc;    // qcqp.py:176
        if (grad_P != nullptr) {                                              // qcqp.py:174: -(dl l^T)
            double* Gp = grad_P + prob * (long)(N * N);
            for (int k = 0; k < N; ++k) {
                const double v = -(lane_bcast(bc, k) * xi);
                if (is_coord) __builtin_nontemporal_store(v, Gp + k * N + lane);
            }
        }
        if (ir_steps != nullptr && lane == 0) ir_steps[prob] = steps;
    }
    if (use_worklist && threadIdx.x == 0) worklist_release(ws, claim.count, (int)gridDim.x);
}

bool bwd_wave_qcqp_big_supported(int kind, int N) { return kind == kKindQCQP && N > 32 && N <= 64 && (N & 1) == 0; }

template <int NX, int NCT>
static hipError_t launch_big(const BwdArgs& a, bool use_worklist, hipStream_t s)
{
    const long cap = 1L << 22;
    const unsigned grid = (unsigned)(a.B < (use_worklist ? 2048L : cap) ? (a.B > 0 ? a.B : 1) : (use_worklist ? 2048L : cap));
    return launch(bwd_wave_qcqp_big_kernel<NX, NCT>, dim3(grid), dim3(64), 0, s, a.P, a.q, a.l_n, a.mu, a.x, a.grad_x, a.grad_P,
                  a.grad_q, a.grad_l_n, a.grad_mu, a.gamma, a.dgamma, a.B, a.N, a.epsilon, a.ir_steps, a.ws,
                  use_worklist ? 1 : 0);
}

hipError_t launch_bwd_wave_qcqp_big(const BwdArgs& a, bool use_worklist, hipStream_t s)
{
    if (a.B == 0) return hipSuccess;
    if (!bwd_wave_qcqp_big_supported(kKindQCQP, a.N)) return hipErrorInvalidValue;
    // coordinates: ceil(N / 16) tiles; contacts: ceil(N / 32) tiles (17 .. 32 of them)
    if (a.N <= 48) return launch_big<3, 2>(a, use_worklist, s);
    return launch_big<4, 2>(a, use_worklist, s);
}

} // namespace dqq
