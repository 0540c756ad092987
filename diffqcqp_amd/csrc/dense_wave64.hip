// dense_wave64.hip -- general (dense P) forward solve for N = 64: ONE WAVE per problem, the whole problem in
// registers (BASELINE configs[4]: B=65536, N=64).
//
// Same algorithm as the reference (Solver::solveQP / solveQCQP / solveBoxQP / solveSignedBoxQP,
// Solver.cpp:61-123, 521-582, 198-261, 374-439): power iteration, adaptive-rho ADMM with the explicit inverse of
// P + (rho+mu) I rebuilt at every rho update (the reference's llt() + solveInPlace(Identity), :76-77, :100-101,
// :114-115).  What is different from the workgroup-per-problem kernel (dense_block.hip) is the execution model:
//
//   * no LDS, no workgroup barrier: the 64 x 64 matrix lives in 128 VGPRs per lane in the matrix-core tile
//     layout (wave_tile.h), two independent waves per SIMD, eight problems in flight per CU;
//   * refactorisation = block Gauss-Jordan sweep on the f64 matrix cores, register to register (240 MFMAs + four
//     16 x 16 pivot blocks inverted by DPP-broadcast sweeps), result -M^-1 in the SAME layout;
//   * ADMM iteration = 64 v_fmac_f64 with a DPP row_newbcast operand + a 3-step cross-row reduce-scatter; the
//     element-wise update, the residual maxima and the stop / rho logic are wave-uniform;
//   * waves iterate independently: a problem costs its own iteration count, not its workgroup's barrier count.
//
// The inverse is computed by a different elimination than the reference's Cholesky (sums associate
// differently: ~1e-16 * cond(M), as for every kernel of the general path); the trajectory -- rho schedule and
// iteration count -- is required to match the oracle's in tests/.  Like the reference's LLT (Solver.cpp:76) only
// the lower triangle of P enters the factorisation, while the power iteration multiplies by the full P (:51).
#include "kkt_core.h"
#include "launch.h"
#include "wave_tile.h"

namespace dqq {

// G <- tile layout of A^T for the row-major 64 x 64 matrix A: G[ti][tj][r] of lane (g,n) = A[16tj+n][16ti+4r+g],
// so that WaveTile64::matvec returns A x (no symmetry assumed).
static DQQ_D void load_tiles_transposed(v4d (&G)[4][4], const double* __restrict__ A, int lane)
{
    const double* base = A + (lane & 15) * 64 + (lane >> 4);
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
        for (int tj = 0; tj < 4; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) G[ti][tj][r] = base[(16 * tj) * 64 + 16 * ti + 4 * r];
}

// G <- tile layout of the symmetric matrix whose lower triangle is A's (what LLT reads, Solver.cpp:76):
// S[a][b] = A[max(a,b)][min(a,b)].  Off-diagonal tiles are statically one side or the other; inside the diagonal
// tiles the side depends on the lane.
static DQQ_D void load_tiles_lower_symmetric(v4d (&G)[4][4], const double* __restrict__ A, int lane)
{
    const int g = lane >> 4, n = lane & 15;
    const double* rowmajor = A + g * 64 + n;   // + (16ti + 4r) * 64 + 16tj : A[16ti+4r+g][16tj+n]
    const double* colmajor = A + n * 64 + g;   // + (16tj) * 64 + 16ti + 4r : A[16tj+n][16ti+4r+g]
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
        for (int tj = 0; tj < 4; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o_row = (16 * ti + 4 * r) * 64 + 16 * tj, o_col = (16 * tj) * 64 + 16 * ti + 4 * r;
                if (ti > tj) G[ti][tj][r] = rowmajor[o_row];
                else if (ti < tj) G[ti][tj][r] = colmajor[o_col];
                else G[ti][tj][r] = (4 * r + g >= n) ? rowmajor[o_row] : colmajor[o_col];
            }
}

// Diagonal of the matrix <- d (one element per lane: lane l = entry l).  In tile (t,t) lane (g,n) holds the diagonal
// entry 16t+n in register n >> 2 iff (n & 3) == g.
static DQQ_D void set_tile_diagonal(v4d (&G)[4][4], double d, int lane)
{
    const int g = lane >> 4, n = lane & 15;
    const bool on_diag = (n & 3) == g;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const double dt = lane_gather(d, 16 * t + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) G[t][t][r] = (on_diag && (n >> 2) == r) ? dt : G[t][t][r];
    }
}

template <int KIND>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void fwd_dense_wave64_kernel(
    const double* __restrict__ P, const double* __restrict__ q, const double* __restrict__ l_n,
    const double* __restrict__ mu_c, const double* __restrict__ v_sign, double* __restrict__ x, long B, double eps,
    double mu, int max_iter, int adaptive, int* __restrict__ iters, int* __restrict__ ws, int use_worklist)
{
    // KIND 2 / 3 (box / signed box QP): l_n = l_min, mu_c = l_max per coordinate
    constexpr int N = 64;
    constexpr bool QP_LIKE = (KIND != 1);
    const int lane = threadIdx.x;
    const int xsrc = 4 * (lane & 15) + (lane >> 4);
    const long count = use_worklist ? (long)ws[kWsCount] : B;

    for (long w = blockIdx.x; w < count; w += gridDim.x) {
        const long prob = use_worklist ? (long)ws[kWsEntries + w] : w;
        const double* Pg = P + prob * (long)(N * N);
        WaveTile64 W;
        // ---- power_iteration, Solver.cpp:46-59 (the normalisation by a 1-ulp reciprocal square root)
        load_tiles_transposed(W.G, Pg, lane);
        double v = 0.125; // 1/sqrt(64); already of unit norm
        const int pi_steps = QP_LIKE ? 10 : 100;
        for (int k = 0; k < pi_steps; ++k) {
            const double Av = W.matvec(v, xsrc);
            const double s = wave_sum64(Av * Av);
            v = s > 0 ? Av * fast_rsqrt(s) : Av;
        }
        const double Lmax = wave_sum64(v * W.matvec(v, xsrc));
        RhoSchedule sched;
        sched.init(Lmax, mu);                                    // :72-73 / :531-532
        double rho = sched.rho;
        double mdiag = Pg[lane * (N + 1)] + (rho + mu);          // accumulated shifted diagonal, :75
        bool bad = false;

        const double qi = q[prob * N + lane];
        double rad = 0.0;
        if (KIND == 1) rad = l_n[prob * (N / 2) + lane / 2] * mu_c[prob * (N / 2) + lane / 2];
        double blo = 0.0, bhi = 0.0, bsg = 0.0;
        if (KIND >= 2) {
            blo = l_n[prob * N + lane];
            bhi = mu_c[prob * N + lane];
            if (KIND == 3) { const double vv = v_sign[prob * N + lane]; bsg = (double)((vv > 0) - (vv < 0)); } // :395
        }
        double qp = qi, l2 = 0.0, l2p = 0.0, u = 0.0;
        int it_done = 0;
        bool need_refactor = true;
        double inv_rho = 1.0 / rho;
        for (int it = 0; it < max_iter; ++it) {
            if (need_refactor) { // llt() + solveInPlace(Identity) of P + shift, Solver.cpp:76-77: W.G <- -(P + shift)^-1
                load_tiles_lower_symmetric(W.G, Pg, lane);
                set_tile_diagonal(W.G, mdiag, lane);
                block_sweep_inverse(W.G, lane, bad);
                need_refactor = false;
                inv_rho = 1.0 / rho;
            }
            it_done = it + 1;
            const double l = W.matvec((u + qp) - rho * l2, xsrc);        // :80 / :539 (W.G holds MINUS the inverse)
            qp = qi - mu * l;                                            // :81 / :540
            double z = kAlpha * l + (1 - kAlpha) * l2 + u * inv_rho;     // :82 / :541 (inv_rho = 1/rho)
            if (KIND == 0) {
                z = z < 0 ? 0 : z;
            } else if (KIND >= 2) {
                z = z < blo ? blo : z;                                   // cwiseMax(l_min), :219 / :396
                z = bhi < z ? bhi : z;                                   // cwiseMin(l_max), :220 / :397
                if (KIND == 3) {                                         // v o min(v o l_2, 0), :398
                    double mm = bsg * z;
                    mm = 0 < mm ? 0 : mm;
                    z = bsg * mm;
                }
            } else {                                                     // prox_circle, :505-519
                const double other = partner<1>(z);                      // lane ^ 1
                const double a = (lane & 1) ? other : z, b = (lane & 1) ? z : other;
                const double nrm = sqrt(a * a + b * b);
                if (nrm > rad) z = z * rad / nrm;
            }
            l2 = z;
            u += rho * (kAlpha * l + (1 - kAlpha) * l2p - l2);           // :83 / :543
            const double rd_i = QP_LIKE ? fabs(rho * (l2 - l2p)) : fabs(l2 - l2p);
            const double rp_i = fabs(l2 - (kAlpha * l + (1 - kAlpha) * l2p));
            l2p = l2;
            double rdm, res_prim;
            wave_max2(rd_i, rp_i, rdm, res_prim);
            const double res_dual = QP_LIKE ? rdm : rho * rdm;
            bool stop = res_dual < eps;                                  // :88
            if (KIND == 1) stop = (res_prim < eps + kEpsRel * sqrt(wave_sum64(l * l))) && stop; // :548
            if (stop) break;
            if (adaptive) {
                double delta = 0.0;
                if (sched.template update<QP_LIKE, false>(res_prim, res_dual, delta)) { // :90-120 / :550-580
                    mdiag += delta;
                    rho = sched.rho;
                    need_refactor = true;
                }
            }
        }
        const bool failed = bad || !(rho > 0.0) || !(rho < 1.79e308);
        x[prob * N + lane] = failed ? NAN : l2;
        if (iters != nullptr && lane == 0) iters[prob] = it_done;
    }
    // last wave out re-zeroes the work-list header (nothing to do when the list was empty)
    if (use_worklist && lane == 0) worklist_release(ws, count, (int)gridDim.x);
}

template <int KIND>
static hipError_t launch_wave64(const FwdArgs& a, bool use_worklist, hipStream_t s)
{
    // one wave per problem, dispatched by the hardware as waves retire (iteration counts differ per problem);
    // work-list mode: a fixed grid of 8 waves per CU strides over the list
    const long cap = 1L << 22;
    const unsigned grid = use_worklist ? 2048u : (unsigned)(a.B < cap ? a.B : cap);
    hipLaunchKernelGGL(fwd_dense_wave64_kernel<KIND>, dim3(grid), dim3(64), 0, s, a.P, a.q, a.l_n, a.mu, a.v, a.x, a.B,
                       a.eps, a.mu_prox, a.max_iter, a.adaptive, a.iters, a.ws, use_worklist ? 1 : 0);
    return hipGetLastError();
}

bool fwd_dense_wave64_supported(int N) { return N == 64; }

hipError_t launch_fwd_dense_wave64(int kind, const FwdArgs& a, bool use_worklist, hipStream_t s)
{
    if (a.B == 0) return hipSuccess;
    if (a.N != 64) return hipErrorInvalidValue;
    switch (kind) {
    case 0: return launch_wave64<0>(a, use_worklist, s);
    case 1: return launch_wave64<1>(a, use_worklist, s);
    case 2: return launch_wave64<2>(a, use_worklist, s);
    case 3: return launch_wave64<3>(a, use_worklist, s);
    default: return hipErrorInvalidValue;
    }
}

} // namespace dqq
