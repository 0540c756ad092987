// dense_wave64.hip -- general (dense P) forward solve for N = 64: ONE WAVE per problem, the whole problem in
// registers (BASELINE configs[4]: B=65536, N=64).
//
// Same algorithm as the reference (Solver::solveQP / solveQCQP / solveBoxQP / solveSignedBoxQP,
// Solver.cpp:61-123, 521-582, 198-261, 374-439): power iteration, adaptive-rho ADMM with the explicit inverse of
// P + (rho+mu) I rebuilt at every rho update (the reference's llt() + solveInPlace(Identity), :76-77, :100-101,
// :114-115).  What is different from the LDS wave-per-problem kernel (dense.hip / dense_core.h) is the execution model:
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
#include "wave_chol.h"
#include "wave_tile.h"

namespace dqq {

// The forward kernel takes any N <= 64: the matrix is embedded in (16 NT) x (16 NT), NT = ceil(N / 16), padded with
// the identity (PAD): the padded coordinates decouple (their rows of the inverse are unit vectors), their vector
// entries are zero from start to end, and sums / maxima over the wave are unaffected.

// G <- tile layout of A^T for the row-major n x n matrix A: G[ti][tj][r] of lane (g,l) = A[16tj+l][16ti+4r+g],
// so that WaveTile::matvec returns A x (no symmetry assumed).
template <int NT, bool PAD>
static DQQ_D void load_tiles_transposed(v4d (&G)[NT][NT], const double* __restrict__ A, int n, int lane)
{
    // addressing: uniform base (SGPRs, advanced per tile) + one 32-bit per-lane offset for all loads -- as a
    // per-lane 64-bit pointer plus constants the compiler keeps dozens of address pairs alive and spills them
    const int g = lane >> 4, l = lane & 15;
    const unsigned lo = l * n + g;
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int tj = 0; tj < NT; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * tj + l, col = 16 * ti + 4 * r + g;
                if (!PAD || (row < n && col < n)) G[ti][tj][r] = (A + ((16 * tj) * n + 16 * ti + 4 * r))[lo];
                else G[ti][tj][r] = (row == col) ? 1.0 : 0.0;
            }
}

// Diagonal of the matrix <- d (one element per lane: lane l = entry l).  In tile (t,t) lane (g,l) holds the diagonal
// entry 16t+l in register l >> 2 iff (l & 3) == g.
template <int NT>
static DQQ_D void set_tile_diagonal(v4d (&G)[NT][NT], double d, int lane)
{
    const int g = lane >> 4, l = lane & 15;
    const bool on_diag = (l & 3) == g;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const double dt = lane_gather(d, 16 * t + l);
#pragma unroll
        for (int r = 0; r < 4; ++r) G[t][t][r] = (on_diag && (l >> 2) == r) ? dt : G[t][t][r];
    }
}

// The lower triangle of the symmetric matrix (what every refactorisation starts from), kept in LDS between
// refactorisations: the off-diagonal lower tiles row-major with an odd row stride (17: a tile is read along its rows
// for the lower half of the matrix and along its columns for the mirrored upper half, both conflict-free), the
// diagonal tiles as packed lower triangles.  16.9 KiB for 4 x 4 tiles: eight waves per CU keep 135 of the 160 KiB.
// Round 2 re-read P from global memory at every rho update: 1.94x the algorithmic HBM traffic.
constexpr int kLowLd = 17;
template <int NT>
struct LowerLds {
    static constexpr int NOFF = NT * (NT - 1) / 2;
    static constexpr int DIAG0 = NOFF * 16 * kLowLd;
    static constexpr int DOUBLES = DIAG0 + NT * 136;
    static constexpr int tile_base(int ti, int tj) { return (ti * (ti - 1) / 2 + tj) * 16 * kLowLd; } // ti > tj
};

// The LDS image, written from the TRANSPOSED tile layout the power iteration leaves in registers
// (load_tiles_transposed: T[ti][tj][r] of lane (g,l) = A[16 tj + l][16 ti + 4 r + g]): the lower triangle of A is all in
// there, so the first factorisation needs no second pass over P in memory (round 5: that pass re-read the lower triangle
// row-wise AND column-wise -- 0.55 GB of the 2.76 GB a 65536 x 64 forward moved, 1.25x its algorithmic bytes).
// Entry (a, b) of A, a >= b, sits in tile (ta, tb) = (a / 16, b / 16): for ta > tb it is T[tb][ta][r](g, l) with
// a = 16 ta + l, b = 16 tb + 4 r + g -- stored at row l, column 4 r + g of the LDS tile (ta, tb).
template <int NT, bool PAD>
static DQQ_D void store_lower_to_lds_from_transposed(const v4d (&T)[NT][NT], double* __restrict__ lds, int n, int lane)
{
    using L = LowerLds<NT>;
    const int g = lane >> 4, l = lane & 15;
#pragma unroll
    for (int ta = 0; ta < NT; ++ta)
#pragma unroll
        for (int tb = 0; tb <= ta; ++tb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int al = l, bl = 4 * r + g;                       // position inside the tile
                double v = T[tb][ta][r];
                if (PAD) {                                              // the padding of the SYMMETRIC matrix is the identity
                    const int a = 16 * ta + al, b = 16 * tb + bl;
                    if (!(a < n && b < n)) v = (a == b) ? 1.0 : 0.0;
                }
                if (ta > tb) lds[L::tile_base(ta, tb) + al * kLowLd + bl] = v;
                else if (al >= bl) lds[L::DIAG0 + 136 * ta + al * (al + 1) / 2 + bl] = v;
            }
}

template <int NT>
static DQQ_D void load_lower_from_lds(v4d (&G)[NT][NT], const double* __restrict__ lds, int lane)
{
    using L = LowerLds<NT>;
    const int g = lane >> 4, l = lane & 15;
    const int rowpart = g * kLowLd + l;                 // S[16 ti + 4 r + g][16 tj + l], ti > tj: tile (ti,tj) along its rows
    const int colpart = l * kLowLd + g;                 // ti < tj: tile (tj,ti) entry (l, 4 r + g): along its columns
    const int tril = l * (l + 1) / 2 + g;               // diagonal tile, upper half: packed entry (l, 4 r + g)
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int tj = 0; tj < NT; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double v;
                if (ti > tj) v = lds[L::tile_base(ti, tj) + 4 * r * kLowLd + rowpart];
                else if (ti < tj) v = lds[L::tile_base(tj, ti) + 4 * r + colpart];
                else {
                    const int a = 4 * r + g;
                    v = lds[L::DIAG0 + 136 * ti + ((a >= l) ? a * (a + 1) / 2 + l : tril + 4 * r)];
                }
                G[ti][tj][r] = v;
            }
}

template <int KIND, int NT, bool PAD>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, NT == 4 ? 2 : (NT == 3 ? 3 : 4)))) void fwd_dense_wave64_kernel(
    const double* __restrict__ P, const double* __restrict__ q, const double* __restrict__ l_n,
    const double* __restrict__ mu_c, const double* __restrict__ v_sign, double* __restrict__ x, long B, int N, double eps,
    double mu, int max_iter, int adaptive, int* __restrict__ iters, int* __restrict__ ws, int use_worklist)
{
    // KIND 2 / 3 (box / signed box QP): l_n = l_min, mu_c = l_max per coordinate.  PAD: N < 16 NT.
    constexpr bool QP_LIKE = (KIND != 1);
    WorkClaim claim; // (launch.h: direct mode, or dynamic pick-up from the work-list)
    claim.open(ws, use_worklist, N, B);
    __shared__ __attribute__((aligned(16))) double s_lower[LowerLds<NT>::DOUBLES]; // one wave per workgroup
    __shared__ __attribute__((aligned(16))) double s_tr[16 * kTrLd];                // tile transposes of the sweep

    for (long w = blockIdx.x;; w += gridDim.x) {
        const long prob = claim.next(ws, B, w); // wave-uniform (SGPRs: P's addressing uses a scalar base)
        if (prob < 0) break;
        // (no claim ahead here, unlike the backward kernels: forward problems differ 2x in iteration count, and a
        // problem claimed at the start of a long one waits while other waves run dry -- measured: 3.79 -> 3.96 ms)
        // everything derived from the lane index is recomputed per problem: hoisted out of this loop (which runs
        // once per wave outside the work-list mode) those dozens of masks and offsets only occupy registers
        int lane = threadIdx.x;
        asm volatile("" : "+v"(lane));
        const int xsrc = 4 * (lane & 15) + (lane >> 4);
        const double* Pg = P + prob * (long)(N * N);
        // lanes >= N hold no coordinate (NT < 4) or a padded one (PAD): all their vector entries stay zero
        const bool live = (NT == 4 && !PAD) || lane < N;
        WaveTile<NT> W;
        // ---- power_iteration, Solver.cpp:46-59 (the normalisation by a 1-ulp reciprocal square root)
        load_tiles_transposed<NT, PAD>(W.G, Pg, N, lane);
        double v = live ? 1.0 / sqrt((double)N) : 0.0; // of unit norm up to rounding
        {
            const double s = wave_sum64(v * v);
            v = s > 0 ? v * fast_rsqrt(s) : v;
        }
        const int pi_steps = QP_LIKE ? 10 : 100;
        for (int k = 0; k < pi_steps; ++k) {
            const double Av = W.matvec(v, xsrc);
            const double s = wave_sum64(Av * Av);
            v = s > 0 ? Av * fast_rsqrt(s) : Av;
        }
        const double Lmax = wave_sum64(v * W.matvec(v, xsrc));
        // the lower triangle of P goes to LDS straight from these registers: every factorisation starts from there
        store_lower_to_lds_from_transposed<NT, PAD>(W.G, s_lower, N, lane);
        wave_lds_fence();
        RhoSchedule sched;
        sched.init(Lmax, mu);                                    // :72-73 / :531-532
        double rho = sched.rho;
        // accumulated shifted diagonal, :75 -- P's diagonal from the LDS image just written (lane i: entry (i, i) of diagonal
        // tile i / 16), not from memory again (64 lanes x one 64-byte line each)
        const int dl_ = lane & 15;
        double mdiag = live ? s_lower[LowerLds<NT>::DIAG0 + 136 * ((lane >> 4) < NT ? (lane >> 4) : 0) + dl_ * (dl_ + 1) / 2 + dl_] + (rho + mu) : 1.0;
        bool bad = false;

        const double qi = live ? q[prob * N + lane] : 0.0;
        double rad = 0.0;
        if (KIND == 1) rad = live ? l_n[prob * (N / 2) + lane / 2] * mu_c[prob * (N / 2) + lane / 2] : 0.0;
        double blo = 0.0, bhi = 0.0, bsg = 0.0;
        if (KIND >= 2 && live) {
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
                load_lower_from_lds<NT>(W.G, s_lower, lane);   // (never from memory: P was read once, for the power iteration)
                set_tile_diagonal<NT>(W.G, mdiag, lane);
                block_sweep_inverse<NT>(W.G, lane, bad, s_tr);
                need_refactor = false;
                inv_rho = 1.0 / rho;
            }
            it_done = it + 1;
#if defined(DQQ_SALU_PROBE)
            // measurement only (tools/ab_salu_probe.sh): DQQ_SALU_PROBE extra scalar instructions per ADMM
            // iteration.  Result (profiles/r07_ab_salu_probe.txt): +32 per iteration = +5.2 % of the kernel, +64 = +12.6 % --
            // the scalar stream is NOT hidden behind the other wave's vector work at two waves per SIMD; a scalar
            // instruction costs this kernel about what a vector one does.  Of the ~49 an iteration executes, 15 are
            // hazard nops on the true dependencies of the residual reduction and ~22 the rho schedule's wave-uniform
            // control flow (dearer as vector code); splitting the loop into factorisation epochs removed 4: 3655 vs
            // 3660 us, no measurable change, not kept.
#pragma unroll
            for (int sp = 0; sp < DQQ_SALU_PROBE; ++sp) { int sd; asm volatile("s_mov_b32 %0, 0" : "=s"(sd)); }
#endif
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
            const double rd_i = QP_LIKE ? rho * (l2 - l2p) : l2 - l2p;         // signed: the maxima take |.|
            const double rp_i = l2 - (kAlpha * l + (1 - kAlpha) * l2p);
            l2p = l2;
            double rdm, res_prim;
            wave_max2_abs(rd_i, rp_i, rdm, res_prim);
            const double res_dual = QP_LIKE ? rdm : rho * rdm;
            bool stop = res_dual < eps;                                  // :88
            if (KIND == 1) stop = (res_prim < eps + kEpsRel * sqrt(wave_sum64(l * l))) && stop; // :548
            if (stop) break;
            if (adaptive) {
                double delta = 0.0;
                if (sched.template update<QP_LIKE, false>(res_prim, res_dual, delta)) { // :90-120 / :550-580
                    mdiag += live ? delta : 0.0;
                    rho = sched.rho;
                    need_refactor = true;
                }
            }
        }
        const bool failed = bad || !(rho > 0.0) || !(rho < 1.79e308);
        if (live) x[prob * N + lane] = failed ? NAN : l2;
        if (iters != nullptr && lane == 0) iters[prob] = it_done;
    }
    // last wave out re-zeroes the work-list header (nothing to do when the list was empty)
    if (use_worklist && threadIdx.x == 0) worklist_release(ws, claim.count, (int)gridDim.x);
}

template <int KIND, int NT, bool PAD>
static hipError_t launch_wave64(const FwdArgs& a, bool use_worklist, hipStream_t s)
{
    // one wave per problem, dispatched by the hardware as waves retire (iteration counts differ per problem);
    // work-list mode: a fixed grid of as many waves as the chip holds claims the entries one at a time
    const long cap = 1L << 22;
    const long wl = 1024L * (NT == 4 ? 2 : (NT == 3 ? 3 : 4)); // the waves the chip holds of this instantiation
    const unsigned grid = (unsigned)(a.B < (use_worklist ? wl : cap) ? (a.B > 0 ? a.B : 1) : (use_worklist ? wl : cap));
    return launch(fwd_dense_wave64_kernel<KIND, NT, PAD>, dim3(grid), dim3(64), 0, s, a.P, a.q, a.l_n, a.mu, a.v, a.x, a.B,
                  a.N, a.eps, a.mu_prox, a.max_iter, a.adaptive, a.iters, a.ws, use_worklist ? 1 : 0);
}

template <int KIND>
static hipError_t launch_wave64_kind(const FwdArgs& a, bool use_worklist, hipStream_t s)
{
    if (a.N == 64) return launch_wave64<KIND, 4, false>(a, use_worklist, s);
    if (a.N > 48) return launch_wave64<KIND, 4, true>(a, use_worklist, s);
    if (a.N == 48) return launch_wave64<KIND, 3, false>(a, use_worklist, s);
    if (a.N > 32) return launch_wave64<KIND, 3, true>(a, use_worklist, s);
    if (a.N == 32) return launch_wave64<KIND, 2, false>(a, use_worklist, s);
    return launch_wave64<KIND, 2, true>(a, use_worklist, s);
}

// every 16 < N <= 64 (N <= 16: the lane / team kernels hold the whole problem per lane or per 16 lanes)
bool fwd_dense_wave64_supported(int N) { return N > 16 && N <= 64; }

hipError_t launch_fwd_dense_wave64(int kind, const FwdArgs& a, bool use_worklist, hipStream_t s)
{
    if (a.B == 0) return hipSuccess;
    if (!fwd_dense_wave64_supported(a.N)) return hipErrorInvalidValue;
    switch (kind) {
    case 0: return launch_wave64_kind<0>(a, use_worklist, s);
    case 1: return launch_wave64_kind<1>(a, use_worklist, s);
    case 2: return launch_wave64_kind<2>(a, use_worklist, s);
    case 3: return launch_wave64_kind<3>(a, use_worklist, s);
    default: return hipErrorInvalidValue;
    }
}

// ---------------------------------------------------------------- backward (QP), 16 < N <= 64
// One wave per problem; the composition of pybindings.cpp:24-30 -> Solver::dualFromPrimalQP (Solver.cpp:125-134),
// Solver::solveDerivativesQP (:136-196), Solver::iterative_refinement (:15-44) and the gradient assembly of
// qcqp.py:48-51, on the system in the ORIGINAL index order with the active rows / columns masked
// (A~[a][k] = P[a][k] if a and k are inactive, l_a if a = k is active, 0 otherwise -- a symmetric permutation of the
// reference's blkdiag(diag(l_A), P_II), see small_bwd_core.h).
//
// P is read ONCE (tile layout of P^T in registers): gamma = -(P l + q), then the masks are applied in place and A^T b
// follows from the same registers.  The UPPER tiles of K = A~ A~^T + mu I are accumulated on the matrix cores
// (NT (NT+1) / 2 tiles), kept twice: one copy is factored and inverted in place (wave_chol.h: block Cholesky + explicit
// inverse, the reference's llt() + solveInPlace(Identity), :22-23), the other serves the refinement residual
// K x - A^T b (:30).  Nothing is parked in memory (round 2 parked K in the grad_P slot and streamed it back twice:
// 2.6x the algorithmic traffic); products with the symmetric matrices use the upper tiles and their transposes.
// N is padded with the identity to 16 NT (PAD): a padded coordinate is inactive with x = g = 0, its block of K is
// 1 + mu and its solution entry stays zero.
template <int NT, bool PAD>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, NT >= 3 ? 2 : 3))) void bwd_dense_chol_qp_kernel(
    const double* __restrict__ P, const double* __restrict__ q, const double* __restrict__ x,
    const double* __restrict__ grad_x, double* __restrict__ grad_P, double* __restrict__ grad_q, long B, int N,
    double dual_eps, int* __restrict__ ir_steps, int* __restrict__ ws, int use_worklist)
{
    WorkClaim claim; // (launch.h: direct mode, or dynamic pick-up from the work-list)
    claim.open(ws, use_worklist, N, B);
    __shared__ __attribute__((aligned(16))) double s_trb[16 * kTrLd]; // tile transposes (one wave per workgroup)
    for (long w = blockIdx.x;; w += gridDim.x) {
        const long prob = claim.next(ws, B, w); // wave-uniform (SGPRs: P's addressing uses a scalar base)
        if (prob < 0) break;
        claim.ahead_issue(ws); // (work-list: the ticket for this wave's next problem travels while this one is solved)
        int lane = threadIdx.x;
        asm volatile("" : "+v"(lane)); // see the forward kernel: nothing lane-derived is hoisted out of the loop
        const int g = lane >> 4, n = lane & 15;
        const int xsrc = 4 * n + g;
        const double* Pg = P + prob * (long)(N * N);
        const bool live = (NT == 4 && !PAD) || lane < N;
        const double xi = live ? x[prob * N + lane] : 0.0, gi = live ? grad_x[prob * N + lane] : 0.0;
        const double qi = live ? q[prob * N + lane] : 0.0;
        WaveChol<NT> C;
        v4d Kc[NT][NT];
        double Ab;
        bool is_act;
        {
            WaveTile<NT> A;
            load_tiles_transposed<NT, PAD>(A.G, Pg, N, lane);                 // A.G[tk][ta][r] = P[16ta+n][16tk+4r+g]
            // dualFromPrimalQP, Solver.cpp:125-134, and the active set of solveDerivativesQP, :139-147
            double gamma = -(A.matvec(xi, xsrc) + qi);
            if (xi > dual_eps) gamma = 0;
            is_act = live && gamma < -kActiveEps;
            const unsigned long long am = __ballot(is_act);
#pragma unroll
            for (int tk = 0; tk < NT; ++tk) {                                 // A~ in place, :148-158
                const double dk = lane_gather(xi, 16 * tk + n);
#pragma unroll
                for (int ta = 0; ta < NT; ++ta)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool row_act = (am >> (16 * ta + n)) & 1ull, col_act = (am >> (16 * tk + 4 * r + g)) & 1ull;
                        double t = (row_act || col_act) ? 0.0 : A.G[tk][ta][r];
                        if (ta == tk) t = (row_act && 4 * r + g == n) ? dk : t;
                        A.G[tk][ta][r] = t;
                    }
            }
            Ab = A.matvec(is_act ? 0.0 : gi, xsrc);                           // A^T b (:19), b = [0; grad_I]
            if (is_act) Ab = 0.0;
            const v4d zero = {0.0, 0.0, 0.0, 0.0};                            // K = A~ A~^T + mu_ir I (:20-21), upper tiles
            const bool on_diag = (n & 3) == g;
#pragma unroll
            for (int ta = 0; ta < NT; ++ta)
#pragma unroll
                for (int tb = ta; tb < NT; ++tb) {
                    v4d acc = zero;
#pragma unroll
                    for (int tk = 0; tk < NT; ++tk) acc = tile_xty(acc, A.G[tk][ta], A.G[tk][tb]);
                    if (ta == tb) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[r] += (on_diag && (n >> 2) == r) ? kMuIr : 0.0;
                    }
                    C.U[ta][tb] = acc;
                    Kc[ta][tb] = acc;
                }
        }
        claim.ahead_entry(ws, B);
        bool bad = false;
        C.factor(lane, bad);                                                  // :22
        C.invert_in_place(lane, s_trb);                                              // :23; C.U = upper tiles of K^-1
        const double KinvAb = sym_upper_matvec<NT>(C.U, Ab, xsrc, lane, s_trb);      // :27
        double xs = 0.0;
        IrControl ctl;
        ctl.init();
        int steps = 0;
        for (int it = 0; it < kIrMaxIter; ++it) {
            steps = it + 1;
            xs = it == 0 ? KinvAb : KinvAb + kMuIr * sym_upper_matvec<NT>(C.U, xs, xsrc, lane, s_trb); // :29 (first body: x = 0)
            const double d = sym_upper_matvec<NT>(Kc, xs, xsrc, lane, s_trb) - Ab;   // :30
            const double res = sqrt(wave_sum64(d * d));                       // :31
            if (ctl.update(res)) break;                                       // :32-41
        }
        claim.ahead_done(ws, B);
        const double dl = bad ? NAN : (is_act ? 0.0 : xs);                    // :187-191
        if (live && grad_q != nullptr) grad_q[prob * N + lane] = -dl;        // qcqp.py:49
        if (grad_P != nullptr) {                                              // qcqp.py:48: -(dl l^T)
            double* Gp = grad_P + prob * (long)(N * N);
            for (int k = 0; k < N; ++k) {
                const double v = -(lane_bcast(dl, k) * xi);
                if (live) __builtin_nontemporal_store(v, Gp + k * N + lane);
            }
        }
        if (ir_steps != nullptr && lane == 0) ir_steps[prob] = steps;
    }
    if (use_worklist && threadIdx.x == 0) worklist_release(ws, claim.count, (int)gridDim.x);
}

template <int NT, bool PAD>
static hipError_t launch_bwd_chol(const BwdArgs& a, bool use_worklist, hipStream_t s)
{
    const long cap = 1L << 22;
    const long wl = 1024L * (NT >= 3 ? 2 : 3); // the waves the chip holds of this instantiation
    const unsigned grid = (unsigned)(a.B < (use_worklist ? wl : cap) ? (a.B > 0 ? a.B : 1) : (use_worklist ? wl : cap));
    return launch(bwd_dense_chol_qp_kernel<NT, PAD>, dim3(grid), dim3(64), 0, s, a.P, a.q, a.x, a.grad_x, a.grad_P, a.grad_q,
                  a.B, a.N, a.epsilon, a.ir_steps, a.ws, use_worklist ? 1 : 0);
}

// QP backward: every 16 < N <= 64, everything in registers, nothing allocated
bool bwd_dense_wave64_supported(int kind, int N) { return kind == 0 && N > 16 && N <= 64; }

hipError_t launch_bwd_dense_wave64(int kind, const BwdArgs& a, bool use_worklist, hipStream_t s)
{
    if (a.B == 0) return hipSuccess;
    if (!bwd_dense_wave64_supported(kind, a.N)) return hipErrorInvalidValue;
    if (a.N == 64) return launch_bwd_chol<4, false>(a, use_worklist, s);
    if (a.N > 48) return launch_bwd_chol<4, true>(a, use_worklist, s);
    if (a.N == 48) return launch_bwd_chol<3, false>(a, use_worklist, s);
    if (a.N > 32) return launch_bwd_chol<3, true>(a, use_worklist, s);
    if (a.N == 32) return launch_bwd_chol<2, false>(a, use_worklist, s);
    return launch_bwd_chol<2, true>(a, use_worklist, s);
}

} // namespace dqq
