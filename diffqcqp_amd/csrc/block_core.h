// block_core.h -- LDS-resident blocked factorisation on the f64 matrix cores, shared by the workgroup-per-
// problem kernels (dense_block.hip: forward and QP backward at N = 32 / 64; bwd_block.hip: QCQP / box QP
// backward with 48- or 96-unknown systems).  One 256-thread workgroup works on one N x N symmetric positive
// definite matrix held row-major in LDS (row stride BlockGeom<N>::LD): blocked right-looking Cholesky with
// 16-column panels, L^-1 by blocked forward substitution, and M^-1 = L^-T L^-1 (or a plain Gram matrix) as
// 16x16 tile products (v_mfma_f64_16x16x4_f64).  See dense_block.hip for the measurements.
#pragma once

#include "common.h"

namespace dqq {

template <int N>
struct BlockGeom {
    static constexpr int T = 256;
    static constexpr int R = (N == 32 || N == 64) ? T / N : 2; // row padding (doubles) of the row-major regions
    static constexpr int LD = N + R;           // row stride of the row-major regions
    static constexpr int REGION = N * LD;      // doubles per region
    static constexpr int VEC = 8 * N + N;      // WaveRows<N>::LDS_DOUBLES
    static constexpr size_t LDS_BYTES = sizeof(double) * (2 * REGION + VEC + 2); // + the failure flag
};

typedef double v4d __attribute__((ext_vector_type(4)));

// value held by lane `src` (wave-uniform) of the calling wave
DQQ_D double lane_bcast(double v, int src)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

// One 16x16 f64 tile product on the matrix cores: acc += sum_{k<K} A[i][k] * B[k][j] with
// A[i][k] = a[i*ars + k*acs], B[k][j] = b[k*brs + j*bcs]  (gfx950 v_mfma_f64_16x16x4_f64: lane l feeds
// A[l&15][l>>4], B[l>>4][l&15]; result register r of lane l is C[(l>>4) + 4r][l&15]).
DQQ_D v4d tile_mma(v4d acc, const double* a, int ars, int acs, const double* b, int brs, int bcs, int K, int l,
                   bool negate_a)
{
    const double* ap = a + (l & 15) * ars + (l >> 4) * acs;
    const double* bp = b + (l >> 4) * brs + (l & 15) * bcs;
    // 16 columns of K per trip: all eight operands are loaded before the first of the four dependent MFMAs
    // is issued (one LDS latency per trip instead of four), and two accumulators halve the dependent chain
    v4d acc2 = {0.0, 0.0, 0.0, 0.0};
    for (int k = 0; k < K; k += 16) {
        double av[4], bv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            av[u] = ap[(k + 4 * u) * acs];
            bv[u] = bp[(k + 4 * u) * brs];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const double x = negate_a ? -av[u] : av[u];
            if (u & 1) acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, bv[u], acc2, 0, 0, 0);
            else acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, bv[u], acc, 0, 0, 0);
        }
    }
    return acc + acc2;
}

// Diagonal block kb of the blocked factorisation, in registers (lanes 0-15 = rows of the block; the other
// lanes mirror them): right-looking Cholesky of the 16x16 block of W, then its inverse (column c per lane
// c) into the diagonal block of LinvT (LinvT[c][i] = (L^-1)[i][c]).  Broadcasts via v_readlane, no LDS
// round trips inside.  EVERY wave of the workgroup runs this on the same input and stores the same bits
// to the same place: nobody has to wait for a designated wave, and a wave only needs its own stores to
// be visible (wave_lds_fence) before it reads the block back.  The factor L11 itself is not stored --
// nothing downstream reads it (the panel solve, the inverse and the product use L11^-1).
template <int N>
DQQ_D void diag_block_factor(const double* W, double* LinvT, int kb, int l, bool& bad)
{
    using G = BlockGeom<N>;
    const int row = (l & 15);
    double w[16], rinv[16];
    const double* wrow = W + (16 * kb + row) * G::LD + 16 * kb;
#pragma unroll
    for (int j = 0; j < 16; ++j) w[j] = wrow[j];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const double d = lane_bcast(w[k], k);
        bad = bad || !(d > 0.0);
        const double rs = fast_rsqrt(d);
        rinv[k] = rs;
        w[k] = (row == k) ? d * rs : w[k] * rs;
#pragma unroll
        for (int j = k + 1; j < 16; ++j) w[j] -= w[k] * lane_bcast(w[k], j);
    }
    double y[16]; // column `row` of the inverse of the block
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        double t = (row == i) ? 1.0 : 0.0;
#pragma unroll
        for (int j = 0; j < i; ++j) t -= lane_bcast(w[j], i) * y[j];
        y[i] = t * rinv[i];
    }
    if (l < 16) {
        double* yrow = LinvT + (16 * kb + row) * G::LD + 16 * kb;
#pragma unroll
        for (int j = 0; j < 16; ++j) yrow[j] = y[j];
    }
}

// Blocked (16-column panels) right-looking Cholesky of W + L^-1 (as LinvT), 256 threads.  Per panel: the
// diagonal block in registers (every wave, see above), panel solve and trailing update as 16x16x16 tile
// products on the f64 matrix cores; two workgroup barriers per panel.  The off-diagonal blocks of L^-1
// are then built column by column, one block column per wave, which needs no workgroup barrier at all:
//   Linv[i][j] = -Linv[i][i] * sum_{k=j}^{i-1} L[i][k] * Linv[k][j]     (Linv[a][b] lives at LinvT[b][a])
// only depends on L (complete), the diagonal blocks (every wave has them) and blocks of the same column.
// On return (after a barrier) LinvT holds L^-T on and right of its diagonal blocks; the blocks left of
// them are scratch.  W's diagonal blocks keep the input, its strict lower blocks hold L.
template <int N, bool DIAG_ALL_WAVES = false>
DQQ_D void block_cholesky_and_inverse(double* W, double* LinvT, double* fail_flag, int t, bool& bad)
{
    using G = BlockGeom<N>;
    constexpr int NT = N / 16;
    const int wave = t >> 6, l = t & 63;
    for (int kb = 0; kb < NT; ++kb) {
        if (kb > 0) __syncthreads(); // the trailing update of panel kb-1 is complete
        if (DIAG_ALL_WAVES) {
            diag_block_factor<N>(W, LinvT, kb, l, bad);
            wave_lds_fence();
        } else {
            if (wave == 0) diag_block_factor<N>(W, LinvT, kb, l, bad);
            __syncthreads();
        }
        if (kb == NT - 1) break;
        // panel: L[ib][kb] = A[ib][kb] * L11^-T ; (L11^-T)[k][j] = Linv11[j][k] = LinvT[16kb+k][16kb+j]
        for (int ib = kb + 1 + wave; ib < NT; ib += 4) {
            v4d acc = {0.0, 0.0, 0.0, 0.0};
            acc = tile_mma(acc, W + (16 * ib) * G::LD + 16 * kb, G::LD, 1, LinvT + (16 * kb) * G::LD + 16 * kb, G::LD, 1,
                           16, l, false);
            wave_lds_fence(); // all lanes have read the A tile before it is overwritten by L
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) W[(16 * ib + (l >> 4) + 4 * rg) * G::LD + 16 * kb + (l & 15)] = acc[rg];
        }
        __syncthreads();
        // trailing update: A[ib][jb] -= L[ib][kb] * L[jb][kb]^T for kb < jb <= ib
        int tile = 0;
        for (int ib = kb + 1; ib < NT; ++ib)
            for (int jb = kb + 1; jb <= ib; ++jb, ++tile) {
                if ((tile & 3) != wave) continue;
                v4d acc;
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) acc[rg] = W[(16 * ib + (l >> 4) + 4 * rg) * G::LD + 16 * jb + (l & 15)];
                acc = tile_mma(acc, W + (16 * ib) * G::LD + 16 * kb, G::LD, 1, W + (16 * jb) * G::LD + 16 * kb, 1, G::LD,
                               16, l, true);
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) W[(16 * ib + (l >> 4) + 4 * rg) * G::LD + 16 * jb + (l & 15)] = acc[rg];
            }
    }
    if (t == 0 && bad) *fail_flag = 1.0; // non-positive pivot: poison the result (NaN)
    // block column j of L^-1 by wave j.  Per-wave 16x16 scratch tile: a block of LinvT LEFT of the diagonal
    // (last block row, block column j) -- nothing else ever touches those blocks.
    for (int j = wave; j < NT - 1; j += 4) {
        double* T = LinvT + (16 * (NT - 1)) * G::LD + 16 * j;
        for (int i = j + 1; i < NT; ++i) {
            v4d acc = {0.0, 0.0, 0.0, 0.0};
            // A = L[16i.., 16j..16i), B[k][c] = Linv[16j+k][16j+c] = LinvT[(16j+c)*LD + 16j+k]
            acc = tile_mma(acc, W + (16 * i) * G::LD + 16 * j, G::LD, 1, LinvT + (16 * j) * G::LD + 16 * j, 1, G::LD,
                           16 * (i - j), l, false);
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) T[((l >> 4) + 4 * rg) * G::LD + (l & 15)] = acc[rg];
            wave_lds_fence();
            v4d res = {0.0, 0.0, 0.0, 0.0};
            // A = Linv[16i+a][16i+k] = LinvT[(16i+k)*LD + 16i+a]  (negated), B = T
            res = tile_mma(res, LinvT + (16 * i) * G::LD + 16 * i, 1, G::LD, T, G::LD, 1, 16, l, true);
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) // element (row a, col c) of Linv[i][j] -> LinvT[16j+c][16i+a]
                LinvT[(16 * j + (l & 15)) * G::LD + 16 * i + (l >> 4) + 4 * rg] = res[rg];
            wave_lds_fence();
        }
    }
    __syncthreads();
}

// Minv = L^-T L^-1 = LinvT LinvT^T (row-major, stride LD) on the f64 matrix cores.  Wave w owns the
// tile row ti = w (N = 64: 4 waves x 4 tiles; N = 32: waves 0-1 x 2 tiles).
// Minv[a][b] = sum_k LinvT[a][k] * LinvT[b][k]; LinvT rows are zero left of the diagonal block, so
// tile (ti,tj) only needs k >= 16*max(ti,tj).  Tiles (ti,tj) and (tj,ti) sum the same products in the
// same order: the result is bitwise symmetric.  FULL: rows without structure (plain Gram matrix A A^T).
template <int N, bool FULL = false>
DQQ_D void block_inverse_product(const double* LinvT, double* out, int t)
{
    using G = BlockGeom<N>;
    constexpr int NT = N / 16;
    const int wave = t >> 6, l = t & 63;
    for (int ti = wave; ti < NT; ti += 4)
    for (int tj = 0; tj < NT; ++tj) {
        v4d acc = {0.0, 0.0, 0.0, 0.0};
        const double* arow = LinvT + (16 * ti + (l & 15)) * G::LD + (l >> 4);
        const double* brow = LinvT + (16 * tj + (l & 15)) * G::LD + (l >> 4);
        const int s0 = FULL ? 0 : 4 * (ti > tj ? ti : tj);
        v4d acc2 = {0.0, 0.0, 0.0, 0.0};
        for (int s = s0; s < N / 4; s += 4) { // 16 columns of k per trip (s0 is a multiple of 4)
            double av[4], bv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { av[u] = arow[4 * (s + u)]; bv[u] = brow[4 * (s + u)]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (u & 1) acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc2, 0, 0, 0);
                else acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
            }
        }
        acc = acc + acc2;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int row = 16 * ti + (l >> 4) + 4 * rg, col = 16 * tj + (l & 15);
            out[row * G::LD + col] = acc[rg];
        }
    }
}


} // namespace dqq
