// wave_chol.h -- block Cholesky of a symmetric positive definite matrix of up to 64 x 64 held by ONE wave64 in
// registers (tile layout of wave_tile.h), and the solves with its factor.  The backward's systems
// K = A^T A + 1e-7 I (Solver::iterative_refinement, Solver.cpp:15-44) are where it matters: with a singular or
// rank-deficient P (the reference's own G4, Solver.cpp:899-907) cond(K) reaches 1e9 and beyond, and the in-place
// block Gauss-Jordan sweep of wave_tile.h -- forward stable only: its explicit inverse carries an absolute error of
// cond(K) eps |K^-1| -- returned gradients that were off by O(1) (tests/test_gpu_reference_inputs.py).  Cholesky
// + substitution is backward stable like the reference's llt(): measured 1e-9 on the same inputs.
//
//   K = U^T U, U upper block-triangular with 16 x 16 tiles.  Stored in the UPPER tiles of `U`:
//     U[K][J], K < J : the tile U_KJ = L_KK^-1 A_KJ (A = what is left of K after the steps before K)
//     U[K][K]        : W_K = L_KK^-T, the INVERSE of the diagonal block of the factor, transposed (tile [k][c] =
//                      (L_KK^-1)[c][k]) -- the panel products and the substitutions only ever need that inverse
//   Every tile product is an X^T Y on the f64 matrix cores (tile_xty), register to register:
//     U_KJ  = W_K^T A_KJ                    (panel)
//     A_IJ -= U_KI^T U_KJ, K < I <= J       (trailing update, upper tiles only: 10 products for 4 x 4 tiles)
//     T^T   = tile_xty(T, I16)              (a transpose is a product with the identity tile: exact)
//   The 16 x 16 diagonal block is factored with lane = row in each 16-lane row of the wave (DPP row_newbcast
//   operands, as the pivot inversion of wave_tile.h): right-looking Cholesky, then the rows of L^-1 by back
//   substitution, 120 + 120 v_fmac_f64_dpp.
//   Solve K x = b: forward U^T y = b, backward U x = y, block by block.  A product T^T v of a tile with 16 vector
//   entries is 4 v_fmac_f64_dpp per lane (lane (g,n) holds T[4r+g][n]; the entries v[4r+g] come from lane r of the
//   lane's 16-lane row) + an all-reduce over the four 16-lane rows (two permlane swaps).
//   Lower tiles of `U` are never touched (dead registers).
#pragma once

#include <type_traits>

#include "wave_tile.h"

namespace dqq {

template <int I, int E, class F>
DQQ_D void static_for(F&& f)
{
    if constexpr (I < E) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, E>(f);
    }
}

// wave-uniform value -> SGPR pair
DQQ_D double to_sgpr(double v)
{
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)),
                            __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

// sum over the four 16-lane rows of the wave, result in all of them (bitwise identical: a+b == b+a)
DQQ_D double allreduce_rows(double a)
{
    double lo, hi, e, o;
    swap32(a, a, lo, hi); // lo = rows (0,1,0,1), hi = rows (2,3,2,3)
    const double s = lo + hi;
    swap16(s, s, e, o);   // e = rows (0,0,2,2), o = rows (1,1,3,3) of s
    return e + o;
}

// the identity as a tile: register r of lane (g,n) = [4r+g == n]
DQQ_D v4d identity_tile(int lane)
{
    const int g = lane >> 4, n = lane & 15;
    v4d I;
#pragma unroll
    for (int r = 0; r < 4; ++r) I[r] = (4 * r + g == n) ? 1.0 : 0.0;
    return I;
}

DQQ_D v4d tile_transpose(const v4d& T, const v4d& I16)
{
    const v4d zero = {0.0, 0.0, 0.0, 0.0};
    return tile_xty(zero, T, I16); // [i][j] = sum_k T[k][i] I[k][j] = T[j][i]
}

// acc += sum_r T[r] * (lane BC0 + r of this lane's 16-lane row of x0)
template <int BC0>
DQQ_D void tile_dot4(double& acc, const v4d& T, double x0)
{
    fmac_bcast<BC0 + 0>(acc, x0, T[0]);
    fmac_bcast<BC0 + 1>(acc, x0, T[1]);
    fmac_bcast<BC0 + 2>(acc, x0, T[2]);
    fmac_bcast<BC0 + 3>(acc, x0, T[3]);
}

// ------------------------------------------------------------------------------------------------------
// 16 x 16 diagonal block: B (symmetric positive definite, tile layout) -> W = L^-T as a tile, B = L L^T.
// Lane (.,n) of every 16-lane row holds row n: a[j] = B[n][j]; all four rows of the wave do the same work.
template <int K>
DQQ_D void chol16_step(double (&a)[16], double (&rinv)[16], bool& bad)
{
    const double d = __builtin_amdgcn_update_dpp(0.0, a[K], 0x150 + K, 0xf, 0xf, true); // a_kk (row_newbcast:K)
    bad = bad | (__ballot(!(d > 0.0)) != 0);
    const double rs = fast_rsqrt(d);
    rinv[K] = to_sgpr(rs);                      // 1 / L_kk, wave-uniform
    const double ak = dpp_source(a[K] * rs);    // column K of L: L[n][K] in lane n (n >= K; lanes above hold junk
    a[K] = ak;                                  // that only ever feeds their own, unused, entries)
    const double nc = -ak;
    static_for<K + 1, 16>([&](auto jc) __attribute__((always_inline)) {
        constexpr int J = decltype(jc)::value;
        fmac_bcast<J>(a[J], ak, nc);            // a[n][J] -= L[J][K] L[n][K]
    });
}

// row n of L^-1 in lane n, entry j, for j = 15 .. 0:  z_j L_jj = [n == j] - sum_{i > j} z_i L_ij ; a[j] <- -z_j
template <int J>
DQQ_D void linv16_step(double (&a)[16], const double (&rinv)[16], int n)
{
    double acc = (n == J) ? 1.0 : 0.0;
    if constexpr (J < 15) {
        const double col = dpp_source(a[J]);    // L[i][J] in lane i
        static_for<J + 1, 16>([&](auto ic) __attribute__((always_inline)) {
            constexpr int I = decltype(ic)::value;
            fmac_bcast<I>(acc, col, a[I]);      // += L[I][J] * (-z_I)
        });
    }
    a[J] = -(acc * rinv[J]);
}

DQQ_D v4d chol16_inverse_factor(const v4d& T, int lane, bool& bad)
{
    const int n = lane & 15;
    double a[16], rinv[16];
    // all-gather over the four 16-lane rows: a[4r + g'] = T[r] of lane (g', n) = B[4r + g'][n] = B[n][4r + g']
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        double h0, h1;
        swap32(T[r], T[r], h0, h1);
        swap16(h0, h0, a[4 * r + 0], a[4 * r + 1]);
        swap16(h1, h1, a[4 * r + 2], a[4 * r + 3]);
    }
    static_for<0, 16>([&](auto kc) __attribute__((always_inline)) { chol16_step<decltype(kc)::value>(a, rinv, bad); });
    static_for<0, 16>([&](auto jc) __attribute__((always_inline)) { linv16_step<15 - decltype(jc)::value>(a, rinv, n); });
    // tile layout: W[r] of lane (g,n) = W[4r+g][n] = (L^-1)[n][4r+g] = -a[4r+g]; the choice by g is made by the swaps
    // (all four rows hold the same values; see diag16_inverse)
    v4d W;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        double lo, hi, pick, unused;
        swap16(a[4 * r + 0], a[4 * r + 1], lo, unused);
        swap16(a[4 * r + 2], a[4 * r + 3], hi, unused);
        swap32(lo, hi, pick, unused);
        W[r] = -pick;
    }
    return W;
}

// ------------------------------------------------------------------------------------------------------
template <int NT>
struct WaveChol {
    v4d U[NT][NT]; // upper tiles only (see the header); in: the upper tiles of K

    template <int K>
    DQQ_D void factor_step(int lane, bool& bad)
    {
        const v4d zero = {0.0, 0.0, 0.0, 0.0};
        const v4d W = chol16_inverse_factor(U[K][K], lane, bad);
        U[K][K] = W;
#pragma unroll
        for (int J = K + 1; J < NT; ++J) U[K][J] = tile_xty(zero, W, U[K][J]);
#pragma unroll
        for (int I = K + 1; I < NT; ++I) {
            const v4d nX = -U[K][I];
#pragma unroll
            for (int J = I; J < NT; ++J) U[I][J] = tile_xty(U[I][J], nX, U[K][J]);
        }
    }

    DQQ_D void factor(int lane, bool& bad)
    {
        static_for<0, NT>([&](auto kc) __attribute__((always_inline)) { this->template factor_step<decltype(kc)::value>(lane, bad); });
    }

    // After factor(): the explicit inverse K^-1 = L^-T L^-1 (what the reference's llt().solveInPlace(Identity)
    // leaves, Solver.cpp:22-23), upper tiles, in place.  Two passes over the same registers:
    //   1. L^-1 block column j into storage row j (slot [j][i] <- the tile (L^-1)_ij, i >= j):
    //        (L^-1)_jj = W_j^T,   (L^-1)_ij = -W_i^T-applied: -L_ii^-1 sum_{k=j}^{i-1} L_ik (L^-1)_kj
    //      with L_ik (L^-1)_kj = tile_xty(U_ki, (L^-1)_kj).  Column j only reads storage rows >= j and, of row j, the
    //      tile it is about to overwrite: ascending j, ascending i is safe.
    //   2. K^-1 row I into storage row I: (K^-1)_IJ = sum_{M >= J} (L^-1)_MI^T (L^-1)_MJ, I <= J; row I is the last
    //      reader of block column I of L^-1.
    // 40 tile products for 4 x 4 tiles.  Why the explicit inverse when two substitutions per product would do (and are
    // more accurate): the refinement loop's exit test (Solver.cpp:30-39) compares the ROUNDING NOISE of x = Kinv * Ab
    // with 1e-10.  Through substitutions that noise is ~1e-14 and the loop always leaves after one body; through
    // the explicit inverse -- entries ~1e7 on a singular P -- it is ~1e-11 ... 1e-10 and the reference takes three
    // bodies on 10-95 % of such problems (tools/probe_illcond.py).  Same formulas as the reference => same coin.
    // trbuf: 16 x 17 doubles of wave-private LDS for the tile transposes (tile_transpose_lds: an FP64 MFMA keeps the
    // FP64 vector ALUs busy for 69 cycles on this chip, an LDS round trip does not)
    DQQ_D void invert_in_place(int lane, double* __restrict__ trbuf)
    {
        const v4d zero = {0.0, 0.0, 0.0, 0.0};
        static_for<0, NT>([&](auto jc) __attribute__((always_inline)) {
            constexpr int J = decltype(jc)::value;
            const v4d Vj = tile_transpose_lds(U[J][J], trbuf, lane); // (L^-1)_jj; U[J][J] (= W_j) stays until the end of column j
            static_for<J + 1, NT>([&](auto ic) __attribute__((always_inline)) {
                constexpr int I = decltype(ic)::value;
                v4d S = tile_xty(zero, U[J][I], Vj);             // k = j: U_ji^T (L^-1)_jj
                static_for<J + 1, I>([&](auto kc) __attribute__((always_inline)) {
                    constexpr int K = decltype(kc)::value;
                    S = tile_xty(S, U[K][I], U[J][K]);           // U_ki^T (L^-1)_kj   (slot [J][K] already holds (L^-1)_kj)
                });
                U[J][I] = -tile_xty(zero, U[I][I], S);           // -(W_i)^T S = -L_ii^-1 S
            });
            U[J][J] = Vj;
        });
        static_for<0, NT>([&](auto ic) __attribute__((always_inline)) {
            constexpr int I = decltype(ic)::value;
            v4d row[NT];
            static_for<I, NT>([&](auto jc) __attribute__((always_inline)) {
                constexpr int J = decltype(jc)::value;
                v4d acc = zero;
                static_for<J, NT>([&](auto mc) __attribute__((always_inline)) {
                    constexpr int M = decltype(mc)::value;
                    acc = tile_xty(acc, U[I][M], U[J][M]);       // (L^-1)_MI^T (L^-1)_MJ
                });
                row[J] = acc;
            });
            static_for<I, NT>([&](auto jc) __attribute__((always_inline)) { U[I][decltype(jc)::value] = row[decltype(jc)::value]; });
        });
    }

    // x = K^-1 b by two block substitutions with the factor (after factor(), NOT after invert_in_place());
    // b, x one element per lane (lanes >= 16 NT: ignored / zero)
    DQQ_D double solve(double b, int lane) const
    {
        const int g = lane >> 4, n = lane & 15;
        const int lsrc = (lane & 48) + 4 * (n & 3) + g; // lane (g, r) <- entry 4r + g of a vector replicated per row
        const v4d I16 = identity_tile(lane);
        double yrep[NT], y0[NT], x0[NT];
        // forward: U^T y = b, y_J = L_JJ^-1 (b_J - sum_{K<J} U_KJ^T y_K)
#pragma unroll
        for (int J = 0; J < NT; ++J) {
            double z = lane_gather(b, 16 * J + n);
            if (J > 0) {
                double acc = 0.0;
#pragma unroll
                for (int K = 0; K < J; ++K) tile_dot4<0>(acc, U[K][J], y0[K]);
                z -= allreduce_rows(acc);
            }
            const double z0 = dpp_source(lane_gather(z, lsrc));
            double acc2 = 0.0;
            tile_dot4<0>(acc2, U[J][J], z0);              // W_J^T z = L_JJ^-1 z
            yrep[J] = allreduce_rows(acc2);
            y0[J] = dpp_source(lane_gather(yrep[J], lsrc));
        }
        // backward: U x = y, x_K = L_KK^-T (y_K - sum_{J>K} U_KJ x_J); U_KJ x_J = (U_KJ^T)^T x_J
        double x = 0.0;
#pragma unroll
        for (int K = NT - 1; K >= 0; --K) {
            double w = yrep[K];
            if (K < NT - 1) {
                double acc = 0.0;
#pragma unroll
                for (int J = K + 1; J < NT; ++J) tile_dot4<0>(acc, tile_transpose(U[K][J], I16), x0[J]);
                w -= allreduce_rows(acc);
            }
            const double w0 = dpp_source(lane_gather(w, lsrc));
            double acc2 = 0.0;
            tile_dot4<0>(acc2, tile_transpose(U[K][K], I16), w0); // (L_KK^-1)^T w
            const double xr = allreduce_rows(acc2);
            if (K > 0) x0[K] = dpp_source(lane_gather(xr, lsrc));
            x = (g == K) ? xr : x;
        }
        return x;
    }
};

// a[0..3] of lane (g,n) = partial sums of y[16 t + n] over the entries = g (mod 4) of the contraction: reduce over the
// four 16-lane rows, scattering t = row (the tail of WaveTile::matvec); result: y[l] in lane l
DQQ_D double reduce_scatter4(double a0, double a1, double a2, double a3)
{
    double p, q2, s02, s13, e, o;
    swap32(a0, a2, p, q2);
    s02 = p + q2;
    swap32(a1, a3, p, q2);
    s13 = p + q2;
    swap16(s02, s13, e, o);
    return e + o;
}

// y = S x for the symmetric S whose UPPER tiles are in Su (x, y one element per lane; xsrc = 4 (lane & 15) + (lane >> 4)):
// the lower tiles are transposes of the upper ones, made through LDS (trbuf: 16 x 17 doubles, wave-private)
template <int NT>
DQQ_D double sym_upper_matvec(const v4d (&Su)[NT][NT], double x, int xsrc, int lane, double* __restrict__ trbuf)
{
    const double x0 = dpp_source(lane_gather(x, xsrc)); // lane (g, n') <- x[4 n' + g]
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    static_for<0, NT>([&](auto tjc) __attribute__((always_inline)) {
        constexpr int TJ = decltype(tjc)::value;
        static_for<0, TJ + 1>([&](auto tic) __attribute__((always_inline)) {
            constexpr int TI = decltype(tic)::value;
            tile_dot4<4 * TI>(a[TJ], Su[TI][TJ], x0);                     // rows of block TI into block TJ
            if constexpr (TI < TJ) tile_dot4<4 * TJ>(a[TI], tile_transpose_lds(Su[TI][TJ], trbuf, lane), x0);
        });
    });
    double p, q2, s02, s13, e, o;
    swap32(a[0], a[2], p, q2);
    s02 = p + q2;
    swap32(a[1], a[3], p, q2);
    s13 = p + q2;
    swap16(s02, s13, e, o);
    return e + o;
}

} // namespace dqq
