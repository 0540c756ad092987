// wave_tile.h -- one wave64 owns one symmetric matrix of up to 64 x 64 entirely in registers (gfx950).
//
// Storage ("tile layout"): the matrix is cut into 4 x 4 tiles of 16 x 16; tile (ti,tj) is a v4d per lane in
// the accumulator layout of v_mfma_f64_16x16x4_f64:
//     lane l = (g = l >> 4, n = l & 15), register r:   G[ti][tj][r] = S[16 ti + 4 r + g][16 tj + n]
// 64 doubles (128 VGPRs) per lane, no LDS.  Two facts make that layout self-sufficient:
//   * a tile in that layout is DIRECTLY an operand of the matrix core: register s supplies the k-slice
//     {4s..4s+3} of the contraction index = the tile's ROW index, for the A and for the B port alike.  Hence
//     every product of the form X^T Y (contraction over the rows of both) runs register to register;
//     with symmetric matrices (S[I][K] = S[K][I]^T) that is all a block Gauss-Jordan sweep needs.
//   * by symmetry lane (g,n) also holds row (16 tj + n) of S at the columns {16 ti + 4 r + g}: a mat-vec is
//     64 v_fmac_f64 with a DPP row_newbcast operand (lane n' = 4 ti + r of each 16-lane row supplies
//     x[16 ti + 4 r + g]) followed by a 3-step reduce-scatter over the four 16-lane rows
//     (v_permlane32_swap / v_permlane16_swap), which leaves y[l] in lane l.
// Vectors live one element per lane (lane l = element l).
#pragma once

#include "common.h"

namespace dqq {

typedef double v4d __attribute__((ext_vector_type(4)));

// value held by lane `src` (wave-uniform) of the calling wave
DQQ_D double lane_bcast(double v, int src)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}


// one tile-row T[0..3] streamed from memory: acc[tj] += T[tj][R] * (lane BC of the 16-lane row of x0)
#define DQQ_FMAC_BCAST_TROW(T, R, BC)                                                                          \
    asm("v_fmac_f64_dpp %0, %4, %5 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"                              \
        "v_fmac_f64_dpp %1, %4, %6 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"                              \
        "v_fmac_f64_dpp %2, %4, %7 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"                              \
        "v_fmac_f64_dpp %3, %4, %8 row_newbcast:%9 row_mask:0xf bank_mask:0xf"                                  \
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)                                                                \
        : "v"(x0), "v"(T[0][R]), "v"(T[1][R]), "v"(T[2][R]), "v"(T[3][R]), "n"(BC))

// {a[0..31] | b[0..31]} and {a[32..63] | b[32..63]}   (v_permlane32_swap on both halves of a double)
DQQ_D void swap32(double a, double b, double& lo_halves, double& hi_halves)
{
    const auto l = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
    const auto h = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
    lo_halves = __hiloint2double(h[0], l[0]);
    hi_halves = __hiloint2double(h[1], l[1]);
}

// {a.row0 | b.row0 | a.row2 | b.row2} and {a.row1 | b.row1 | a.row3 | b.row3}   (v_permlane16_swap)
DQQ_D void swap16(double a, double b, double& even_rows, double& odd_rows)
{
    const auto l = __builtin_amdgcn_permlane16_swap(__double2loint(a), __double2loint(b), false, false);
    const auto h = __builtin_amdgcn_permlane16_swap(__double2hiint(a), __double2hiint(b), false, false);
    even_rows = __hiloint2double(h[0], l[0]);
    odd_rows = __hiloint2double(h[1], l[1]);
}

// A DPP source written by a VALU instruction needs two wait states before the DPP read.  The mat-vec's source
// comes from ds_bpermute (no hazard), but the compiler may copy it with a v_mov right before the first
// v_fmac_f64_dpp of an asm block it cannot see into: this pins the value and pays the two states once.
DQQ_D double dpp_source(double v)
{
    asm volatile("s_nop 1" : "+v"(v));
    return v;
}

// value of v held by lane `src_lane` (per-lane index), through the LDS crossbar (no LDS memory)
DQQ_D double lane_gather(double v, int src_lane)
{
    const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2hiint(v));
    return __hiloint2double(hi, lo);
}

DQQ_D double wave_sum64(double v)
{
    const double m = LaneGroup<16>::sum(v);
    return (lane_bcast(m, 0) + lane_bcast(m, 16)) + (lane_bcast(m, 32) + lane_bcast(m, 48));
}

// wave-uniform double held by lane `src` (compile-time) -> SGPR pair
template <int SRC>
DQQ_D double lane_to_sgpr(double v)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), SRC), __builtin_amdgcn_readlane(__double2loint(v), SRC));
}

// max over the wave of |a| and of |b|: one butterfly for both, results wave-uniform (SGPRs)
DQQ_D void wave_max2_abs(double a, double b, double& ma, double& mb)
{
    double lo, hi, e, o;
    swap32(a, b, lo, hi); // lo = {a[0..31] | b[0..31]}, hi = {a[32..63] | b[32..63]}
    const double m = LaneGroup<16>::max(max_abs2(lo, hi)); // rows 0,1: max |a| over rows {r, r+2}; rows 2,3: |b|
    swap16(m, m, e, o);   // e = m of rows (0,0,2,2), o = m of rows (1,1,3,3)
    const double mm = max_raw(e, o); // lanes 0..31: max |a|, lanes 32..63: max |b|
    ma = lane_to_sgpr<0>(mm);
    mb = lane_to_sgpr<32>(mm);
}

// acc += (lane BC of this lane's 16-lane row of x0) * m
template <int BC>
DQQ_D void fmac_bcast(double& acc, double x0, double m)
{
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x0), "v"(m), "n"(BC));
}

// 16 v_fmac_f64_dpp in ONE asm statement (tile row TI of a 4 x 4 matrix): between separate asm statements the
// compiler's hazard recogniser, which cannot see the DPP operand inside, pads every hand-off with an s_nop
#define DQQ_MATVEC_TROW16(G, TI)                                                                                \
    asm("v_fmac_f64_dpp %0, %4, %5 row_newbcast:%21 row_mask:0xf bank_mask:0xf\n\t"                             \
        "v_fmac_f64_dpp %1, %4, %6 row_newbcast:%21 row_mask:0xf bank_mask:0xf\n\t"                             \
        "v_fmac_f64_dpp %2, %4, %7 row_newbcast:%21 row_mask:0xf bank_mask:0xf\n\t"                             \
        "v_fmac_f64_dpp %3, %4, %8 row_newbcast:%21 row_mask:0xf bank_mask:0xf\n\t"                             \
        "v_fmac_f64_dpp %0, %4, %9 row_newbcast:%22 row_mask:0xf bank_mask:0xf\n\t"                             \
        "v_fmac_f64_dpp %1, %4, %10 row_newbcast:%22 row_mask:0xf bank_mask:0xf\n\t"                            \
        "v_fmac_f64_dpp %2, %4, %11 row_newbcast:%22 row_mask:0xf bank_mask:0xf\n\t"                            \
        "v_fmac_f64_dpp %3, %4, %12 row_newbcast:%22 row_mask:0xf bank_mask:0xf\n\t"                            \
        "v_fmac_f64_dpp %0, %4, %13 row_newbcast:%23 row_mask:0xf bank_mask:0xf\n\t"                            \
        "v_fmac_f64_dpp %1, %4, %14 row_newbcast:%23 row_mask:0xf bank_mask:0xf\n\t"                            \
        "v_fmac_f64_dpp %2, %4, %15 row_newbcast:%23 row_mask:0xf bank_mask:0xf\n\t"                            \
        "v_fmac_f64_dpp %3, %4, %16 row_newbcast:%23 row_mask:0xf bank_mask:0xf\n\t"                            \
        "v_fmac_f64_dpp %0, %4, %17 row_newbcast:%24 row_mask:0xf bank_mask:0xf\n\t"                            \
        "v_fmac_f64_dpp %1, %4, %18 row_newbcast:%24 row_mask:0xf bank_mask:0xf\n\t"                            \
        "v_fmac_f64_dpp %2, %4, %19 row_newbcast:%24 row_mask:0xf bank_mask:0xf\n\t"                            \
        "v_fmac_f64_dpp %3, %4, %20 row_newbcast:%24 row_mask:0xf bank_mask:0xf"                                 \
        : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3])                                                        \
        : "v"(x0), "v"(G[TI][0][0]), "v"(G[TI][1][0]), "v"(G[TI][2][0]), "v"(G[TI][3][0]), "v"(G[TI][0][1]),      \
          "v"(G[TI][1][1]), "v"(G[TI][2][1]), "v"(G[TI][3][1]), "v"(G[TI][0][2]), "v"(G[TI][1][2]),              \
          "v"(G[TI][2][2]), "v"(G[TI][3][2]), "v"(G[TI][0][3]), "v"(G[TI][1][3]), "v"(G[TI][2][3]),              \
          "v"(G[TI][3][3]), "n"(4 * TI + 0), "n"(4 * TI + 1), "n"(4 * TI + 2), "n"(4 * TI + 3))

template <int NT, int I>
struct MatvecRows { // rows I, I+1, ... of the 4 NT broadcast steps (BC = I = 4 ti + r), unrolled at compile time
    static DQQ_D void run(const v4d (&G)[NT][NT], double x0, double (&acc)[4])
    {
#pragma unroll
        for (int tj = 0; tj < NT; ++tj) fmac_bcast<I>(acc[tj], x0, G[I / 4][tj][I % 4]);
        if constexpr (I + 1 < 4 * NT) MatvecRows<NT, I + 1>::run(G, x0, acc);
    }
};

// The matrix of one problem: (16 NT) x (16 NT), NT x NT tiles.  NT = 4 is the 64 x 64 case; smaller NT leave the
// lanes >= 16 NT of every vector idle (they hold zeros).
template <int NT>
struct WaveTile {
    v4d G[NT][NT];

    // y = S x for the symmetric S held in G (or y = A x when G holds the tile layout of A^T); x, y one
    // element per lane.  xsrc = 4 (lane & 15) + (lane >> 4).
    DQQ_D double matvec(double x, int xsrc) const
    {
        const double x0 = dpp_source(lane_gather(x, xsrc)); // lane (g, n') <- x[4 n' + g]
        double a[4] = {0.0, 0.0, 0.0, 0.0};
        if constexpr (NT == 4) {
            DQQ_MATVEC_TROW16(G, 0);
            DQQ_MATVEC_TROW16(G, 1);
            DQQ_MATVEC_TROW16(G, 2);
            DQQ_MATVEC_TROW16(G, 3);
        } else {
            MatvecRows<NT, 0>::run(G, x0, a);
        }
        // a[tj] of lane (g,n) = partial sum of y[16 tj + n] over the columns = g (mod 4): reduce over g,
        // scattering tj = g (accumulators beyond NT are zero)
        double p, q2, s02, s13, e, o;
        swap32(a[0], a[2], p, q2);
        s02 = p + q2; // rows 0,1: a0 summed over {g, g+2}; rows 2,3: a2
        swap32(a[1], a[3], p, q2);
        s13 = p + q2;
        swap16(s02, s13, e, o);
        return e + o;
    }
};
using WaveTile64 = WaveTile<4>;

// The same mat-vec with the matrix streamed one tile-row at a time (T[tj] = tiles (TK, tj) of the layout):
//     MatvecStream mv; mv.begin(x, xsrc); for TK: mv.add_row<TK>(T); y = mv.finish();
struct MatvecStream {
    double x0, a0, a1, a2, a3;
    DQQ_D void begin(double x, int xsrc)
    {
        x0 = dpp_source(lane_gather(x, xsrc));
        a0 = a1 = a2 = a3 = 0.0;
    }
    template <int TK>
    DQQ_D void add_row(const v4d (&T)[4])
    {
        DQQ_FMAC_BCAST_TROW(T, 0, 4 * TK + 0);
        DQQ_FMAC_BCAST_TROW(T, 1, 4 * TK + 1);
        DQQ_FMAC_BCAST_TROW(T, 2, 4 * TK + 2);
        DQQ_FMAC_BCAST_TROW(T, 3, 4 * TK + 3);
    }
    DQQ_D double finish() const
    {
        double p, q2, s02, s13, e, o;
        swap32(a0, a2, p, q2);
        s02 = p + q2;
        swap32(a1, a3, p, q2);
        s13 = p + q2;
        swap16(s02, s13, e, o);
        return e + o;
    }
};

// ------------------------------------------------------------------------------------------------------
// 16 x 16 symmetric positive definite block, inverse by 16 symmetric sweeps, in registers.
// In: T = the block in tile layout (T[r] of lane (g,n) = B[4r+g][n]).  Out: its inverse in tile layout.
// Each 16-lane row of the wave works on the same block redundantly: lane (.,i) holds row i of the block in
// 16 registers, a pivot column is handed to the other rows of the block as a DPP row_newbcast operand.
// Sweep k on the stored rows a_i (true row = s_i * stored row, s_i = 1 until row i has been the pivot):
//     d = a_kk, t_i = a_ik / d, b_i = -s_i t_i (= minus the true a_ik / d = a_ki / d by symmetry)
//     rows i != k:  a_ij += a_ik b_j  (j != k),  a_ik = t_i
//     row k: unchanged except a_kk = -1, s_k = 1/d      (deferring the scaling of the pivot row keeps the
//     update of all rows the same instruction; scaling it in place would cost a second pass per sweep)
// After 16 sweeps s_i * a_i = -B^-1.   bad: a pivot was not positive.
#define DQQ_SWEEP5(J0, J1, J2, J3, J4)                                                                         \
    asm("s_nop 1\n\t"                                                                                           \
        "v_fmac_f64_dpp %0, %5, %6 row_newbcast:" #J0 " row_mask:0xf bank_mask:0xf\n\t"                         \
        "v_fmac_f64_dpp %1, %5, %6 row_newbcast:" #J1 " row_mask:0xf bank_mask:0xf\n\t"                         \
        "v_fmac_f64_dpp %2, %5, %6 row_newbcast:" #J2 " row_mask:0xf bank_mask:0xf\n\t"                         \
        "v_fmac_f64_dpp %3, %5, %6 row_newbcast:" #J3 " row_mask:0xf bank_mask:0xf\n\t"                         \
        "v_fmac_f64_dpp %4, %5, %6 row_newbcast:" #J4 " row_mask:0xf bank_mask:0xf"                              \
        : "+v"(a[J0]), "+v"(a[J1]), "+v"(a[J2]), "+v"(a[J3]), "+v"(a[J4])                                       \
        : "v"(b), "v"(nc))
#define DQQ_SWEEP15(K, J0, J1, J2, J3, J4, J5, J6, J7, J8, J9, J10, J11, J12, J13, J14)                        \
    do {                                                                                                        \
        DQQ_SWEEP5(J0, J1, J2, J3, J4);                                                                         \
        DQQ_SWEEP5(J5, J6, J7, J8, J9);                                                                         \
        DQQ_SWEEP5(J10, J11, J12, J13, J14);                                                                    \
    } while (0)

template <int K>
DQQ_D void sweep16_step(double (&a)[16], double& ns, int n, bool& bad)
{
    // ns = MINUS the row scale s: b carries the sign, so the update below is a plain a_ij += b_j * a_ik
    const double d = __builtin_amdgcn_update_dpp(0.0, a[K], 0x150 + K, 0xf, 0xf, true); // a_kk of this 16-lane row
    bad = bad | (__ballot(!(d > 0.0)) != 0); // decided here (scalar, no branch), or all 64 pivots stay alive for a late test
    const double rd = fast_rcp(d);
    const double c = a[K];
    const double t = c * rd;
    const double b = t * ns;                 // - true a_ik / d
    double nc = c;
    if (n == K) {                            // the pivot row (an exec-mask region, no per-lane selects): keeps its
        nc = 0.0;                            // entries, a_kk = -1, and from now on carries the scale 1/d
        ns = -rd;
    }
    a[K] = (n == K) ? -1.0 : t;
    // the 15 columns j != K, the next pivot's column first
    if constexpr (K == 0) DQQ_SWEEP15(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
    if constexpr (K == 1) DQQ_SWEEP15(1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 0);
    if constexpr (K == 2) DQQ_SWEEP15(2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 0, 1);
    if constexpr (K == 3) DQQ_SWEEP15(3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 0, 1, 2);
    if constexpr (K == 4) DQQ_SWEEP15(4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 0, 1, 2, 3);
    if constexpr (K == 5) DQQ_SWEEP15(5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 0, 1, 2, 3, 4);
    if constexpr (K == 6) DQQ_SWEEP15(6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 0, 1, 2, 3, 4, 5);
    if constexpr (K == 7) DQQ_SWEEP15(7, 8, 9, 10, 11, 12, 13, 14, 15, 0, 1, 2, 3, 4, 5, 6);
    if constexpr (K == 8) DQQ_SWEEP15(8, 9, 10, 11, 12, 13, 14, 15, 0, 1, 2, 3, 4, 5, 6, 7);
    if constexpr (K == 9) DQQ_SWEEP15(9, 10, 11, 12, 13, 14, 15, 0, 1, 2, 3, 4, 5, 6, 7, 8);
    if constexpr (K == 10) DQQ_SWEEP15(10, 11, 12, 13, 14, 15, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9);
    if constexpr (K == 11) DQQ_SWEEP15(11, 12, 13, 14, 15, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10);
    if constexpr (K == 12) DQQ_SWEEP15(12, 13, 14, 15, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11);
    if constexpr (K == 13) DQQ_SWEEP15(13, 14, 15, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12);
    if constexpr (K == 14) DQQ_SWEEP15(14, 15, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13);
    if constexpr (K == 15) DQQ_SWEEP15(15, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14);
}

DQQ_D v4d diag16_inverse(const v4d& T, int lane, bool& bad)
{
    const int n = lane & 15;
    double a[16];
    // all-gather over the four 16-lane rows: a[4r + g'] = T[r] of lane (g', n) = B[n][4r + g'] (symmetry)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        double h0, h1;
        swap32(T[r], T[r], h0, h1); // h0 = T[r] of rows (0,1,0,1), h1 = T[r] of rows (2,3,2,3)
        swap16(h0, h0, a[4 * r + 0], a[4 * r + 1]);
        swap16(h1, h1, a[4 * r + 2], a[4 * r + 3]);
    }
    double s = -1.0; // minus the row scale
    sweep16_step<0>(a, s, n, bad);  sweep16_step<1>(a, s, n, bad);  sweep16_step<2>(a, s, n, bad);
    sweep16_step<3>(a, s, n, bad);  sweep16_step<4>(a, s, n, bad);  sweep16_step<5>(a, s, n, bad);
    sweep16_step<6>(a, s, n, bad);  sweep16_step<7>(a, s, n, bad);  sweep16_step<8>(a, s, n, bad);
    sweep16_step<9>(a, s, n, bad);  sweep16_step<10>(a, s, n, bad); sweep16_step<11>(a, s, n, bad);
    sweep16_step<12>(a, s, n, bad); sweep16_step<13>(a, s, n, bad); sweep16_step<14>(a, s, n, bad);
    sweep16_step<15>(a, s, n, bad);
    // back to tile layout: D[r] of lane (g,n) = Binv[4r+g][n] = Binv[n][4r+g] = -s * a[4r+g].  All four 16-lane
    // rows hold the same 16 values, so the choice by g is made by the swaps themselves (no per-lane select:
    // `g & 1 ? a[i+1] : a[i]` is turned into a dynamically indexed load by the compiler, which sends the whole
    // array to scratch)
    v4d D;
    const double ns = s; // already minus the scale
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        double lo, hi, pick, unused;
        swap16(a[4 * r + 0], a[4 * r + 1], lo, unused); // rows 0,2: a[4r], rows 1,3: a[4r+1]
        swap16(a[4 * r + 2], a[4 * r + 3], hi, unused); // rows 0,2: a[4r+2], rows 1,3: a[4r+3]
        swap32(lo, hi, pick, unused);                   // rows 0,1: lo, rows 2,3: hi
        D[r] = pick * ns;
    }
    return D;
}

// ------------------------------------------------------------------------------------------------------
// The same inverse with the block SPREAD over the wave instead of replicated four times (round 3).
// By symmetry the tile layout already is a row layout: T[c] of lane (g,n) = B[4c+g][n] = B[n][4c+g], i.e. lane
// (g,n) holds row n, columns 4c+g (c = 0..3) -- four registers instead of sixteen, no gather in, none out.
// Sweep k = 4 cb + gp on the stored rows (true row = s_n * stored row, as above):
//     col_n = a_nk (column k lives in register cb of the lanes of group gp; the four columns of a block cb are
//             all-gathered once per four sweeps and kept current by one extra FMA each),
//     d = col_k,  m_n = -col_n / d  (0 for the pivot row),
//     a_nj += a_kj m_n   for this lane's four columns j: a_kj is register c of lane k of the lane's own 16-lane row,
//             a DPP row_newbcast operand -- row k itself, not its mirror image, so the update never reads the scales;
//     column k: the lanes that hold it first take the pattern (-1 at row k, 0 elsewhere), and the same FMA then
//             leaves 0 + (-1) m_n = a_nk / d in rows n != k and the -1 in row k;   s_k = 1/d.
// 4 + (3,2,1,0) FMAs per sweep instead of 15, ~23 instead of ~37 instructions with the reciprocal and the masks.
#define DQQ_SPREAD_FMAC(REG, K)                                                                                 \
    asm("v_fmac_f64_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(REG) : "v"(m), "n"(K))

template <int CB, int GP>
DQQ_D void sweep16_spread_step(double (&a)[4], double (&e)[4], double& s, bool& bad)
{
    constexpr int K = 4 * CB + GP;
    constexpr unsigned long long kRowK = 0x0001000100010001ull << K;   // lanes n == K of the four 16-lane rows
    constexpr unsigned long long kGroup = 0xffffull << (16 * GP);      // the lanes that hold column K in a[CB]
    constexpr unsigned long long kGroupK = 1ull << (16 * GP + K);
    const double col = e[GP];
    const double d = __builtin_amdgcn_update_dpp(0.0, col, 0x150 + K, 0xf, 0xf, true); // a_kk in every lane
    bad = bad | (__ballot(!(d > 0.0)) != 0);
    const double rd = fast_rcp(d);
    double m = col * -rd;
    unsigned long long saved;
    asm volatile("s_mov_b64 %[sv], exec\n\t"
                 "s_and_b64 exec, %[mk], %[sv]\n\t" // (never more lanes than the caller had)
                 "v_mov_b64 %[m], 0\n\t"        // pivot row: stays as it is ...
                 "v_mov_b64 %[s], %[rd]\n\t"     // ... and carries the scale 1/d from now on
                 "s_and_b64 exec, %[mg], %[sv]\n\t"
                 "v_mov_b64 %[acb], 0\n\t"      // column K: the pattern the update turns into a_nk / d
                 "s_and_b64 exec, %[mgk], %[sv]\n\t"
                 "v_mov_b64 %[acb], -1.0\n\t"
                 "s_mov_b64 exec, %[sv]"
                 : [m] "+v"(m), [s] "+v"(s), [acb] "+v"(a[CB]), [sv] "=&s"(saved)
                 : [rd] "v"(rd), [mk] "s"(kRowK), [mg] "s"(kGroup), [mgk] "s"(kGroupK)
                 : "scc");
    // the later pivot columns of this block first (the next sweep reads one of them through DPP right away)
    if constexpr (GP < 1) DQQ_SPREAD_FMAC(e[1], K);
    if constexpr (GP < 2) DQQ_SPREAD_FMAC(e[2], K);
    if constexpr (GP < 3) DQQ_SPREAD_FMAC(e[3], K);
    // a[CB] last: the masked moves above wrote it, and a DPP read wants two wait states after a VALU write
    // (nothing pads the inside of an asm statement)
    DQQ_SPREAD_FMAC(a[(CB + 1) & 3], K);
    DQQ_SPREAD_FMAC(a[(CB + 2) & 3], K);
    DQQ_SPREAD_FMAC(a[(CB + 3) & 3], K);
    DQQ_SPREAD_FMAC(a[CB], K);
}

template <int CB>
DQQ_D void sweep16_spread_block(double (&a)[4], double& s, bool& bad)
{
    // all-gather of register CB over the four 16-lane rows: e[g'] = a[CB] of lane (g', n) = a_{n, 4 CB + g'}
    double e[4], h0, h1;
    asm volatile("s_nop 1" : "+v"(a[CB])); // written by an asm FMA a moment ago; the lane swaps read it
    swap32(a[CB], a[CB], h0, h1);
    swap16(h0, h0, e[0], e[1]);
    swap16(h1, h1, e[2], e[3]);
    sweep16_spread_step<CB, 0>(a, e, s, bad);
    sweep16_spread_step<CB, 1>(a, e, s, bad);
    sweep16_spread_step<CB, 2>(a, e, s, bad);
    sweep16_spread_step<CB, 3>(a, e, s, bad);
}

DQQ_D v4d diag16_inverse_spread(const v4d& T, int lane, bool& bad)
{
    (void)lane;
    double a[4] = {T[0], T[1], T[2], T[3]};
    double s = 1.0;
    sweep16_spread_block<0>(a, s, bad);
    sweep16_spread_block<1>(a, s, bad);
    sweep16_spread_block<2>(a, s, bad);
    sweep16_spread_block<3>(a, s, bad);
    // s_n a_n = -B^-1, and lane (g,n) register c is entry (n, 4c+g) = (4c+g, n) of it: tile layout as it stands
    v4d D;
#pragma unroll
    for (int c = 0; c < 4; ++c) D[c] = a[c] * -s;
    return D;
}

// out (+)= X^T Y for two tiles in tile layout (contraction over their row index), on the matrix core
DQQ_D v4d tile_xty(v4d acc, const v4d& X, const v4d& Y)
{
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(X[s], Y[s], acc, 0, 0, 0);
    return acc;
}

// The transpose of a tile through a wave-private LDS buffer of 16 x 17 doubles: written along its rows, read along
// its columns (the odd stride keeps both conflict-free).  8 LDS instructions instead of the 4 MFMAs of a product with
// the identity -- and on MI355X an FP64 MFMA occupies the FP64 vector ALUs for 69 cycles, an LDS
// round trip does not.
constexpr int kTrLd = 17;
DQQ_D v4d tile_transpose_lds(const v4d& T, double* __restrict__ buf, int lane)
{
    const int g = lane >> 4, n = lane & 15;
#pragma unroll
    for (int r = 0; r < 4; ++r) buf[(4 * r + g) * kTrLd + n] = T[r];
    wave_lds_fence();
    v4d R;
#pragma unroll
    for (int r = 0; r < 4; ++r) R[r] = buf[n * kTrLd + 4 * r + g];
    wave_lds_fence(); // the buffer is free for the next tile
    return R;
}

// In place: G (symmetric positive definite, tile layout) -> -G^-1, by four block sweeps:
//   D = G_KK^-1;  B_J = D G_KJ;  G_IJ -= G_KI^T B_J (I <= J, both != K);  G_JI = G_IJ^T;  G_KJ = B_J;  G_JK = B_J^T;
//   G_KK = -D
// The matrix stays symmetric, so only the upper half of the update and B_J are products on the matrix cores (3 + 6 per
// step for 4 x 4 tiles); their mirror images are transposes through LDS (`trbuf`, 16 x 17 doubles).  Round 2 computed
// all 15 tiles of a step as products (208 live MFMAs per sweep; now 144).
template <int NT, int K>
DQQ_D void block_sweep_step(v4d (&G)[NT][NT], int lane, bool& bad, double* __restrict__ trbuf)
{
    const v4d zero = {0.0, 0.0, 0.0, 0.0};
#ifdef DQQ_PIVOT_REPLICATED
    const v4d D = diag16_inverse(G[K][K], lane, bad);
#else
    const v4d D = diag16_inverse_spread(G[K][K], lane, bad);
#endif
    v4d Bt[NT];
#pragma unroll
    for (int J = 0; J < NT; ++J)
        if (J != K) Bt[J] = tile_xty(zero, D, G[K][J]);
#pragma unroll
    for (int I = 0; I < NT; ++I) {
        if (I == K) continue;
        const v4d nX = -G[K][I];
#pragma unroll
        for (int J = I; J < NT; ++J)
            if (J != K) G[I][J] = tile_xty(G[I][J], nX, Bt[J]);
    }
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int J = I + 1; J < NT; ++J)
            if (I != K && J != K) G[J][I] = tile_transpose_lds(G[I][J], trbuf, lane);
#pragma unroll
    for (int J = 0; J < NT; ++J)
        if (J != K) {
            G[K][J] = Bt[J];
            G[J][K] = tile_transpose_lds(Bt[J], trbuf, lane);
        }
    G[K][K] = -D;
}

template <int NT>
DQQ_D void block_sweep_inverse(v4d (&G)[NT][NT], int lane, bool& bad, double* __restrict__ trbuf)
{
    block_sweep_step<NT, 0>(G, lane, bad, trbuf);
    if constexpr (NT > 1) block_sweep_step<NT, 1>(G, lane, bad, trbuf);
    if constexpr (NT > 2) block_sweep_step<NT, 2>(G, lane, bad, trbuf);
    if constexpr (NT > 3) block_sweep_step<NT, 3>(G, lane, bad, trbuf);
}

} // namespace dqq
