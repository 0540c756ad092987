// stream_tile.h -- coalesced streaming of a wave's tile of P in the (B,N,N) layout with the
// on-the-fly diagonality check, shared by the forward and backward fast paths.
#pragma once

#include "common.h"

namespace dqq {

// One group of UU chunks of 128 doubles (16 B per lane, 1 KiB per wave instruction), all loads issued before the first
// is looked at: diagonal entries to sd[problem*N + row], the bit patterns of the off-diagonal entries ORed into nz.
template <int N, int UU, bool GUARD>
static DQQ_D void stream_group_diag(const double* __restrict__ Pw, int limit, double* sd, int lane, int k0, unsigned& nz)
{
    double2 v[UU];
#pragma unroll
    for (int j = 0; j < UU; ++j) {
        const int f = (k0 + j) * 128 + 2 * lane;
        if (GUARD) v[j] = f < limit ? *reinterpret_cast<const double2*>(Pw + f) : make_double2(0.0, 0.0);
        else v[j] = *reinterpret_cast<const double2*>(Pw + f);
    }
#pragma unroll
    for (int j = 0; j < UU; ++j) {
        const int f = (k0 + j) * 128 + 2 * lane;
        const int row = f / N; // == problem*N + r
        const int r = row % N, c = f % N; // c is even; (r,c) and (r,c+1) are this lane's entries
        const unsigned b0 = nonzero_bits(v[j].x), b1 = nonzero_bits(v[j].y);
        if (c == r) { sd[row] = v[j].x; nz |= b1; }
        else if (c + 1 == r) { sd[row] = v[j].y; nz |= b0; }
        else nz |= b0 | b1;
    }
}

// Streams NCH chunks of a tile of P, writes the diagonal entries to sd[problem*N + row] and returns a non-zero word if
// any off-diagonal entry is not +-0.  GUARD: the tile is ragged (last tile of the batch).
// Large tiles (N >= 32: 32+ KiB) stop streaming at the first group of chunks that shows a non-zero off-diagonal -- the
// tile is not going to take the fast path, and a dense batch would otherwise be read twice in full -- and their first
// group can be a PROBE of two chunks (the first rows of the tile's first problem): the verifying pass over a dense
// 65536 x 64 batch read 16 KiB per wave = 0.5 GB = 0.10 ms before it knew (round 3).  The forward probes (same box,
// B=262144 N=32 diagonal: 430 us with, 436 without: the probe also staggers the waves' bursts); the backward does
// not -- it rarely streams P at all (the forward's flags), and the extra code cost that kernel 3 % (404 -> 417 us).
template <int N, int NCH, bool GUARD, bool PROBE = false>
static DQQ_D unsigned stream_tile_diag(const double* __restrict__ Pw, int limit, double* sd, int lane)
{
    unsigned nz = 0;
    constexpr int U = NCH < 16 ? (NCH < 8 ? NCH : 8) : 16;   // loads in flight per group (N = 32 forward: 8 / 16 / 32 -> 475 / 453 / 488 us)
    if constexpr (NCH >= 32 && PROBE) {
        constexpr int U0 = 2;
        stream_group_diag<N, U0, GUARD>(Pw, limit, sd, lane, 0, nz);
        if (__any(nz != 0)) return nz; // wave-uniform
        stream_group_diag<N, U - U0, GUARD>(Pw, limit, sd, lane, U0, nz);
        for (int k0 = U; k0 < NCH; k0 += U) {
            if (__any(nz != 0)) break;
            stream_group_diag<N, U, GUARD>(Pw, limit, sd, lane, k0, nz);
        }
    } else {
        for (int k0 = 0; k0 < NCH; k0 += U) {
            stream_group_diag<N, U, GUARD>(Pw, limit, sd, lane, k0, nz);
            if (NCH > 32 && __any(nz != 0)) break;
        }
    }
    return nz;
}

// N = 8: the same stream (every chunk of the tile, no early stop), that ALSO says WHICH problems are not diagonal: a chunk
// of 128 doubles is two whole 8 x 8 matrices -- lanes 0..31 one, lanes 32..63 the next --, so one ballot per chunk
// classifies two problems, in scalar registers (round 5: the fused forward used to stream a non-diagonal tile a second
// time to find this out, tile_problem_flags below: 33.5 MB of a dense 65536 x 8 batch).  Bit p of the result: problem p of
// the tile has a non-zero off-diagonal entry.
template <int NCH, bool GUARD>
static DQQ_D unsigned long long stream_tile_diag_pmask8(const double* __restrict__ Pw, int limit, double* sd, int lane)
{
    constexpr int N = 8;
    static_assert(NCH <= 32, "at most 64 problems per tile");
    // (Round 6 tried the lane-constant form of this loop -- for N = 8 a chunk is two whole matrices, so WHICH of a lane's two
    // doubles is a diagonal entry and where it goes in sd does not depend on the chunk: one execution-mask region per group
    // instead of two per chunk, ~165 instead of ~435 instructions per tile.  The forwards read 0.35 us SLOWER, A/B on one box
    // (profiles/r07_ab_lean_stream_rejected.txt): this phase waits on HBM, its instructions are hidden, and the leaner
    // form waits for all sixteen loads before it looks at the first.)
    // every chunk's verdict is kept (NCH <= 16 registers, live only here, next to the 2 * U of the loads in flight) and only
    // OR-ed on the path a diagonal tile takes; the ballots that turn them into a per-problem mask run for a tile that HAS a
    // non-zero off-diagonal.  (With a ballot per chunk on every tile the headline's forwards read 1-1.5 % slower, A/B.)
    unsigned bs[NCH], nz = 0;
    constexpr int U = NCH < 16 ? NCH : 16;
#pragma unroll
    for (int k0 = 0; k0 < NCH; k0 += U) {
        double2 v[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int f = (k0 + j) * 128 + 2 * lane;
            if (GUARD) v[j] = f < limit ? *reinterpret_cast<const double2*>(Pw + f) : make_double2(0.0, 0.0);
            else v[j] = *reinterpret_cast<const double2*>(Pw + f);
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int f = (k0 + j) * 128 + 2 * lane;
            const int row = f / N, r = row % N, c = f % N;
            const unsigned b0 = nonzero_bits(v[j].x), b1 = nonzero_bits(v[j].y);
            unsigned b;
            if (c == r) { sd[row] = v[j].x; b = b1; }
            else if (c + 1 == r) { sd[row] = v[j].y; b = b0; }
            else b = b0 | b1;
            bs[k0 + j] = b;
            nz |= b;
        }
    }
    unsigned long long pmask = 0;
    if (__builtin_expect(__any(nz != 0), 0)) {
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const unsigned long long m = __ballot(bs[k] != 0);
            const unsigned long long two = ((m & 0xffffffffull) != 0 ? 1ull : 0ull) | ((m >> 32) != 0 ? 2ull : 0ull);
            pmask |= two << (2 * k);
        }
    }
    return pmask;
}

// N = 8, one lane per problem: the wave's tile of 64 matrices (32 KiB, contiguous) goes through LDS WHOLE -- coalesced 16-byte
// loads in, row stride N*N + 1 doubles (conflict-free when every lane then reads its own matrix) -- and is classified on the
// way as above.  Everything the lane needs of P afterwards (the diagonal, or the general solve's rows at every
// refactorisation) comes from there: P is read from HBM exactly once (round 5; the one-lane layout used to read it three
// times and more: stream, classification, the solve's own loads per refactorisation -- 240 MB per 65536 x 8 QCQP forward
// against 46 MB algorithmic).
template <bool GUARD>
static DQQ_D unsigned long long stage_tile_lane8(const double* __restrict__ Pw, int nvalid, double* st, int lane)
{
    constexpr int NN = 64, TS = NN + 1, CPP = NN / 2;   // doubles per matrix, LDS row stride, 16-byte chunks per matrix
    unsigned long long pmask = 0;
    const int last = nvalid * CPP - 1;
#pragma unroll 8
    for (int k = 0; k < CPP; ++k) {
        const int ch = k * 64 + lane;                   // chunk of the tile: matrix 2k + lane / 32, chunk lane % 32 of it
        const bool in = !GUARD || ch <= last;
        const double2 t = in ? *reinterpret_cast<const double2*>(Pw + 2 * (long)ch) : make_double2(0.0, 0.0);
        const int pp = ch / CPP, w = ch % CPP;
        st[pp * TS + 2 * w] = t.x;
        st[pp * TS + 2 * w + 1] = t.y;
        const int r = w / 4, c = 2 * (w % 4);           // entries (r, c) and (r, c + 1)
        const unsigned b0 = nonzero_bits(t.x), b1 = nonzero_bits(t.y);
        const unsigned b = (c == r) ? b1 : (c + 1 == r) ? b0 : (b0 | b1);
        const unsigned long long m = __ballot(b != 0);
        const unsigned long long two = ((m & 0xffffffffull) != 0 ? 1ull : 0ull) | ((m >> 32) != 0 ? 2ull : 0ull);
        pmask |= two << (2 * k);
    }
    return pmask;
}

// Which problems of a tile are not diagonal?  The tile streamed once more, coalesced (it has just been read: L2), each chunk's
// verdict dropped into flags[problem] (LDS, `nflags` ints, zeroed here).  Only tiles that stream_tile_diag found non-diagonal come
// here (fwd_diag.hip): reading a lane's own matrix instead -- N*N*8 bytes from its neighbour's, 64 cache lines per load
// instruction -- cost an all-dense 65536 x 8 forward on one lane per problem 4 us more.
template <int N, int NCH, int PPW>
static DQQ_D void tile_problem_flags(const double* __restrict__ Pw, int limit, int* flags, int lane)
{
    if (lane < PPW) flags[lane] = 0;
    wave_lds_fence();
    constexpr int U = NCH < 8 ? NCH : 8;
    for (int k0 = 0; k0 < NCH; k0 += U) {
        double2 v[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int f = (k0 + j) * 128 + 2 * lane;
            v[j] = f < limit ? *reinterpret_cast<const double2*>(Pw + f) : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int f = (k0 + j) * 128 + 2 * lane;
            const int row = f / N, r = row % N, c = f % N;
            const unsigned b0 = nonzero_bits(v[j].x), b1 = nonzero_bits(v[j].y);
            const unsigned b = (c == r) ? b1 : (c + 1 == r) ? b0 : (b0 | b1);
            if (b != 0) flags[row / N] = 1;
        }
    }
    wave_lds_fence();
}

} // namespace dqq
