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
#if defined(DQQ_STREAM_U) // developer experiment: loads in flight per group of the large tiles
    constexpr int U = NCH >= 32 ? DQQ_STREAM_U : (NCH < 16 ? (NCH < 8 ? NCH : 8) : 16);
#else
    constexpr int U = NCH < 16 ? (NCH < 8 ? NCH : 8) : 16;
#endif
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
