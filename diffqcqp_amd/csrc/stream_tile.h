// stream_tile.h -- coalesced streaming of a wave's tile of P in the (B,N,N) layout with the
// on-the-fly diagonality check, shared by the forward and backward fast paths.
#pragma once

#include "common.h"

namespace dqq {

// Streams NCH chunks of 128 doubles (16 B per lane, 1 KiB per wave instruction) of a tile of P,
// writes the diagonal entries to sd[problem*N + row] and returns a non-zero word if any
// off-diagonal entry is not +-0.  GUARD: the tile is ragged (last tile of the batch).
// Large tiles (N >= 32: 64+ KiB) stop streaming at the first group of chunks that shows a non-zero off-diagonal
// -- the tile is not going to take the fast path, and a dense batch would otherwise be read twice in full.
template <int N, int NCH, bool GUARD>
static DQQ_D unsigned stream_tile_diag(const double* __restrict__ Pw, int limit, double* sd, int lane)
{
    unsigned nz = 0;
    constexpr int U = NCH < 16 ? (NCH < 8 ? NCH : 8) : 16;
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
            const int row = f / N; // == problem*N + r
            const int r = row % N, c = f % N; // c is even; (r,c) and (r,c+1) are this lane's entries
            const unsigned b0 = nonzero_bits(v[j].x), b1 = nonzero_bits(v[j].y);
            if (c == r) { sd[row] = v[j].x; nz |= b1; }
            else if (c + 1 == r) { sd[row] = v[j].y; nz |= b0; }
            else nz |= b0 | b1;
        }
        if (NCH > 32 && __any(nz != 0)) break; // wave-uniform
    }
    return nz;
}

} // namespace dqq
