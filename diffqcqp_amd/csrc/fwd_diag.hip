// fwd_diag.hip -- batched ADMM forward solve, diagonal-P fast path (gfx950).
//
// One launch replaces the Python batch loop of the reference (qcqp.py:29-31 for
// QPFn2.forward, :149-151 for QCQPFn2.forward).  A wave64 owns a TILE of
// PPW = 64/LPP consecutive problems; LPP adjacent lanes share a problem and
// each lane keeps E = N/LPP coordinates of every state vector in VGPRs for the
// whole solve (admm_core.h).  Nothing but P, q, (l_n, mu) is read from HBM and
// nothing but x (and the optional iteration counts) is written.
//
// P arrives in the drop-in layout (B,N,N).  The wave streams its tile of P with
// fully coalesced 16-byte loads (1 KiB per wave instruction), checks on the fly
// that every off-diagonal entry is exactly +-0, and drops the diagonal entries
// into LDS in [problem][coordinate] order, from where each lane picks up its E
// values.  A tile with any non-zero off-diagonal does not take the fast path:
// for N <= 16 the same wave solves its problems one by one with the general
// per-problem routine (dense_core.h); for larger N (where that routine's LDS
// footprint would cost the fast path its occupancy) the problem indices are
// appended to the fallback work-list that the general dense kernel (dense.hip)
// drains right after this launch.
#include <atomic>
#include <type_traits>

#include "admm_core.h"
#include "dense_core.h"
#include "group_dense.h"
#include "launch.h"
#include "stream_tile.h"

namespace dqq {

// N = 8, two lanes per problem (the bench shape), fused: four waves per SIMD (128 VGPRs) so that the forwards of two
// problem families -- or a forward and a backward -- are co-resident on a SIMD.  One lane per problem
// (round 4: the layout for batches that are mostly non-diagonal, below) is the opposite trade: a problem's whole matrix in
// its lane's registers, one wave per SIMD.
#define DQQ_FWD_DIAG_OCCUPANCY(KIND, N, LPP, FUSE) \
    __attribute__((amdgpu_waves_per_eu(((FUSE) && (N) == 8 && (LPP) == 2 && (KIND) < 2) ? 4 : 1, 8)))

// knob fwd_respread (tuning.h): once at most this many (0..16) of a wave's 32 problems are still iterating, they move onto
// twice the lanes (admm_core.h admm_fwd_diag_respread; N = 8, two lanes per problem, QP / QCQP).  0 = never.
// Results do not depend on it (bit-identical, tests/test_gpu_respread.py).
// knob fwd_respread2_from: ... from this iteration on only (the kernel's `respread2_at` argument carries both: at2 | from << 8).
// knob fwd_respread2: once at most this many (0..8) of the re-spread problems are still iterating, they move again,
// onto EIGHT lanes per problem (one coordinate per lane).  0 = never.  Bit-identical results.
int lane_defer_for(int kind); // fwd_lane_dense.hip: the general routines' deferred refactorisation (option lane_defer)
constexpr bool fwd_diag_respreads(int kind, int n, int lpp) { return kind < 2 && n == 8 && lpp == 2; }

template <int KIND, int N, int LPP, int WPB, bool FUSE>
__global__ __launch_bounds__(64 * WPB) DQQ_FWD_DIAG_OCCUPANCY(KIND, N, LPP, FUSE) void fwd_diag_kernel(const double* __restrict__ P,
                                                            const double* __restrict__ q,
                                                            const double* __restrict__ l_n,
                                                            const double* __restrict__ mu_c,
                                                            const double* __restrict__ v_sign, double* __restrict__ x,
                                                            long B, double eps, double mu_prox, int max_iter,
                                                            int adaptive, int layout, int* __restrict__ iters,
                                                            int* __restrict__ ws,
                                                            double* __restrict__ pdiag_out,
                                                            unsigned char* __restrict__ flags_out, int respread_at,
                                                            int respread2_at, int gdefer)
{
    constexpr int E = N / LPP;       // coordinates per lane
    constexpr int PPW = 64 / LPP;    // problems per wave tile
    constexpr int NCH = N * E / 2;   // 16-byte-per-lane chunks in a tile of P
    static_assert(E >= 2 && E % 2 == 0 && E * LPP == N, "bad N/LPP");
    // FUSE (small N, small batches): a non-diagonal tile is solved right here by the general per-problem
    // routine (its LDS scratch aliases the diagonal staging buffer); otherwise the tile is queued for
    // the dense kernel launched behind this one.
    // STAGE (N = 8 on ONE lane per problem, QP / QCQP -- the layout of batches that are mostly non-diagonal): the wave's
    // whole tile of P goes through LDS (stage_tile_lane8) and is read from HBM once.  33 KB per wave: affordable exactly
    // here, where the kernel runs one wave per SIMD anyway (DQQ_FWD_DIAG_OCCUPANCY).
    constexpr bool STAGE = FUSE && N == 8 && LPP == 1 && KIND < 2;
    constexpr int SMEM = STAGE ? 64 * (N * N + 1)
                               : ((FUSE && dense_fwd_lds_doubles(N) > 64 * E) ? dense_fwd_lds_doubles(N) : 64 * E);
    __shared__ __attribute__((aligned(16))) double s_diag[WPB][SMEM];
    constexpr bool AGG = !FUSE && WPB > 1; // queue non-diagonal tiles with ONE atomic per workgroup (see launch.h)
    __shared__ int s_cnt[2];
    // the segmented work-list's capacity, kWsSegCap(B) = B/32 + 512 slots per segment, rests on: one tile per wave, the
    // grid exactly ceil(tiles / WPB), at most 256 problems per workgroup
    static_assert(!worklist_segmented(N) || PPW * WPB <= 256, "segmented work-list: at most 256 problems per workgroup");
    // FUSE, N <= 8: a non-diagonal tile is solved by this wave, 64/LD problems at a time with LD = max(N/2, LPP) lanes
    // per problem (group_dense.h): two rows of the matrices per lane keep the general solve inside the register
    // budget of the diagonal arithmetic
    // (one lane per problem, QP / QCQP: the general solve on the caller's mapping too -- a lane per problem, LD == LPP)
    constexpr int LD = (N == 8 && LPP == 1 && KIND < 2) ? 1 : ((N / 2 > LPP) ? N / 2 : LPP);
    constexpr bool GD = FUSE && N <= 8 && LD <= 4 && group_dense_supported(KIND, N, LD);
    [[maybe_unused]] bool dense_tile = false; // GD: this tile holds non-diagonal problems
    [[maybe_unused]] unsigned long long pmask = 0;   // N = 8, GD: bit p = problem p of the tile is not diagonal (from the stream)

    // the wave index is wave-uniform: in an SGPR, the tile's position (`first`, `nvalid`, pointers) is scalar arithmetic and
    // costs no vector registers -- as a VGPR value the fused forward kept `first` and `nvalid` in SCRATCH (spilled and
    // reloaded in front of the stream of P, on the path every tile takes)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const long tile = (long)blockIdx.x * WPB + wave;
    const long first = tile * PPW;
    if (first >= B) return; // whole wave leaves before any workgroup barrier
    // this launch may fill the work-list: the words only its drain writes must be zero (launch.h, work-list hygiene) -- loaded
    // here by the first wave, looked at where the tile is queued
    [[maybe_unused]] WorklistIdle idle{0, 0, 0, 0};
    [[maybe_unused]] const bool prepares = !FUSE && tile == 0 && ws != nullptr && layout == DQQ_P_AUTO;
    if constexpr (!FUSE) {
        if (prepares) idle = worklist_prepare_begin(ws, lane);
    }
    const int nvalid = (B - first) < PPW ? (int)(B - first) : PPW;
    const int pl = lane / LPP;
    const bool valid = pl < nvalid;

    double p[E], qv[E], xv[E], rad[E / 2];
    constexpr int EB = (KIND >= 2) ? E : 1;
    double lo[EB], hi[EB], sg[EB]; // box kinds: l_n = l_min, mu_c = l_max, per coordinate

    // q and the constraint data first: their latency hides behind the stream of P (0.8 us of a 30 us launch)
    {
        const double* qq = q + first * N + lane * E;
#pragma unroll
        for (int e = 0; e < E; e += 2) {
            double2 t = valid ? *reinterpret_cast<const double2*>(qq + e) : make_double2(0.0, 0.0);
            qv[e] = t.x; qv[e + 1] = t.y;
        }
    }
    if (KIND == 1) {
        const long co = first * (N / 2) + lane * (E / 2);
#pragma unroll
        for (int c = 0; c < E / 2; ++c) rad[c] = valid ? l_n[co + c] * mu_c[co + c] : 1.0; // pybindings.cpp:57
    } else {
#pragma unroll
        for (int c = 0; c < E / 2; ++c) rad[c] = 0.0;
    }
    if constexpr (KIND >= 2) {
        const long bo = first * N + lane * E;
#pragma unroll
        for (int e = 0; e < E; e += 2) {
            const double2 a = valid ? *reinterpret_cast<const double2*>(l_n + bo + e) : make_double2(0.0, 0.0);
            const double2 b = valid ? *reinterpret_cast<const double2*>(mu_c + bo + e) : make_double2(0.0, 0.0);
            lo[e] = a.x; lo[e + 1] = a.y;
            hi[e] = b.x; hi[e + 1] = b.y;
            sg[e] = sg[e + 1] = 0.0;
            if (KIND == 3) {
                const double2 c = valid ? *reinterpret_cast<const double2*>(v_sign + bo + e) : make_double2(0.0, 0.0);
                sg[e] = (double)((c.x > 0) - (c.x < 0));       // cwiseSign, Solver.cpp:395
                sg[e + 1] = (double)((c.y > 0) - (c.y < 0));
            }
        }
    }

    if (layout == DQQ_P_DIAG) {
        const double* pp = P + first * N + lane * E;
#pragma unroll
        for (int e = 0; e < E; e += 2) {
            double2 t = valid ? *reinterpret_cast<const double2*>(pp + e) : make_double2(1.0, 1.0);
            p[e] = t.x; p[e + 1] = t.y;
        }
    } else if (GD && layout == DQQ_P_DENSE) {
        // the caller declares P general: no diagonal to look for, every tile takes the group solve
        dense_tile = true;
#pragma unroll
        for (int e = 0; e < E; ++e) p[e] = 1.0;
    } else {
        const double* Pw = P + first * (long)(N * N);
        const int limit = nvalid * N * N; // doubles of P that belong to this tile
        double* sd = s_diag[wave];
        bool tile_dense;   // wave-uniform
        if constexpr (STAGE) {
            pmask = (nvalid == PPW) ? stage_tile_lane8<false>(Pw, nvalid, sd, lane) : stage_tile_lane8<true>(Pw, nvalid, sd, lane);
            tile_dense = pmask != 0;
        } else if constexpr (GD && N == 8) {
            pmask = (nvalid == PPW) ? stream_tile_diag_pmask8<NCH, false>(Pw, limit, sd, lane)
                                    : stream_tile_diag_pmask8<NCH, true>(Pw, limit, sd, lane);
            tile_dense = pmask != 0;
        } else {
            const unsigned nz = (nvalid == PPW) ? stream_tile_diag<N, NCH, false, true>(Pw, limit, sd, lane)
                                                : stream_tile_diag<N, NCH, true, true>(Pw, limit, sd, lane);
            tile_dense = __any(nz != 0);
        }
        if constexpr (!GD) {   // (GD: problem by problem, below)
            if (tile_dense && flags_out != nullptr && valid && (lane % LPP) == 0) flags_out[first + pl] = 2; // seen, not diagonal
        }
        if constexpr (GD) {
            dense_tile = tile_dense;
        } else if constexpr (FUSE) {
            if (tile_dense) {
                for (int j = 0; j < nvalid; ++j)
                    dense_fwd_problem<KIND>(P, q, l_n, mu_c, v_sign, x, iters, first + j, N, eps, mu_prox, max_iter,
                                            adaptive, sd, lane);
                return;
            }
        } else {
            if (prepares) worklist_prepare_end(ws, lane, idle);
            const bool queued = worklist_push<AGG, worklist_segmented(N)>(ws, B, first, tile_dense ? nvalid : 0, lane, s_cnt);
            if (tile_dense) {
                if (!queued && valid) {   // (launch.h, work-list hygiene: the tile will not be solved -- say so in its outputs)
                    double* xx = x + first * N + lane * E;
#pragma unroll
                    for (int e = 0; e < E; ++e) xx[e] = __builtin_nan("");
                    if (iters != nullptr && (lane % LPP) == 0) iters[first + pl] = -1;   // (it comes from torch.empty)
                }
                return;
            }
        }
        wave_lds_fence();
        if constexpr (STAGE) {   // the lane's own matrix: its diagonal
#pragma unroll
            for (int e = 0; e < E; ++e) p[e] = valid ? sd[lane * (N * N + 1) + e * (N + 1)] : 1.0;
        } else {
#pragma unroll
            for (int e = 0; e < E; ++e) p[e] = valid ? sd[lane * E + e] : 1.0;
        }
    }

    int it = 0;
    // `mine`: this lane's problem is solved by the diagonal arithmetic below.  A tile with non-diagonal problems hands THOSE
    // to the general solve, problem by problem (round 4, late): which routine solves a problem -- and with it the last
    // bits of its x -- must not depend on its neighbours or on how many lanes a problem has (the layout follows the batch
    // size and the caller's hint flags, include/diffqcqp_hip.h; routed by whole wave tiles, the diagonal neighbours of a non-diagonal problem took
    // the general solve on two lanes per problem and the diagonal arithmetic on four: 1e-14 apart).
    bool mine = valid;
    [[maybe_unused]] unsigned long long dmask = 0;   // the lanes whose problem is non-diagonal (wave-uniform)
    if constexpr (GD) {
        // (unlikely: the blocks are placed behind the diagonal path's code -- the two-stream headline step reads 0.3 us less,
        // A/B of the builds at 100-step regions, five alternations: two ~30 KB forwards share the CUs' instruction caches.
        // The general solve itself comes LAST, when nothing of the diagonal arithmetic is live any more: between the
        // classification and the diagonal solve it kept that state alive across itself and the spills landed on the path
        // every tile takes, headline step 56 -> 69 us.)
        if (__builtin_expect(dense_tile, 0)) {
            bool densep = valid;   // declared dense: every problem
            if (layout != DQQ_P_DENSE) {
                if constexpr (N == 8) {   // classified by the stream itself (stream_tile.h)
                    densep = valid && ((pmask >> pl) & 1ull) != 0;
                } else {
                    int* pf = reinterpret_cast<int*>(s_diag[wave]);   // (the staged diagonals are not needed any more, below)
                    tile_problem_flags<N, NCH, PPW>(P + first * (long)(N * N), nvalid * N * N, pf, lane);
                    densep = valid && pf[pl] != 0;
                    wave_lds_fence();
                }
            }
            if (flags_out != nullptr && densep && (lane % LPP) == 0) flags_out[first + pl] = 2; // seen, not diagonal
            dmask = __ballot(densep);
            mine = valid && !densep;
            // the diagonal of the problems that stay: from P itself (the stream of a large tile stops at the first group of
            // chunks with a non-zero off-diagonal, stream_tile.h: the staging buffer may be incomplete)
            if constexpr (N == 8) {   // the stream of an 8 x 8 tile never stops early: the staged diagonals are complete
#pragma unroll
                for (int e = 0; e < E; ++e) p[e] = mine ? p[e] : 1.0;
            } else {
                const double* Pg = P + (first + pl) * (long)(N * N) + ((lane % LPP) * E) * (N + 1);
#pragma unroll
                for (int e = 0; e < E; ++e) p[e] = mine ? Pg[e * (N + 1)] : 1.0;
            }
        }
    }
    // N = 8 on two lanes per problem: the tail of the tile moves onto four lanes per problem (admm_core.h)
    constexpr bool RSP = fwd_diag_respreads(KIND, N, LPP);
    [[maybe_unused]] bool moved = false; // this lane's problem was finished (and stored) in the re-spread layout
    if (__builtin_expect(!GD || !dense_tile || __any(mine), 1)) {   // (a tile that is all non-diagonal: nothing for it)
        if constexpr (RSP)
            it = admm_fwd_diag_respread<KIND>(p, qv, rad, eps, mu_prox, max_iter, adaptive, mine, xv, respread_at,
                                              respread2_at, s_diag[wave], x + first * N, iters ? iters + first : nullptr, moved);
        else
            it = admm_fwd_diag<KIND, E, LaneGroup<LPP>>(p, qv, rad, N, eps, mu_prox, max_iter, adaptive, mine, xv, lo,
                                                        hi, sg);
    }

    if (mine) {
        if (!moved) {
            double* xx = x + first * N + lane * E;
#pragma unroll
            for (int e = 0; e < E; e += 2) *reinterpret_cast<double2*>(xx + e) = make_double2(xv[e], xv[e + 1]);
            if (iters != nullptr && (lane % LPP) == 0) iters[first + pl] = it;
        }
        // hand the verified diagonal to the backward of the same problems (it then skips the P stream)
        if (flags_out != nullptr && (lane % LPP) == 0) flags_out[first + pl] = 1;
        if (pdiag_out != nullptr) {
            double* pp = pdiag_out + first * N + lane * E;
#pragma unroll
            for (int e = 0; e < E; e += 2) *reinterpret_cast<double2*>(pp + e) = make_double2(p[e], p[e + 1]);
        }
    }
    if constexpr (GD) {
        // the tile's non-diagonal problems: the general solve on LD lanes per problem, 64 / LD problems per pass; it reads its
        // inputs and writes x / iters itself, in its own mapping
        if (__builtin_expect(dense_tile, 0)) {
            if constexpr (STAGE) {
                if (layout != DQQ_P_DENSE)
                    group_dense_tile<KIND, N, LD, PPW, true>(P, q, l_n, mu_c, v_sign, x, iters, first, nvalid, eps, mu_prox,
                                                             max_iter, adaptive, lane, gdefer, dmask, LPP, s_diag[wave]);
                else
                    group_dense_tile<KIND, N, LD, PPW>(P, q, l_n, mu_c, v_sign, x, iters, first, nvalid, eps, mu_prox, max_iter,
                                                       adaptive, lane, gdefer, dmask, LPP);
            } else {
                group_dense_tile<KIND, N, LD, PPW>(P, q, l_n, mu_c, v_sign, x, iters, first, nvalid, eps, mu_prox, max_iter,
                                                   adaptive, lane, gdefer, dmask, LPP);
            }
        }
    }
}

template <int KIND, int N, int LPP, int WPB, bool FUSE>
static hipError_t launch_one(const FwdArgs& a, hipStream_t s)
{
    constexpr int PPW = 64 / LPP;
    const long ntiles = (a.B + PPW - 1) / PPW;
    const long nblocks = (ntiles + WPB - 1) / WPB;
    if (nblocks == 0) return hipSuccess;
    return launch((fwd_diag_kernel<KIND, N, LPP, WPB, FUSE>), dim3((unsigned)nblocks), dim3(64 * WPB), 0, s, a.P, a.q,
                       a.l_n, a.mu, a.v, a.x, a.B, a.eps, a.mu_prox, a.max_iter, a.adaptive, a.layout, a.iters, a.ws,
                       a.pdiag_out, a.flags_out, std::min(16, std::max(0, knob_fwd_respread())),
                       // (the second move's threshold and the iteration from which it applies travel in one int: at2 | from << 8)
                       std::min(8, std::max(0, knob_fwd_respread2())) | (std::min(1 << 20, std::max(0, knob_fwd_respread2_from())) << 8),
                       lane_defer_for(KIND));
}

// What the shipped build instantiates is what its routing can reach (tuning.h: the knobs are constants there): four waves
// per workgroup always; the in-kernel general solve (FUSE) for N <= 8 only.  The developer build adds one wave per
// workgroup ("wpb" = 1) and the fused form for N = 16 ("fuse_fallback" = 1).
template <int KIND, int N, int LPP>
static hipError_t launch_wpb(const FwdArgs& a, int wpb, bool fuse, hipStream_t s)
{
    if constexpr (fwd_diag_fuses(N) && (kTuning || N <= 8)) {
        if (fuse) {
            if constexpr (kTuning) { if (wpb == 1) return launch_one<KIND, N, LPP, 1, true>(a, s); }
            return launch_one<KIND, N, LPP, 4, true>(a, s);
        }
    }
    if constexpr (kTuning) { if (wpb == 1) return launch_one<KIND, N, LPP, 1, false>(a, s); }
    return launch_one<KIND, N, LPP, 4, false>(a, s);
}

// Lanes-per-problem choices the kernel is instantiated for (E = N/LPP coordinates per lane,
// E even and <= 8), smallest first.
static const int* lpp_choices(int N, int& count)
{
    static const int c2[] = {1}, c4[] = {1, 2}, c8[] = {1, 2, 4}, c16[] = {2, 4, 8}, c32[] = {4, 8, 16},
                     c64[] = {8, 16, 32};
    switch (N) {
    case 2: count = 1; return c2;
    case 4: count = 2; return c4;
    case 8: count = 3; return c8;
    case 16: count = 3; return c16;
    case 32: count = 3; return c32;
    case 64: count = 3; return c64;
    default: count = 0; return nullptr;
    }
}

// Built-in choice, from sweeps on MI355X (tools/probe_lpp_sweep.py; us per forward launch, QP / QCQP):
//   N = 8    B = 49152: LPP 4 20.6 / 25.5, LPP 2 23.0 / 28.7;   65536: LPP 2 25.0 / 30.3, LPP 4 27.4 / 34.5;
//            131072: LPP 2 39.9 / 48.0, LPP 4 47.3 / 55.5;      262144: LPP 1 67 / 93, LPP 2 (fused kernel) 66 / 79;
//            1048576: LPP 1 238 / 326, LPP 2 (fused kernel) 222 / 264
//   N = 16   B = 32768: LPP 8 33.3 / 38.4, LPP 4 33.8 / 38.8;   65536: LPP 4 51.2 / 59.6, LPP 2 54.2 / 66.6;
//            262144: LPP 4 150.3 / 169.1, LPP 2 153.6 / 194.4, LPP 8 166.1 / 184.4
//   N = 4    LPP 2 below 131072 problems (32768: 12.6 / 15.6 against 16.4 / 22.2), LPP 1 from there on
// Fewer lanes per problem = fewer instructions per problem (the scalar rho / tau / stop logic and the pow()
// prologue are replicated on every lane of a problem), more lanes = more waves to hide the latency of a small
// batch behind; four coordinates per lane is the sweet spot of a batch that fills the chip (at N = 8 also because
// those tiles re-spread their tails, admm_fwd_diag_respread).
// N >= 32: a wave first streams 32+ KiB of P per problem, and the smaller its tile, the finer that stream
// interleaves with other waves' arithmetic -- the most lanes per problem win at every batch size measured
// (N=32 QP forward, LPP 4 / 8 / 16: B=32768 73 / 68 / 63 us, B=262144 478 / 458 / 442 us; QCQP B=32768
// 84 / 78 / 69 us; N=64 alike).
int fwd_diag_default_lpp(int N, long B, int kind)
{
    int count = 0;
    const int* c = lpp_choices(N, count);
    if (count == 0) return 0;
    if (N >= 32) return c[count - 1];
    if (N == 16) return B <= 40960 ? 8 : 4;
    if (N == 8) return B < 57344 ? 4 : 2;
    for (int i = 0; i < count; ++i)
        if (B * c[i] / 64 >= 2048) return c[i];
    return c[count - 1];
}

bool fwd_diag_supported(int N) { return fwd_diag_default_lpp(N, 1, 0) != 0; }


template <int KIND>
static bool launch_kind(const FwdArgs& a, int lpp, int wpb, bool fuse, hipStream_t s, hipError_t& err)
{
#define DQQ_CASE(NN, LL) \
    if (a.N == NN && lpp == LL) { err = launch_wpb<KIND, NN, LL>(a, wpb, fuse, s); return true; }
    // the lane layouts fwd_diag_default_lpp (+ the hint's one lane, + DQQ_P_DENSE's N / 2) can ask for ...
    DQQ_CASE(2, 1)
    DQQ_CASE(4, 1) DQQ_CASE(4, 2)
    DQQ_CASE(8, 1) DQQ_CASE(8, 2) DQQ_CASE(8, 4)
    DQQ_CASE(16, 4) DQQ_CASE(16, 8)
    DQQ_CASE(32, 16)
    DQQ_CASE(64, 32)
    if constexpr (kTuning) {   // ... and, behind "fwd_lpp" in the developer build, the others the sweeps compared them with
        DQQ_CASE(16, 2)
        DQQ_CASE(32, 4) DQQ_CASE(32, 8)
        DQQ_CASE(64, 8) DQQ_CASE(64, 16)
    }
#undef DQQ_CASE
    return false;
}

// In-kernel solve of non-diagonal tiles (fused) against queueing them for the dense kernel launched behind (work-list),
// forward, us per launch (tools/probe_fuse_n.py; diagonal batch QP / QCQP, then dense batch QP / QCQP):
//   N = 2  B = 131072: fused 14 / 19, 34 / 25, work-list 17 / 21, 53 / 46;  1048576: fused 68 / 87, 154 / 106,
//          work-list 74 / 92, 279 / 271                                      -> fused at every size
//   N = 4  B = 65536: fused 16 / 22, 38 / 38, work-list 17 / 24, 45 / 48;   1048576: fused 122 / 157, 371 / 350,
//          work-list 122 / 162, 321 / 344                                    -> fused up to 131072 problems
//   N = 8  (two lanes per problem, 128 VGPRs either way) B = 131072: fused 40 / 48, 222 / 214, work-list 43 / 50,
//          185 / 182;  1048576: fused 222 / 264, 1396 / 1338, work-list 236 / 278, 1308 / 1237
//          -> fused at every size: the diagonal batch is what DQQ_P_AUTO is for (a batch known to be dense has
//          DQQ_P_DENSE), and it saves the drain launch (2.5-5 us of a 56 us step at the bench shape)
// Only where the fallback solves a whole tile at once (group_dense.h, N <= 8): the per-problem fallback of N = 16
// makes a dense batch 3-10x slower than the work-list route (4096 x 16: 735 vs 81 us), for 3 us saved on a diagonal one.
bool fwd_diag_fuses_fallback(int N, long B)
{
    if (!(fwd_diag_supported(N) && fwd_diag_fuses(N) && N <= 8)) return false;
    return N == 4 ? B <= 131072 : true;
}

// DQQ_P_DENSE batches the fused kernel's group solve (group_dense.h) takes from the lane-per-problem kernel:
// N = 8 below 32 Ki problems, where 64 problems per wave leave most of the chip idle (4096 x 8: 44 vs 86 us).
bool fwd_diag_takes_dense(int kind, int N, long B) { return kind < 2 && N == 8 && B <= 32768; }

bool fwd_diag_will_fuse(int N, long B, int layout, int fuse_opt)
{
    // a batch declared dense is only ever sent here for the sizes whose general routine lives in this kernel
    // (group_dense.h: N <= 8); anything else would queue tiles on a work-list the DENSE route does not have
    if (layout == DQQ_P_DENSE) return fwd_diag_fuses(N) && N <= 8;
    return layout != DQQ_P_DIAG && fwd_diag_supported(N) && fwd_diag_fuses(N) &&
           (fuse_opt < 0 ? fwd_diag_fuses_fallback(N, B) : fuse_opt != 0);
}


// lpp / wpb == 0 -> built-in choice; an lpp the kernel is not instantiated for falls back to the
// built-in one.  *needs_fallback: the caller must launch the dense kernel in work-list mode next.
hipError_t launch_fwd_diag(int kind, const FwdArgs& a, int lpp, int wpb, int fuse_opt, hipStream_t s,
                           bool* needs_fallback)
{
    if (wpb != 1 && wpb != 4) wpb = 4;
    const bool fuse = fwd_diag_will_fuse(a.N, a.B, a.layout, fuse_opt);
    if (lpp <= 0) {
        lpp = fwd_diag_default_lpp(a.N, a.B, kind);
        // a batch declared dense takes the general solve's own mapping (N/2 lanes per problem): one pass per tile
        if (a.layout == DQQ_P_DENSE && a.N <= 8) lpp = a.N / 2;
        // A DQQ_P_AUTO batch the caller expects to be half or more non-diagonal (DQQ_F_EXPECT_DENSE: dqq_hint_flags from the
        // report word of the last backward, launch.h) runs on ONE lane per problem: the general solve with a problem's whole matrix
        // in its lane's registers instead of four lanes exchanging rows (65536 x 8 all non-diagonal, forward: QP 106 -> 65 us,
        // QCQP 116 -> 84; one problem in 10: 95 -> 68, 109 -> 82; tools/probe_sparse_dense_lpp.py).  Below that share two
        // lanes stay: since non-diagonal problems are handed to the general solve one by one (the kernel above), a sparse
        // few cost one pass of it per affected 16-problem block -- one in 1000: 46 / 52 us, what four lanes per problem took.
        // The same bits either way: neither the diagonal arithmetic nor the general solve depends on the lane layout, and
        // which of the two a problem gets depends on the problem alone.
        if (a.layout == DQQ_P_AUTO && fuse && a.N == 8 && kind < 2 && lpp == 2 && knob_fwd_feedback() != 0 &&
            (a.hints & DQQ_F_EXPECT_DENSE) != 0) {
            lpp = 1;
            g_fwd_feedback_routes.fetch_add(1, std::memory_order_relaxed);
        }
    }
    if (needs_fallback) *needs_fallback = (a.layout == DQQ_P_AUTO) && !fuse;
    hipError_t e = hipErrorInvalidValue;
    auto dispatch = [&](int l) {
        switch (kind) {
        case 0: return launch_kind<0>(a, l, wpb, fuse, s, e);
        case 1: return launch_kind<1>(a, l, wpb, fuse, s, e);
        case 2: return launch_kind<2>(a, l, wpb, fuse, s, e);
        case 3: return launch_kind<3>(a, l, wpb, fuse, s, e);
        default: return false;
        }
    };
    bool found = dispatch(lpp);
    if (!found) found = dispatch(fwd_diag_default_lpp(a.N, a.B, kind));
    return found ? e : hipErrorInvalidValue;
}

} // namespace dqq

