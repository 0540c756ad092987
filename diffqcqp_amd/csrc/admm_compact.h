// admm_compact.h -- the diagonal-P ADMM solve of admm_core.h with the tiles of a WORKGROUP repacked on the way.
//
// Why: the problems of a wave tile stop after different numbers of iterations (bench distribution: mean 17.6,
// a tile of 32 runs 26.6), and a wave keeps issuing whole iterations for its last problem.  Here the waves of a
// workgroup meet at fixed iteration counts ("checkpoints").  At a checkpoint every lane whose problem has stopped
// writes its result and becomes a HOLE; if the problems still running in the workgroup fit into one wave fewer,
// the wave holding the fewest of them (the DONOR) hands each of its running problems -- the whole per-lane state,
// through LDS -- to a hole of another wave and ends.  The arithmetic of a problem does not depend on where it
// runs (same lane parity inside its group, same instruction stream: admm_diag_body.inc), so x and the iteration
// counts are bit-identical to admm_fwd_diag; only the number of wave-iterations issued changes.
//
// Synchronisation: one s_barrier per checkpoint for the count exchange and one more when a donor exists.  Waves
// that have ended do not take part in barriers (hardware counts the surviving waves only); every surviving wave
// derives the same `live` mask, donor and hole numbering from the same shared counts.
#pragma once

#include "admm_core.h"

namespace dqq {

#if defined(__HIPCC__)

// first checkpoint and spacing: nothing stops before ~10 iterations at the reference's tolerances; later
// checkpoints are spaced wider so that a long solve (tight eps) pays a bounded share for them.  Every checkpoint
// costs the workgroup the imbalance of its waves over the segment (measured 0.5 us median, 1 us p90, at three
// iterations per segment; tools/probe_timeline.py), so they are few.
constexpr int kCompactFirst = 18;
DQQ_HD int compact_next(int c) { return c + (c < 30 ? 4 : c < 60 ? 8 : c < 120 ? 16 : 32); }

// Workgroup barrier for hand-offs through LDS only: waits for this wave's LDS operations, not for its global
// stores (__syncthreads() also drains vmcnt, i.e. it would stall every checkpoint for the round trip of the
// results just written).
DQQ_D void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int KIND, int E>
struct CompactLds {
    static constexpr int kSlots = 5 * E + (KIND == 1 ? E / 2 : 0) + 5 + 2;
    int cnt[2][8];
    double state[kSlots][64];
};

// valid: the lane starts with a problem; idx: its index in the batch (carried along when the problem moves).
// Results are stored from here (x, iters), `flags`/`pdiag` style outputs are the caller's business.
template <int KIND, int E, int LPP, int WPB>
DQQ_D void admm_fwd_diag_compact(const double (&p)[E], const double (&q_in)[E], const double* rad_in, int n,
                                 double eps, double mu, int max_iter, int adaptive, bool valid, long idx,
                                 double* __restrict__ x_out, int* __restrict__ iters_out, CompactLds<KIND, E>& sh,
                                 int wave, unsigned live)
{
    static_assert(KIND == 0 || KIND == 1, "QP and QCQP only");
    static_assert(WPB <= 8, "cnt[] holds 8 waves");
    using G = LaneGroup<LPP>;
    constexpr bool QP_LIKE = (KIND != 1);
    constexpr int PPW = 64 / LPP;
    constexpr int NR = (KIND == 1) ? E / 2 : 1;
    const int lane = threadIdx.x & 63;
    const double *lo = nullptr, *hi = nullptr, *sg = nullptr;
    double q[E], radv[NR];
#pragma unroll
    for (int e = 0; e < E; ++e) q[e] = q_in[e];
#pragma unroll
    for (int c = 0; c < NR; ++c) radv[c] = (KIND == 1) ? rad_in[c] : 0.0;
    const double* rad = radv;
    (void)lo; (void)hi; (void)sg;
#include "admm_diag_prologue.inc"

    int rho_up = 0, cpt = 0, iters = 0;
    bool act = valid;      // the lane's problem is still iterating
    bool written = !valid; // nothing (more) to store from this lane: a hole
    int base = 0, par = 0;
#ifdef DQQ_TIMELINE
    unsigned long long issued = 0;
    int ncp = 0; // checkpoint number: slots 8 + 3 * ncp + {0: segment done, 1: results written, 2: counts exchanged}
#endif
    int lim = kCompactFirst < max_iter ? kCompactFirst : max_iter;
    DQQ_TL(3);
    for (;;) {
        if (act) {
            for (int it = base; it < lim; ++it) {
#define DQQ_ADMM_ON_STOP { act = false; break; }
#include "admm_diag_body.inc"
#undef DQQ_ADMM_ON_STOP
            }
        }
#ifdef DQQ_TIMELINE
        {
            int m = iters > base ? iters - base : 0; // iterations this lane ran in the segment
            for (int o = 32; o; o >>= 1) m = max(m, __shfl_xor(m, o));
            issued += (unsigned long long)m;
        }
#endif
#ifdef DQQ_TIMELINE
        if (ncp < 8) DQQ_TL(8 + 3 * ncp);
#endif
        const bool last = lim >= max_iter; // wave-uniform: the iteration budget is spent (Solver.cpp:79 loop bound)
        if (last) act = false;
        if (!act && !written) {
            const bool b = G::max(bad ? 1.0 : 0.0) > 0.0;
            double* xx = x_out + idx * n + (lane % LPP) * E;
#pragma unroll
            for (int e = 0; e < E; e += 2)
                *reinterpret_cast<double2*>(xx + e) = make_double2(b ? NAN : l2[e], b ? NAN : l2[e + 1]);
            if (iters_out != nullptr && (lane % LPP) == 0) iters_out[idx] = iters;
            written = true;
        }
        if (last) break;
#ifdef DQQ_TIMELINE
        if (ncp < 8) DQQ_TL(8 + 3 * ncp + 1);
#endif

        // ---- checkpoint: who is still running, in every surviving wave of the workgroup
        const unsigned long long amask = __ballot(act);
        const int mine = __popcll(amask) / LPP;
        if (lane == 0) sh.cnt[par][wave] = mine;
        lds_barrier();
#ifdef DQQ_TIMELINE
        if (ncp < 8) DQQ_TL(8 + 3 * ncp + 2);
        ++ncp;
#endif
        int c[WPB], total = 0, nz = 0, donor = -1, dmin = PPW + 1;
#pragma unroll
        for (int w = 0; w < WPB; ++w) {
            c[w] = ((live >> w) & 1u) ? __builtin_amdgcn_readfirstlane(sh.cnt[par][w]) : 0;
            total += c[w];
            nz += c[w] > 0;
            if (c[w] > 0 && c[w] <= dmin) { dmin = c[w]; donor = w; } // fewest running problems, last such wave
        }
        par ^= 1;
#ifdef DQQ_COMPACT_NOGIVE
        const bool give = false;
#else
        const bool give = nz >= 2 && total <= PPW * (nz - 1);
#endif // the others' holes take the donor's problems
        if (mine == 0) break;                                 // nothing left here (everything is written): end
        if (give) {
            int hole_base = 0; // holes of the receiving waves before this one, in wave order
#pragma unroll
            for (int w = 0; w < WPB; ++w)
                if (w < wave && w != donor && c[w] > 0) hole_base += PPW - c[w];
            const int below = __builtin_amdgcn_mbcnt_hi((unsigned)(amask >> 32),
                                                        __builtin_amdgcn_mbcnt_lo((unsigned)amask, 0u));
            if (wave == donor) {
                if (act) {
                    const int slot = (below / LPP) * LPP + (lane % LPP); // running problems of the donor, packed
                    int k = 0;
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        sh.state[k + 0 * E][slot] = M[e];
                        sh.state[k + 1 * E][slot] = q[e];
                        sh.state[k + 2 * E][slot] = qp[e];
                        sh.state[k + 3 * E][slot] = l2[e];
                        sh.state[k + 4 * E][slot] = u[e];
                        ++k;
                    }
                    k = 5 * E;
                    if (KIND == 1) {
#pragma unroll
                        for (int cc = 0; cc < NR; ++cc) sh.state[k++][slot] = radv[cc];
                    }
                    sh.state[k++][slot] = rho;
                    sh.state[k++][slot] = inv_rho;
                    sh.state[k++][slot] = tau_inc;
                    sh.state[k++][slot] = tau_dec;
                    sh.state[k++][slot] = Mmin;
                    sh.state[k++][slot] = __longlong_as_double(idx);
                    sh.state[k++][slot] = __longlong_as_double((long long)((rho_up + 1) | (cpt << 2) | ((int)bad << 5)));
                }
                lds_barrier();
                break; // the donor's running problems live elsewhere now; its stopped ones are written
            }
            lds_barrier();
            // a hole of a receiving wave: lane groups are numbered through the holes of the workgroup
            const int hbelow = lane - below; // holes (lanes not running) below this lane
            const int h = hole_base + hbelow / LPP;
            if (!act && h < dmin) { // dmin == the donor's count
                const int slot = h * LPP + (lane % LPP);
                int k = 0;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    M[e] = sh.state[k + 0 * E][slot];
                    q[e] = sh.state[k + 1 * E][slot];
                    qp[e] = sh.state[k + 2 * E][slot];
                    l2[e] = sh.state[k + 3 * E][slot];
                    u[e] = sh.state[k + 4 * E][slot];
                    ++k;
                }
                k = 5 * E;
                if (KIND == 1) {
#pragma unroll
                    for (int cc = 0; cc < NR; ++cc) radv[cc] = sh.state[k++][slot];
                }
                rho = sh.state[k++][slot];
                inv_rho = sh.state[k++][slot];
                tau_inc = sh.state[k++][slot];
                tau_dec = sh.state[k++][slot];
                itau_inc = fast_rcp(tau_inc);
                itau_dec = fast_rcp(tau_dec);
                Mmin = sh.state[k++][slot];
                idx = __double_as_longlong(sh.state[k++][slot]);
                const int packed = (int)__double_as_longlong(sh.state[k++][slot]);
                rho_up = (packed & 3) - 1;
                cpt = (packed >> 2) & 7;
                bad = ((packed >> 5) & 1) != 0;
                rcp_all<E>(M, Minv); // what the donor held: Minv is always rcp_all(M)
                act = true;
                written = false;
            }
            // the donor is gone; a wave without running problems has left at `mine == 0`
            unsigned nl = 0;
#pragma unroll
            for (int w = 0; w < WPB; ++w)
                if (c[w] > 0 && w != donor) nl |= 1u << w;
            live = nl;
        } else {
            unsigned nl = 0;
#pragma unroll
            for (int w = 0; w < WPB; ++w)
                if (c[w] > 0) nl |= 1u << w;
            live = nl;
        }
        base = lim;
        const int nx = compact_next(lim);
        lim = nx < max_iter ? nx : max_iter;
    }
    DQQ_TL(4);
#ifdef DQQ_TIMELINE
    DQQ_TL_VAL(6, issued);
#endif
}

#endif // __HIPCC__

} // namespace dqq
