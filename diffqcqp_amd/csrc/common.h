// common.h -- shared definitions for the gfx950 kernels of libdiffqcqp_hip.so.
//
// The per-problem arithmetic lives in host/device templates (admm_core.h,
// kkt_core.h) parameterised by a "group" type that supplies the cross-lane
// reductions.  On the device a group is LPP adjacent lanes of one wave64 that
// share a problem; tests/host_core_check.cpp instantiates the same templates
// with HostGroup (one lane per problem) to check them against the oracle on a
// machine without a GPU.  That host build is a test artefact: the C ABI never
// calls it.
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define DQQ_HD __host__ __device__ __forceinline__
#define DQQ_D __device__ __forceinline__
#else
#define DQQ_HD inline
#endif


namespace dqq {

// ADMM constants of the reference (Solver.cpp:64, 523-524)
constexpr double kMuThresh = 10.0;
constexpr double kAlpha = 1.5;
constexpr double kEpsRel = 1e-4;
// backward constants (Solver.cpp:15 defaults, :140, :639; pybindings.cpp:24,62)
constexpr double kMuIr = 1e-7;
constexpr double kIrEps = 1e-10;
constexpr int kIrMaxIter = 10;
constexpr double kActiveEps = 1e-10;

// 1/sqrt(x) and 1/x to ~1 ulp: hardware seed (v_rsq_f64 / v_rcp_f64, ~2^-24
// relative) + one third-order step (rsqrt) / two Newton steps (rcp) -- about a
// third of the instructions of the IEEE sqrt + divide sequences they stand in
// for in the forward fast path.
DQQ_HD double fast_rsqrt(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    // one third-order step: with e = 1 - x y0^2, 1/sqrt(x) = y0 (1 + e/2 + 3e^2/8 + O(e^3)); e ~ 2^-24 from the
    // hardware seed, so the truncation error is ~2^-70 and the result is rounding-limited (five dependent
    // instructions instead of the eight of two Newton steps)
    const double y0 = __builtin_amdgcn_rsq(x);
    const double e = fma(-(x * y0), y0, 1.0);
    const double p = fma(e, 0.375, 0.5);
    return fma(y0 * e, p, y0);
#else
    return 1.0 / sqrt(x);
#endif
}

DQQ_HD double fast_rcp(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
#if defined(DQQ_RCP_TWO_NEWTON)   // (the form of rounds 1-4: two Newton steps, four dependent instructions behind the seed)
    double y = __builtin_amdgcn_rcp(x);
    double e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    return y;
#else
    // one second-order step, like fast_rsqrt above: with e = 1 - x y0, 1/x = y0 (1 + e + e^2 + O(e^3)); e ~ 2^-26 from the
    // hardware seed, so the truncation error is ~2^-78 and the result is rounding-limited -- three dependent instructions
    // behind the seed instead of four
    const double y0 = __builtin_amdgcn_rcp(x);
    const double e = fma(-x, y0, 1.0);
    const double t = fma(e, e, e);
    return fma(y0, t, y0);
#endif
#else
    return 1.0 / x;
#endif
}

// The rho adaptation of the reference's ADMM loops (Solver.cpp:90-120, 228-258, 405-435, 550-580) as one
// step of a small state machine: rho is increased when the primal residual dominates, decreased when the
// dual one does, at most once every 5 imbalanced iterations (the reference's `cpt % 5 == 0; cpt++`, kept
// here modulo 5); when the direction flips the step factors are damped -- both for the QP-like loops
// (:94-97, :108-111), only the one in use for the QCQP (:554-556, :568-570).
// update() returns true when rho changed; `delta` is what the reference adds to the diagonal of the
// shifted matrix (:98 / :112).  FAST: rho /= tau_dec as a multiplication by a 1-ulp reciprocal (fast paths);
// otherwise the reference's exact expressions.
struct RhoSchedule {
    double rho, tau_inc, tau_dec;
    int rho_up, cpt;

    DQQ_HD void init(double L, double mu)
    {
        rho = sqrt(mu * L) * pow(L / mu, .4);        // :72 / :531
        tau_inc = pow(L / mu, .15);                  // :73 / :532
        tau_dec = tau_inc;
        rho_up = 0;
        cpt = 0;
    }
    template <bool QP_LIKE, bool FAST>
    DQQ_HD bool update(double res_prim, double res_dual, double& delta)
    {
        const bool inc = res_prim > kMuThresh * res_dual;               // :92 / :552
        const bool dec = !inc && (res_dual > kMuThresh * res_prim);     // :106 / :566
        const bool imb = inc || dec;
        const bool fire = imb && (cpt == 0);                            // cpt % 5 == 0
        cpt = imb ? (cpt == 4 ? 0 : cpt + 1) : cpt;                     // cpt++ (kept mod 5)
        if (!fire) return false;
        if (rho_up == (inc ? -1 : 1)) {                                 // direction flipped: damp tau
            const double ti = 1 + .8 * (tau_inc - 1), td = 1 + .8 * (tau_dec - 1);
            if (QP_LIKE) { tau_inc = ti; tau_dec = td; }
            else if (inc) tau_inc = ti;
            else tau_dec = td;
        }
        if (FAST) {
            const double f = inc ? tau_inc : fast_rcp(tau_dec);
            delta = rho * (f - 1);
            rho = rho * f;
        } else if (inc) {
            delta = rho * (tau_inc - 1);                                // :98
            rho *= tau_inc;                                             // :99
        } else {
            delta = rho * (1. / tau_dec - 1);                           // :112
            rho /= tau_dec;                                             // :113
        }
        rho_up = inc ? 1 : -1;
        return true;
    }
};

// max(a, b) as ONE v_max_f64: fmax() makes the compiler quiet both operands first (a v_max_f64 x, x, x each)
// whenever it cannot prove them canonical, e.g. after a DPP move.  Same result for every non-signalling input.
DQQ_HD double max_raw(double a, double b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    return fmax(a, b);
#endif
}

// max(|a|, |b|) and max(a, |b|), one instruction each (source modifiers)
DQQ_HD double max_abs2(double a, double b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double r;
    asm("v_max_f64 %0, |%1|, |%2|" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    return fmax(fabs(a), fabs(b));
#endif
}
DQQ_HD double max_abs1(double a, double b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double r;
    asm("v_max_f64 %0, %1, |%2|" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    return fmax(a, fabs(b));
#endif
}

// One lane owns the whole problem: reductions are the identity.
struct HostGroup {
    static constexpr int kLanes = 1;
    static DQQ_HD double sum(double v) { return v; }
    static DQQ_HD double max(double v) { return v; }
    static DQQ_HD void max2(double&, double&) {}
    static DQQ_HD bool wave_all(bool b) { return b; }
    // x^0.4 and x^0.15 (rho and tau of Solver.cpp:72-73)
    static DQQ_HD void pow_pair(double x, double& p40, double& p15) { p40 = pow(x, .4); p15 = pow(x, .15); }
};

#if defined(__HIPCC__)

constexpr int kWave = 64;

// DPP controls (GFX9 encoding)
constexpr int kDppXor1 = 0xB1;        // quad_perm [1,0,3,2]
constexpr int kDppXor2 = 0x4E;        // quad_perm [2,3,0,1]
constexpr int kDppHalfMirror = 0x141; // row_half_mirror: lane i <-> 7-i within 8
constexpr int kDppRor8 = 0x128;       // row_ror:8: lane i <-> i^8 within 16

template <int CTRL>
DQQ_D double dpp_f64(double v)
{
    // every control used here reads a valid lane for every lane, so the "old" operand never shows: leaving it
    // unbound (0 + bound_ctrl) spares the v_mov that would otherwise seed the destination of each half
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// Value held by the lane whose index differs in bit log2(STEP), valid when the
// value is already uniform over aligned groups of STEP lanes (which is what an
// all-reduce butterfly guarantees).  Pairings are symmetric, so both partners of
// a pair compute bit-identical results for commutative ops: all lanes of a
// problem take identical branches.
template <int STEP>
DQQ_D double partner(double v)
{
    if constexpr (STEP == 1) return dpp_f64<kDppXor1>(v);
    else if constexpr (STEP == 2) return dpp_f64<kDppXor2>(v);
    else if constexpr (STEP == 4) return dpp_f64<kDppHalfMirror>(v);
    else if constexpr (STEP == 8) return dpp_f64<kDppRor8>(v);
    else return __shfl_xor(v, STEP, 64);
}

// LPP adjacent lanes (LPP a power of two, aligned) share one problem.
template <int LPP>
struct LaneGroup {
    static constexpr int kLanes = LPP;
    // x^0.4 and x^0.15: with two or more lanes per problem the even lane evaluates one power and the odd lane
    // the other (one pow() per lane, a per-lane exponent), then they swap -- all lanes of the problem end up with
    // the same two bit patterns
    static DQQ_D void pow_pair(double x, double& p40, double& p15)
    {
        if constexpr (LPP >= 2) {
            const bool odd = (threadIdx.x & 1) != 0;
            const double mine = pow(x, odd ? .15 : .4);
            const double other = partner<1>(mine);
            p40 = odd ? other : mine;
            p15 = odd ? mine : other;
        } else {
            // the same pow() as above -- the one with an exponent the compiler does not know: with the literal the
            // library call is specialised and the last bit differs (round 4: a result must not depend on the layout)
            double e40 = .4, e15 = .15;
            asm volatile("" : "+v"(e40), "+v"(e15));
            p40 = pow(x, e40);
            p15 = pow(x, e15);
        }
    }
    static DQQ_D double sum(double v)
    {
        if constexpr (LPP >= 2) v = v + partner<1>(v);
        if constexpr (LPP >= 4) v = v + partner<2>(v);
        if constexpr (LPP >= 8) v = v + partner<4>(v);
        if constexpr (LPP >= 16) v = v + partner<8>(v);
        if constexpr (LPP >= 32) v = v + partner<16>(v);
        if constexpr (LPP >= 64) v = v + partner<32>(v);
        return v;
    }
    static DQQ_D double max(double v)
    {
        if constexpr (LPP >= 2) v = max_raw(v, partner<1>(v));
        if constexpr (LPP >= 4) v = max_raw(v, partner<2>(v));
        if constexpr (LPP >= 8) v = max_raw(v, partner<4>(v));
        if constexpr (LPP >= 16) v = max_raw(v, partner<8>(v));
        if constexpr (LPP >= 32) v = max_raw(v, partner<16>(v));
        if constexpr (LPP >= 64) v = max_raw(v, partner<32>(v));
        return v;
    }
    // two maxima at once, level by level: the two butterflies are independent chains (DPP move -> hazard slot -> v_max per
    // level) and a wave that is alone on its SIMD -- the last survivors of a launch -- pays every dependent step in full;
    // written one after the other the compiler also scheduled them one after the other.  Same bits as max() twice.
    static DQQ_D void max2(double& a, double& b)
    {
        if constexpr (LPP >= 2) { const double pa = partner<1>(a), pb = partner<1>(b); a = max_raw(a, pa); b = max_raw(b, pb); }
        if constexpr (LPP >= 4) { const double pa = partner<2>(a), pb = partner<2>(b); a = max_raw(a, pa); b = max_raw(b, pb); }
        if constexpr (LPP >= 8) { const double pa = partner<4>(a), pb = partner<4>(b); a = max_raw(a, pa); b = max_raw(b, pb); }
        if constexpr (LPP >= 16) { const double pa = partner<8>(a), pb = partner<8>(b); a = max_raw(a, pa); b = max_raw(b, pb); }
        if constexpr (LPP >= 32) { const double pa = partner<16>(a), pb = partner<16>(b); a = max_raw(a, pa); b = max_raw(b, pb); }
        if constexpr (LPP >= 64) { const double pa = partner<32>(a), pb = partner<32>(b); a = max_raw(a, pa); b = max_raw(b, pb); }
    }
    static DQQ_D bool wave_all(bool b) { return __all(b); }
};

// LDS traffic between lanes of ONE wave: program order is enough in hardware
// (a wave's DS operations execute in order); this only stops the compiler from
// moving accesses across the hand-off.
DQQ_D void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// true when the double is neither +0.0 nor -0.0 (NaN counts as non-zero)
DQQ_D unsigned nonzero_bits(double v)
{
    return (unsigned)__double2loint(v) | ((unsigned)__double2hiint(v) & 0x7fffffffu);
}

#endif // __HIPCC__

} // namespace dqq
