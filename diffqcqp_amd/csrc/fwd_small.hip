// fwd_small.hip -- general (dense P) forward solve for 8 < N <= 16 (5 to 8 contacts): a TEAM of 16 lanes
// per problem, four problems per wave, statically sized loops.
//
// Lane i of a team owns coordinate i: row i of P (power iteration, lower triangle for the factorisation),
// row i of the Cholesky factor and row/column i of M^-1 = (P + (rho+mu) I)^-1 live in its registers; a
// mat-vec is one LDS store of the lane's vector entry, N broadcast reads and N multiply-adds; norms and the
// residual maxima are DPP reductions over the team.  Every lane of a team carries identical copies of the
// scalar state (rho, tau, cpt, ...).  The refactorisation at a rho update (left-looking LLT as Eigen's,
// then the explicit inverse column by column, Solver.cpp:76-77) runs under the exec mask of the teams that
// fire in that iteration.
//
// Algorithm, update order, constants and stopping tests are the reference's (Solver::solveQP / solveQCQP /
// solveBoxQP / solveSignedBoxQP, Solver.cpp:61-123, 521-582, 198-261, 374-439; power_iteration :46-59), as
// in fwd_lane_dense.hip (N <= 8), which this file mirrors.  Ulp-level departures as there: FMA contraction,
// reciprocal-multiply instead of divide (1-ulp rcp / rsqrt), team (tree) sums for the 2-norms, and M^-1
// used through its columns (symmetric up to rounding).
#include "admm_core.h"
#include "launch.h"

namespace dqq {

int lane_defer_for(int kind); // fwd_lane_dense.hip (option lane_defer)

template <int N>
struct SmallFwd {
    static constexpr int T = 16;                       // team width (8 < N <= 16)
    static constexpr int LDA = N;                      // N even: rows of L start 16-byte aligned
    static constexpr int LDS_DOUBLES = N * LDA + 2 * N; // L, two vector buffers
};

// One lane's view of its team's LDS slice and the pieces built on it.
template <int N>
struct TeamSolve {
    using S = SmallFwd<N>;
    double* Lm;   // N x LDA: Cholesky factor, rows published as they are completed
    double* vec;  // 2 x N: mat-vec exchange buffers (alternating)
    int lane;     // team-local
    int flip;

    DQQ_D void init(double* smem, int l) { Lm = smem; vec = smem + N * S::LDA; lane = l; flip = 0; }

    // sum_j row[j] * v_j, v = this lane's entry of the vector (lanes >= N pass 0)
    DQQ_D double matvec(const double (&row)[N], double v)
    {
        double* buf = vec + flip * N;
        flip ^= 1;
        if (lane < N) buf[lane] = v;
        wave_lds_fence();
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < N; ++j) s += row[j] * buf[j];
        return s;
    }

    // Explicit inverse of the symmetric matrix with strict lower triangle Prow[j < lane] and diagonal `diag`
    // (this lane's entry): left-looking Cholesky, then L y = e_c, L^T x = y for column c = lane.
    DQQ_D void chol_inverse(const double (&Prow)[N], double diag, double (&Minv)[N], bool& bad)
    {
        double Lr[N], rinv[N];
        // every lane needs every diagonal entry: one exchange
        double* dg = vec + flip * N;
        flip ^= 1;
        if (lane < N) dg[lane] = diag;
        wave_lds_fence();
#pragma unroll
        for (int k = 0; k < N; ++k) {
            double s = 0.0, t = 0.0;
#pragma unroll
            for (int j = 0; j < k; ++j) {
                const double lkj = Lm[k * S::LDA + j];
                s += lkj * lkj;
                t += Lr[j] * lkj;
            }
            const double piv = dg[k] - s;
            bad = bad || !(piv > 0.0);
            const double rs = fast_rsqrt(piv);
            rinv[k] = rs;
            Lr[k] = (lane == k) ? piv * rs : (Prow[k] - t) * rs;
            if (lane >= k && lane < N) Lm[lane * S::LDA + k] = Lr[k];
            wave_lds_fence();
        }
#pragma unroll
        for (int i = 0; i < N; ++i) {
            double t = (i == lane) ? 1.0 : 0.0;
#pragma unroll
            for (int j = 0; j < i; ++j) t -= Lm[i * S::LDA + j] * Minv[j];
            Minv[i] = t * rinv[i];
        }
#pragma unroll
        for (int i = N - 1; i >= 0; --i) {
            double t = Minv[i];
#pragma unroll
            for (int j = i + 1; j < N; ++j) t -= Lm[j * S::LDA + i] * Minv[j];
            Minv[i] = t * rinv[i];
        }
        wave_lds_fence(); // Lm is rewritten by the next factorisation
    }
};

// The power-iteration vector is normalised after every step, like the reference (Solver.cpp:53), by a 1-ulp
// reciprocal square root: lambda_max^10 between two normalisations would leave the double range for
// |lambda_max| beyond ~1e15 or below ~1e-15, which the diagonal fast path (exact power-of-two scaling) handles.
template <int KIND, int N>
__global__ __launch_bounds__(256, N >= 14 ? 1 : 2) void fwd_small_kernel(const double* __restrict__ P, const double* __restrict__ q,
                                                        const double* __restrict__ l_n,
                                                        const double* __restrict__ mu_c,
                                                        const double* __restrict__ v_sign, double* __restrict__ x,
                                                        long B, double eps, double mu, int max_iter, int adaptive,
                                                        int* __restrict__ iters, int* __restrict__ ws, int use_worklist, int defer)
{
    using S = SmallFwd<N>;
    using G = LaneGroup<S::T>;
    constexpr int T = S::T, TP = 64 / T;
    constexpr bool QP_LIKE = (KIND != 1);
    static_assert(N % 2 == 0 && N <= T, "even N <= 16");
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const int team = lane / T, tl = lane % T;
    if (use_worklist && ws[kWsCount] == 0) return;   // an empty list: one scalar load, before the hygiene checks of launch.h
    const long count = use_worklist ? worklist_checked_count(ws, ws + kWsCount, kWsEntryInts(B)) : B;
    const long slot = ((long)blockIdx.x * wpb + wave) * TP + team;
    if (((long)blockIdx.x * wpb + wave) * TP >= count) { // a wave beyond the end of the list: only the reset ticket
        if (use_worklist && lane == 0) worklist_release(ws, count, (int)(gridDim.x * wpb));
        return;
    }
    const bool valid = slot < count;
    const long prob = valid ? (use_worklist ? worklist_checked_entry(ws, ws[kWsEntries + slot], B) : slot) : 0;
    const bool actn = tl < N;
    const int i = actn ? tl : 0;
    TeamSolve<N> ts;
    ts.init(smem + (wave * TP + team) * S::LDS_DOUBLES, tl);

    // ---- load: row i of P, q_i, the constraint data of coordinate i
    double Prow[N];
    {
        const double* Pg = P + prob * (long)(N * N) + i * N;
#pragma unroll
        for (int j = 0; j < N; j += 2) {
            const double2 t = (valid && actn) ? *reinterpret_cast<const double2*>(Pg + j) : make_double2(0.0, 0.0);
            Prow[j] = t.x;
            Prow[j + 1] = t.y;
        }
        if (!(valid && actn)) {
#pragma unroll
            for (int j = 0; j < N; ++j) if (j == i) Prow[j] = 1.0;
        }
    }
    const double qi = (valid && actn) ? q[prob * N + i] : 0.0;
    double rad = 1.0, blo = 0.0, bhi = 0.0, bsg = 0.0;
    if (valid && actn) {
        if (KIND == 1) rad = l_n[prob * (N / 2) + i / 2] * mu_c[prob * (N / 2) + i / 2]; // pybindings.cpp:57
        if (KIND >= 2) { blo = l_n[prob * N + i]; bhi = mu_c[prob * N + i]; }
        if (KIND == 3) { const double vv = v_sign[prob * N + i]; bsg = (double)((vv > 0) - (vv < 0)); } // :395
    }

    // ---- power_iteration, Solver.cpp:46-59
    double Lmax;
    {
        double v = actn ? 1.0 / sqrt((double)N) : 0.0;
        {
            const double s = G::sum(v * v);
            if (s > 0) v = v / sqrt(s);
        }
        const int pi_steps = QP_LIKE ? 10 : 100;
        for (int k = 0; k < pi_steps; ++k) {
            double Av = ts.matvec(Prow, v);
            if (!actn) Av = 0.0;
            const double s = G::sum(Av * Av);
            v = s > 0 ? Av * fast_rsqrt(s) : Av;
        }
        double Av = ts.matvec(Prow, v);
        if (!actn) Av = 0.0;
        Lmax = G::sum(v * Av);
    }

    // ---- Solver.cpp:72-77 / 531-536
    RhoSchedule sched;
    sched.init(Lmax, mu);
    double rho = sched.rho;
    double inv_rho = fast_rcp(rho);
    bool bad = !(rho > 0.0) || !(rho < 1.79e308);
    double md = 0.0; // the accumulated shifted diagonal entry of this coordinate
#pragma unroll
    for (int j = 0; j < N; ++j) if (j == i) md = Prow[j] + (rho + mu);
    double Minv[N];
    ts.chol_inverse(Prow, md, Minv, bad);

    double qp = qi, l2 = 0.0, u = 0.0;
    // The refactorisation after a rho update is deferred as in fwd_lane_dense.hip (option lane_defer): the team updates
    // rho, 1/rho and its shifted diagonal entry on the spot and sits out until the wave next runs chol_inverse -- every
    // `defer`-th trip, or as soon as no team has anything else to do.  A problem's arithmetic does not depend on it.
    int it_done = 0;
    bool done = !valid || max_iter <= 0, pend = false;
    for (int trip = 0;; ++trip) {
        if (!done && !pend) {
            const double rhs = actn ? rho * l2 - u - qp : 0.0;
            const double l = ts.matvec(Minv, rhs);                                  // :80 / :539
            qp = qi - mu * l;                                                     // :81 / :540
            const double w = kAlpha * l + (1 - kAlpha) * l2;
            double z = w + u * inv_rho;                                           // :82 / :541
            if (KIND == 0) {
                z = fmax(z, 0.0);
            } else if (KIND >= 2) {
                z = z < blo ? blo : z;                                            // cwiseMax(l_min), :219 / :396
                z = bhi < z ? bhi : z;                                            // cwiseMin(l_max), :220 / :397
                if (KIND == 3) {                                                  // v o min(v o l_2, 0), :398
                    double m = bsg * z;
                    m = 0 < m ? 0 : m;
                    z = bsg * m;
                }
            } else {                                                              // prox_circle, :505-519
                const double other = partner<1>(z);
                const double a = (tl & 1) ? other : z, b = (tl & 1) ? z : other; // both lanes: the same expression
                const double n2 = __builtin_fma(b, b, a * a);
                const double rn = fast_rsqrt(n2);
                if (n2 > rad * fabs(rad)) z = z * (rad * rn);   // (on the squares: admm_diag_body.inc, fwd_lane_dense.hip)
            }
            u += rho * (w - z);                                                   // :83 / :543
            const double rd = G::max(actn ? fabs(z - l2) : 0.0);
            const double rp = G::max(actn ? fabs(z - w) : 0.0);
            l2 = z;
            const double res_dual = rho * rd, res_prim = rp;
            it_done += 1;
            bool stop = res_dual < eps;                                           // :88
            if (KIND == 1) {
                const double nl = G::sum(actn ? l * l : 0.0);
                if (stop) stop = res_prim < eps + kEpsRel * sqrt(nl);             // :548
            }
            done = stop || it_done >= max_iter;
            if (!done && adaptive) {
                double delta;
                if (sched.template update<QP_LIKE, true>(res_prim, res_dual, delta)) { // Solver.cpp:90-120 / 550-580
                    rho = sched.rho;
                    inv_rho = fast_rcp(rho);
                    md += delta;
                    pend = true;
                }
            }
        }
        if (__all(done)) break;
        if (__any(pend) && ((trip + 1) % defer == 0 || !__any(!done && !pend))) {
            if (pend) ts.chol_inverse(Prow, md, Minv, bad);                       // llt() + solveInPlace(Identity)
            pend = false;
        }
    }

    bad = G::max(bad ? 1.0 : 0.0) > 0.0;
    if (valid && actn) {
        x[prob * N + i] = bad ? NAN : l2;
        if (iters != nullptr && tl == 0) iters[prob] = it_done;
    }
    if (use_worklist && lane == 0) worklist_release(ws, count, (int)(gridDim.x * wpb));
}

template <int KIND, int N>
static hipError_t launch_small_fwd(const FwdArgs& a, bool use_worklist, hipStream_t s)
{
    using S = SmallFwd<N>;
    constexpr int TP = 64 / S::T, WPB = 4;
    const size_t lds_bytes = sizeof(double) * (size_t)S::LDS_DOUBLES * TP * WPB;
    const long per_block = (long)WPB * TP;
    const long nb = (a.B + per_block - 1) / per_block;
    if (nb == 0) return hipSuccess;
    return launch((fwd_small_kernel<KIND, N>), dim3((unsigned)nb), dim3(64 * WPB), lds_bytes, s, a.P, a.q, a.l_n, a.mu,
                       a.v, a.x, a.B, a.eps, a.mu_prox, a.max_iter, a.adaptive, a.iters, a.ws, use_worklist ? 1 : 0,
                       lane_defer_for(KIND));
}

bool fwd_small_supported(int N) { return N == 10 || N == 12 || N == 14 || N == 16; }

hipError_t launch_fwd_small(int kind, const FwdArgs& a, bool use_worklist, hipStream_t s)
{
    if (a.B == 0) return hipSuccess;
#define DQQ_CASE(NN)                                                          \
    if (a.N == NN) {                                                          \
        switch (kind) {                                                       \
        case 0: return launch_small_fwd<0, NN>(a, use_worklist, s);           \
        case 1: return launch_small_fwd<1, NN>(a, use_worklist, s);           \
        case 2: return launch_small_fwd<2, NN>(a, use_worklist, s);           \
        case 3: return launch_small_fwd<3, NN>(a, use_worklist, s);           \
        default: return hipErrorInvalidValue;                                 \
        }                                                                     \
    }
    DQQ_CASE(10) DQQ_CASE(12) DQQ_CASE(14) DQQ_CASE(16)
#undef DQQ_CASE
    return hipErrorInvalidValue;
}

} // namespace dqq
