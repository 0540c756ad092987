// Cycles of the 16x16 diagonal-block factorisation (Cholesky + inverse in registers, block_core.h) by one
// wave, and candidate reformulations.  hipcc --offload-arch=gfx950 -O3 -I../../diffqcqp_amd/csrc -I../../include
#include "block_core.h"
#include <cstdio>
#include <cmath>
#include <hip/hip_runtime.h>
using namespace dqq;

// V1: same Cholesky, inverse in column-oriented order (independent updates instead of a serial dot product)
template <int N>
DQQ_D void diag_block_factor_v1(const double* W, double* LinvT, int kb, int l, bool& bad)
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
    // lane c = column c of L^-1: t[i] starts as e_c, y[j] = t[j]*rinv[j], then t[i] -= L[i][j]*y[j] for i > j
    double t[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) t[i] = (row == i) ? 1.0 : 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        t[j] = t[j] * rinv[j];
#pragma unroll
        for (int i = j + 1; i < 16; ++i) t[i] -= lane_bcast(w[j], i) * t[j];
    }
    if (l < 16) {
        double* yrow = LinvT + (16 * kb + row) * G::LD + 16 * kb;
#pragma unroll
        for (int j = 0; j < 16; ++j) yrow[j] = t[j];
    }
}

// V2: V1 + the columns of L are exchanged through LDS (one ds_write + broadcast reads per step) instead of
// 2 x (15-k) v_readlane per step
template <int N>
DQQ_D void diag_block_factor_v2(const double* W, double* LinvT, double* colbuf, int kb, int l, bool& bad)
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
        double* cb = colbuf + (k & 1) * 16;
        if (l < 16) cb[row] = w[k];
        wave_lds_fence();
#pragma unroll
        for (int j = k + 1; j < 16; ++j) w[j] -= w[k] * cb[j];
    }
    double t[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) t[i] = (row == i) ? 1.0 : 0.0;
    // L^T is needed row-wise here: L[i][j] for fixed j, all i -> column j of L = what colbuf held at step j;
    // keep all 16 columns in LDS instead (16 x 16 doubles)
    double* Lc = colbuf + 32;
#pragma unroll
    for (int j = 0; j < 16; ++j) if (l < 16) Lc[j * 16 + row] = w[j];
    wave_lds_fence();
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        t[j] = t[j] * rinv[j];
#pragma unroll
        for (int i = j + 1; i < 16; ++i) t[i] -= Lc[j * 16 + i] * t[j];
    }
    if (l < 16) {
        double* yrow = LinvT + (16 * kb + row) * G::LD + 16 * kb;
#pragma unroll
        for (int j = 0; j < 16; ++j) yrow[j] = t[j];
    }
}

constexpr int N = 64, REPS = 200;
template <int V>
__global__ __launch_bounds__(64) void bench(const double* Win, double* out, long long* cyc)
{
    using G = BlockGeom<N>;
    __shared__ __attribute__((aligned(16))) double W[N * G::LD], LT[N * G::LD], cb[32 + 256];
    const int l = threadIdx.x;
    for (int i = l; i < N * G::LD; i += 64) { W[i] = Win[i]; LT[i] = 0.0; }
    __syncthreads();
    bool bad = false;
    const long long t0 = clock64();
    for (int r = 0; r < REPS; ++r) {
        if (V == 0) diag_block_factor<N>(W, LT, r & 3, l, bad);
        if (V == 1) diag_block_factor_v1<N>(W, LT, r & 3, l, bad);
        if (V == 2) diag_block_factor_v2<N>(W, LT, cb, r & 3, l, bad);
        wave_lds_fence();
    }
    const long long t1 = clock64();
    __syncthreads();
    for (int i = l; i < N * G::LD; i += 64) out[i] = LT[i];
    if (l == 0) { cyc[0] = t1 - t0; cyc[1] = bad; }
}

template <int V>
void run(const double* dW, const double* ref, const char* name)
{
    using G = BlockGeom<N>;
    double* out; long long* cyc;
    hipMalloc(&out, sizeof(double) * N * G::LD); hipMalloc(&cyc, 16);
    bench<V><<<1, 64>>>(dW, out, cyc);
    bench<V><<<1, 64>>>(dW, out, cyc);
    static double h[N * G::LD]; long long hc[2];
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(hc, cyc, 16, hipMemcpyDeviceToHost);
    double err = 0;
    for (int i = 0; i < N * G::LD; ++i) err = fmax(err, fabs(h[i] - ref[i]));
    printf("%-34s %8.0f cycles per block   max |diff vs V0| %.2e  bad %lld\n", name, (double)hc[0] / REPS, err, hc[1]);
    hipFree(out); hipFree(cyc);
}

int main()
{
    using G = BlockGeom<N>;
    static double W[N * G::LD], ref[N * G::LD];
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) W[i * G::LD + j] = (i == j) ? 20.0 + i : 1.0 / (1.0 + abs(i - j));
    double* dW; hipMalloc(&dW, sizeof(W)); hipMemcpy(dW, W, sizeof(W), hipMemcpyHostToDevice);
    {   // V0 output as the reference
        double* out; long long* cyc; hipMalloc(&out, sizeof(W)); hipMalloc(&cyc, 16);
        bench<0><<<1, 64>>>(dW, out, cyc);
        hipMemcpy(ref, out, sizeof(ref), hipMemcpyDeviceToHost);
    }
    run<0>(dW, ref, "V0 current (readlane, row-wise inverse)");
    run<1>(dW, ref, "V1 column-oriented inverse");
    run<2>(dW, ref, "V2 V1 + LDS column exchange");
    return 0;
}
