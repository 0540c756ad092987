// Microbenchmark (developer tool): fp64 VALU issue rate of ONE wave per SIMD vs several, with
// NCHAIN independent dependency chains per lane.  Prints cycles per v_fma_f64 per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int NCHAIN>
__global__ void k(double* out, int iters, double a, double b)
{
    double v[NCHAIN];
#pragma unroll
    for (int i = 0; i < NCHAIN; ++i) v[i] = threadIdx.x * 1e-3 + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < NCHAIN; ++i) v[i] = __builtin_fma(v[i], a, b);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NCHAIN; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NCHAIN>
void run(int waves_per_simd)
{
    const int iters = 2000;
    const int blocks = 256 * waves_per_simd; // 256-thread blocks: one wave on each SIMD of a CU
    double* out;
    (void)hipMalloc(&out, sizeof(double) * blocks * 256);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<NCHAIN><<<blocks, 256>>>(out, 10, 1.0000001, 1e-9);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<NCHAIN><<<blocks, 256>>>(out, iters, 1.0000001, 1e-9);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double fmas_per_wave = (double)iters * 8 * NCHAIN;
    const double ns_per_fma_per_simd = ms * 1e6 / (fmas_per_wave * waves_per_simd);
    printf("chains %2d waves/SIMD %d: %.3f ms, %.2f ns per wave-FMA per SIMD (= %.2f cycles at 2.1 GHz, %.2f at 2.4)\n", NCHAIN,
           waves_per_simd, ms, ns_per_fma_per_simd, ns_per_fma_per_simd * 2.1, ns_per_fma_per_simd * 2.4);
    (void)hipFree(out);
}

int main()
{
    for (int w : {1, 2, 4}) {
        run<1>(w); run<2>(w); run<4>(w); run<8>(w);
    }
    return 0;
}
