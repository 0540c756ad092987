// Microbenchmark (developer tool): do v_mfma_f64_16x16x4_f64 and v_fma_f64 of TWO DIFFERENT waves on one SIMD overlap
// on MI355X?  512-thread workgroups (8 waves: waves w and w + 4 share a SIMD), one workgroup per CU.
//   mode 0: all 8 waves run the FMA loop          mode 1: all 8 waves run the MFMA loop
//   mode 2: waves 0-3 FMA, waves 4-7 MFMA (one of each per SIMD)
//   mode 3: waves 0-3 FMA, waves 4-7 idle           mode 4: waves 0-3 idle, waves 4-7 MFMA
// If the f64 matrix pipe were separate from the f64 vector ALUs, mode 2 would take max(mode 3, mode 4); if they
// share the ALUs it takes their sum.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double v4d __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void k(double* out, int iters, int mode, double a, double b)
{
    const int wave = threadIdx.x >> 6;
    const bool do_fma = mode == 0 || ((mode == 2 || mode == 3) && wave < 4);
    const bool do_mfma = mode == 1 || ((mode == 2 || mode == 4) && wave >= 4);
    double s = 0;
    if (do_fma) {
        double v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3 + i;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int r = 0; r < 8; ++r)          // 64 FMAs per trip
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = __builtin_fma(v[i], a, b);
#pragma unroll
        for (int i = 0; i < 8; ++i) s += v[i];
    }
    if (do_mfma) {
        v4d acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = v4d{0, 0, 0, 0};
        const double x = threadIdx.x * 1e-3, y = 1e-3;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r)          // 16 MFMAs per trip (4 independent accumulators)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main()
{
    double* out;
    (void)hipMalloc(&out, sizeof(double) * 256 * 512);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 4000;
    for (int mode = 0; mode < 5; ++mode) {
        k<<<256, 512>>>(out, 10, mode, 1.0000001, 1e-9);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        k<<<256, 512>>>(out, iters, mode, 1.0000001, 1e-9);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("mode %d: %.3f ms  (per trip: %.1f ns = %.0f cycles at 2.4 GHz; a trip = 64 wave-FMAs and/or 16 MFMAs per wave)\n",
               mode, ms, ms * 1e6 / iters, ms * 1e6 / iters * 2.4);
    }
    return 0;
}
