// lone_wave_clock.hip -- what does the shader clock do while ONE wave runs on an otherwise idle chip?
// A dependent v_fma_f64 chain of fixed length, timed with s_memtime (shader clock) and s_memrealtime (constant 100 MHz):
// effective clock and cycles per dependent FMA for 1 wave, for 1 wave per CU, and for a full chip.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/lone_wave_clock.hip -o tools/ubench/bin/lone_wave_clock
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void chain(double* out, unsigned long long* t, long n)
{
    double x = 1.0 + threadIdx.x * 1e-9, a = 1.0000001, b = 1e-12;
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (long i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) x = __builtin_fma(x, a, b);
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
    if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = c1 - c0; t[1] = r1 - r0; }
}

int main()
{
    double* out; unsigned long long *t, h[2];
    hipMalloc(&out, sizeof(double) * 4096 * 256);
    hipMalloc(&t, 16);
    const long n = 200000;   // x 16 dependent FMAs
    for (int rep = 0; rep < 2; ++rep)
        for (int blocks : {1, 256, 4096}) {
            for (int threads : {64, 256}) {
                hipDeviceSynchronize();
                chain<<<blocks, threads>>>(out, t, n);
                hipDeviceSynchronize();
                hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
                const double secs = h[1] / 100e6;
                printf("%5d blocks x %3d threads: %.2f ms, s_memtime %.0f MHz-equivalent, %.1f s_memtime ticks and %.2f ns per dependent FMA\n",
                       blocks, threads, secs * 1e3, h[0] / secs / 1e6, (double)h[0] / (n * 16), secs * 1e9 / (n * 16));
            }
        }
    return 0;
}
