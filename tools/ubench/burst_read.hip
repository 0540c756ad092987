// How fast can 2048 resident waves pull a 33.5 MB tile stream (the P of a B = 65536, N = 8 forward) when every wave
// asks for its 16 KiB at once?  Warm (the buffers of a bench step, 134 MB in all, live in the Infinity Cache) and
// cold (rotating over > 256 MiB).  Compare with the 5.8 us the forward kernel's waves wait for P
// (tools/probe_timeline.py).
#include <cstdio>
#include <hip/hip_runtime.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int NLOAD>
__global__ __launch_bounds__(256) void burst(const double2* __restrict__ p, double* out)
{
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const double2* w = p + wave * (NLOAD * 64) + lane;
    double2 v[NLOAD];
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) v[i] = w[i * 64];
    double s = 0;
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) s += v[i].x + v[i].y;
    if (s == 1.2345) out[0] = s;
}
__global__ void empty_kernel(double* out) { if (threadIdx.x == 12345) out[0] = 1; }

int main()
{
    const long tile_bytes = 16384, waves = 2048, bytes = tile_bytes * waves; // 33.5 MB
    const int NBUF = 12;                                                    // 403 MB in all
    double2* buf[NBUF]; double* out;
    for (int i = 0; i < NBUF; ++i) { CK(hipMalloc(&buf[i], bytes)); CK(hipMemset(buf[i], 0, bytes)); }
    CK(hipMalloc(&out, 64));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto run = [&](int nbuf, int which, const char* name) {
        for (int r = 0; r < 50; ++r) burst<16><<<waves / 4, 256>>>(buf[r % nbuf], out);
        hipEventRecord(a);
        const int R = 400;
        for (int r = 0; r < R; ++r) {
            if (which == 0) empty_kernel<<<waves / 4, 256>>>(out);
            else burst<16><<<waves / 4, 256>>>(buf[r % nbuf], out);
        }
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%-44s %.2f us per launch", name, ms * 1e3 / R);
        if (which) printf("  (%.2f TB/s incl. launch)", bytes / (ms * 1e-3 / R) * 1e-12);
        printf("\n");
    };
    run(1, 0, "empty kernel, 512 x 256");
    run(1, 1, "one 33.5 MB buffer (L2 / Infinity Cache)");
    run(4, 1, "4 buffers, 134 MB (Infinity Cache)");
    run(NBUF, 1, "12 buffers, 403 MB (HBM)");
    return 0;
}
