// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the load patterns of the wave64 kernels (8 bytes per lane).
// The guide's "FETCH_SIZE reads half" note is for 16 B/lane streams.  Each kernel reads exactly `bytes` once from a
// buffer far larger than L2 + Infinity Cache; run under
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -o f -- tools/ubench/bin/fetch_calib
// and compare FETCH_SIZE (KiB) with the printed byte counts.
#include <cstdio>
#include <hip/hip_runtime.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// 16 B per lane, fully coalesced (the diagonal kernels' stream)
__global__ void read16(const double2* p, double* out, long n)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    double s = 0;
    for (; i < n; i += (long)gridDim.x * blockDim.x) { double2 v = p[i]; s += v.x + v.y; }
    if (s == 1.2345) out[0] = s;
}
// 8 B per lane, a wave instruction covers 512 contiguous bytes (tile-order loads, row-major tile rows)
__global__ void read8(const double* p, double* out, long n)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    double s = 0;
    for (; i < n; i += (long)gridDim.x * blockDim.x) s += p[i];
    if (s == 1.2345) out[0] = s;
}
// 8 B per lane, the "transposed tile" pattern of dense_wave64.hip: lane (g,n) reads A[16tj+n][16ti+4r+g] of a 64x64
// matrix: 16 rows x 32 contiguous bytes per wave instruction
__global__ void read8_tile_t(const double* p, double* out, long nmat)
{
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((long)gridDim.x * blockDim.x) >> 6;
    const unsigned lo = (lane & 15) * 64 + (lane >> 4);
    double s = 0;
    for (long m = wave; m < nmat; m += nw) {
        const double* A = p + m * 4096;
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
#pragma unroll
            for (int tj = 0; tj < 4; ++tj)
#pragma unroll
                for (int r = 0; r < 4; ++r) s += (A + ((16 * tj) * 64 + 16 * ti + 4 * r))[lo];
    }
    if (s == 1.2345) out[0] = s;
}
int main()
{
    const long bytes = 4L << 30; // 4 GiB >> 256 MiB Infinity Cache
    double *buf, *out;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, 64));
    CK(hipMemset(buf, 0, bytes));
    read16<<<4096, 256>>>((const double2*)buf, out, bytes / 16);
    read8<<<4096, 256>>>(buf, out, bytes / 8);
    read8_tile_t<<<4096, 256>>>(buf, out, bytes / 32768);
    CK(hipDeviceSynchronize());
    printf("each kernel read %ld bytes = %.0f KiB\n", bytes, bytes / 1024.0);
    return 0;
}
