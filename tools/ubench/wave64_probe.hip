// Checks and times the register-resident 64x64 primitives of diffqcqp_amd/csrc/wave_tile.h on the device:
// the DPP row_newbcast mat-vec, the 16x16 sweep inverse and the block-sweep inverse on the matrix cores.
//   hipcc --offload-arch=gfx950 -O3 -I diffqcqp_amd/csrc -I include tools/ubench/wave64_probe.hip -o gpurun_out/wave64_probe
#include "wave_tile.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <hip/hip_runtime.h>
using namespace dqq;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// tile layout load of a symmetric matrix stored row-major (full)
DQQ_D void load_tiles(v4d (&G)[4][4], const double* S, int lane)
{
    const int g = lane >> 4, n = lane & 15;
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
        for (int tj = 0; tj < 4; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) G[ti][tj][r] = S[(16 * ti + 4 * r + g) * 64 + 16 * tj + n];
}
DQQ_D void store_tiles(const v4d (&G)[4][4], double* S, int lane)
{
    const int g = lane >> 4, n = lane & 15;
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
        for (int tj = 0; tj < 4; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) S[(16 * ti + 4 * r + g) * 64 + 16 * tj + n] = G[ti][tj][r];
}

__global__ __launch_bounds__(64) void k_matvec(const double* S, const double* x, double* y)
{
    const int lane = threadIdx.x;
    WaveTile64 W;
    load_tiles(W.G, S + (size_t)blockIdx.x * 4096, lane);
    const int xsrc = 4 * (lane & 15) + (lane >> 4);
    y[blockIdx.x * 64 + lane] = W.matvec(x[blockIdx.x * 64 + lane], xsrc);
}

__global__ __launch_bounds__(64) void k_diag16(const double* S, double* out)
{
    const int lane = threadIdx.x;
    const int g = lane >> 4, n = lane & 15;
    v4d T;
#pragma unroll
    for (int r = 0; r < 4; ++r) T[r] = S[(4 * r + g) * 16 + n];
    bool bad = false;
    const v4d D = diag16_inverse(T, lane, bad);
#pragma unroll
    for (int r = 0; r < 4; ++r) out[(4 * r + g) * 16 + n] = D[r];
    if (lane == 0) out[256] = bad ? 1.0 : 0.0;
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_inverse(const double* S, double* out, int reps)
{
    const int lane = threadIdx.x;
    WaveTile64 W;
    bool bad = false;
    for (int i = 0; i < reps; ++i) {
        load_tiles(W.G, S + (size_t)(blockIdx.x & 7) * 4096, lane);
        block_sweep_inverse(W.G, lane, bad);
    }
    store_tiles(W.G, out + (size_t)blockIdx.x * 4096, lane);
}

__global__ __launch_bounds__(64) void k_matvec_loop(const double* S, const double* x, double* y, int reps)
{
    const int lane = threadIdx.x;
    WaveTile64 W;
    load_tiles(W.G, S + (size_t)(blockIdx.x & 7) * 4096, lane);
    const int xsrc = 4 * (lane & 15) + (lane >> 4);
    double v = x[lane];
    for (int i = 0; i < reps; ++i) v = W.matvec(v, xsrc) * 0.01 + 0.5;
    y[blockIdx.x * 64 + lane] = v;
}

static void host_inverse(const std::vector<long double>& A, std::vector<long double>& inv, int n)
{
    std::vector<long double> M(A);
    inv.assign(n * n, 0.0L);
    for (int i = 0; i < n; ++i) inv[i * n + i] = 1.0L;
    for (int k = 0; k < n; ++k) {
        int p = k;
        for (int i = k + 1; i < n; ++i) if (fabsl(M[i * n + k]) > fabsl(M[p * n + k])) p = i;
        for (int j = 0; j < n; ++j) { std::swap(M[k * n + j], M[p * n + j]); std::swap(inv[k * n + j], inv[p * n + j]); }
        const long double d = 1.0L / M[k * n + k];
        for (int j = 0; j < n; ++j) { M[k * n + j] *= d; inv[k * n + j] *= d; }
        for (int i = 0; i < n; ++i) {
            if (i == k) continue;
            const long double f = M[i * n + k];
            for (int j = 0; j < n; ++j) { M[i * n + j] -= f * M[k * n + j]; inv[i * n + j] -= f * inv[k * n + j]; }
        }
    }
}

int main()
{
    const int NB = 8;
    std::vector<double> S(NB * 4096), x(NB * 64);
    srand(7);
    auto rnd = []() { return rand() / (double)RAND_MAX; };
    for (int b = 0; b < NB; ++b) {
        std::vector<double> R(4096);
        for (auto& v : R) v = rnd();
        const double shift = (b == 0) ? 0.3 : (b == 1 ? 1e-3 : (b == 2 ? 50.0 : 0.1 + rnd()));
        for (int i = 0; i < 64; ++i)
            for (int j = 0; j < 64; ++j) {
                double s = 0;
                for (int k = 0; k < 64; ++k) s += R[i * 64 + k] * R[j * 64 + k];
                S[b * 4096 + i * 64 + j] = s / 64 + (i == j ? shift : 0.0);
            }
        for (int i = 0; i < 64; ++i) x[b * 64 + i] = 2 * rnd() - 1;
    }
    double *dS, *dx, *dy, *dout;
    CK(hipMalloc(&dS, S.size() * 8)); CK(hipMalloc(&dx, x.size() * 8)); CK(hipMalloc(&dy, 4096 * 64 * 8));
    CK(hipMalloc(&dout, (size_t)4096 * 4096 * 8));
    CK(hipMemcpy(dS, S.data(), S.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dx, x.data(), x.size() * 8, hipMemcpyHostToDevice));

    // 1. mat-vec
    k_matvec<<<NB, 64>>>(dS, dx, dy);
    CK(hipDeviceSynchronize());
    std::vector<double> y(NB * 64);
    CK(hipMemcpy(y.data(), dy, y.size() * 8, hipMemcpyDeviceToHost));
    double e1 = 0;
    for (int b = 0; b < NB; ++b)
        for (int i = 0; i < 64; ++i) {
            long double s = 0;
            for (int j = 0; j < 64; ++j) s += (long double)S[b * 4096 + i * 64 + j] * x[b * 64 + j];
            e1 = fmax(e1, fabs((double)(s - y[b * 64 + i])) / fmax(1.0, fabs((double)s)));
        }
    printf("matvec          max rel err %.3e\n", e1);

    // 2. 16x16 sweep inverse
    {
        std::vector<double> B16(256);
        std::vector<long double> A(256), inv;
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) { B16[i * 16 + j] = S[i * 64 + j]; A[i * 16 + j] = B16[i * 16 + j]; }
        double* d16; CK(hipMalloc(&d16, 256 * 8));
        CK(hipMemcpy(d16, B16.data(), 256 * 8, hipMemcpyHostToDevice));
        k_diag16<<<1, 64>>>(d16, dout);
        CK(hipDeviceSynchronize());
        std::vector<double> o(257);
        CK(hipMemcpy(o.data(), dout, 257 * 8, hipMemcpyDeviceToHost));
        host_inverse(A, inv, 16);
        double e = 0, m = 0;
        for (int i = 0; i < 256; ++i) { e = fmax(e, fabs(o[i] - (double)inv[i])); m = fmax(m, fabs((double)inv[i])); }
        printf("diag16 inverse  max abs err %.3e (max |inv| %.3e) bad=%g\n", e, m, o[256]);
    }

    // 3. block-sweep inverse (result = -S^-1)
    k_inverse<<<NB, 64>>>(dS, dout, 1);
    CK(hipDeviceSynchronize());
    std::vector<double> o(NB * 4096);
    CK(hipMemcpy(o.data(), dout, o.size() * 8, hipMemcpyDeviceToHost));
    for (int b = 0; b < NB; ++b) {
        std::vector<long double> A(4096), inv;
        for (int i = 0; i < 4096; ++i) A[i] = S[b * 4096 + i];
        host_inverse(A, inv, 64);
        double e = 0, m = 0, asym = 0;
        for (int i = 0; i < 64; ++i)
            for (int j = 0; j < 64; ++j) {
                e = fmax(e, fabs(-o[b * 4096 + i * 64 + j] - (double)inv[i * 64 + j]));
                m = fmax(m, fabs((double)inv[i * 64 + j]));
                asym = fmax(asym, fabs(o[b * 4096 + i * 64 + j] - o[b * 4096 + j * 64 + i]));
            }
        printf("inverse[%d]      max abs err %.3e  rel %.3e  asym %.3e\n", b, e, e / m, asym);
    }

    // 4. timing: 8 waves per CU
    hipEvent_t t0, t1;
    CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    for (int wpc : {4, 8, 12}) {
        const int grid = 256 * wpc;
        float ms;
        k_matvec_loop<<<grid, 64>>>(dS, dx, dy, 10);
        CK(hipEventRecord(t0));
        k_matvec_loop<<<grid, 64>>>(dS, dx, dy, 2000);
        CK(hipEventRecord(t1)); CK(hipEventSynchronize(t1)); CK(hipEventElapsedTime(&ms, t0, t1));
        printf("matvec loop  %2d waves/CU: %.1f ns per mat-vec per wave, %.2f ns per mat-vec per CU\n", wpc,
               ms * 1e6 / 2000, ms * 1e6 / 2000 / wpc);
        k_inverse<<<grid, 64>>>(dS, dout, 1);
        CK(hipEventRecord(t0));
        k_inverse<<<grid, 64>>>(dS, dout, 50);
        CK(hipEventRecord(t1)); CK(hipEventSynchronize(t1)); CK(hipEventElapsedTime(&ms, t0, t1));
        printf("inverse loop %2d waves/CU: %.2f us per inverse per wave, %.3f us per inverse per CU\n", wpc,
               ms * 1e3 / 50, ms * 1e3 / 50 / wpc);
    }
    return 0;
}
