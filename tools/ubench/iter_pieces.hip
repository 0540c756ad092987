// Latency of the pieces of one ADMM iteration of the workgroup dense kernel (dense_block.hip), each
// timed as a dependent chain of REPS repetitions by one 256-thread workgroup.
//   hipcc --offload-arch=gfx950 -O3 -I../../diffqcqp_amd/csrc -I../../include iter_pieces.hip -o iter_pieces
#include "common.h"
#include <cstdio>
#include <hip/hip_runtime.h>
using namespace dqq;

DQQ_D double lane_bcast(double v, int src)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

constexpr int REPS = 2000;

template <int PIECE>
__global__ __launch_bounds__(256) void piece(double* out, long long* cycles, double seed)
{
    __shared__ __attribute__((aligned(16))) double lds[2048];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    double* strip = lds + 1024 + 16 * wave;
    double m[16];
    for (int k = 0; k < 16; ++k) m[k] = seed * (k + 1) * 1e-3;
    double v = seed + lane * 1e-3, acc = 0.0;
    int parity = 0;
    __syncthreads();
    const long long t0 = clock64();
    for (int rep = 0; rep < REPS; ++rep) {
        if (PIECE == 0) { // strip write + fence + 8 x ds_read_b128 + 16 FMA
            if ((unsigned)(lane - 16 * wave) < 16u) strip[lane - 16 * wave] = v;
            wave_lds_fence();
            double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
            for (int k = 0; k < 16; k += 4) {
                a0 = fma(m[k], strip[k], a0); a1 = fma(m[k + 1], strip[k + 1], a1);
                a2 = fma(m[k + 2], strip[k + 2], a2); a3 = fma(m[k + 3], strip[k + 3], a3);
            }
            v = (a0 + a1) + (a2 + a3);
        } else if (PIECE == 1) { // partial write + barrier + 4 reads + 3 adds
            double* buf = lds + parity * 256; parity ^= 1;
            buf[wave * 64 + lane] = v;
            __syncthreads();
            v = ((buf[lane] + buf[64 + lane]) + buf[128 + lane]) + buf[192 + lane];
        } else if (PIECE == 2) { // element-wise update (QP)
            const double l = v;
            double z = 1.5 * l + -0.5 * acc + seed * 0.25;
            z = z < 0 ? 0 : z;
            const double rd = fabs(seed * (z - acc)), rp = fabs(z - (1.5 * l - 0.5 * acc));
            acc = z;
            v = rd + rp;
        } else if (PIECE == 3) { // max2: permlane swap + 4 DPP steps + readlanes
            const double a = v, b = v * 0.5;
            const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
            const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
            const double x = fmax(__hiloint2double(hi[0], lo[0]), __hiloint2double(hi[1], lo[1]));
            const double mm = LaneGroup<16>::max(x);
            const double ma = fmax(lane_bcast(mm, 0), lane_bcast(mm, 16));
            const double mb = fmax(lane_bcast(mm, 32), lane_bcast(mm, 48));
            v = ma * 0.999 + mb * 1e-3 + lane * 1e-9;
        } else if (PIECE == 4) { // barrier only
            __syncthreads();
            v = v * 1.0000001;
        } else if (PIECE == 5) { // LDS write -> fence -> read round trip (wave-private)
            strip[lane & 15] = v;
            wave_lds_fence();
            v = strip[(lane + 1) & 15] * 1.0000001;
        } else if (PIECE == 6) { // one DPP max step
            v = fmax(v, partner<1>(v)) * 1.0000001;
        } else if (PIECE == 7) { // readlane -> VALU
            v = lane_bcast(v, 5) * 1.0000001 + lane * 1e-9;
        }
    }
    const long long t1 = clock64();
    out[blockIdx.x * 256 + t] = v + acc;
    if (t == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int PIECE>
void run(const char* name)
{
    double* out; long long* cyc;
    hipMalloc(&out, 256 * 8); hipMalloc(&cyc, 8);
    piece<PIECE><<<1, 256>>>(out, cyc, 1.25);
    piece<PIECE><<<1, 256>>>(out, cyc, 1.25);
    long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-50s %7.1f cycles/rep\n", name, (double)h / REPS);
    hipFree(out); hipFree(cyc);
}

int main()
{
    run<0>("rhs strip write+fence+8 b128 reads+16 FMA");
    run<1>("partial write + barrier + 4 reads + 3 adds");
    run<2>("element-wise update");
    run<3>("max2 (permlane swap, 4 DPP steps, readlanes)");
    run<4>("barrier only (+1 mul)");
    run<5>("wave-private LDS write->read (+1 mul)");
    run<6>("one DPP max step (+1 mul)");
    run<7>("readlane -> VALU (+fma)");
    return 0;
}
