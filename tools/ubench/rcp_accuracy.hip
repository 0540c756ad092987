// rcp_accuracy.hip -- how far are common.h's fast_rcp / fast_rsqrt from the correctly rounded results?  (round 5: fast_rcp
// went from two Newton steps to one second-order step; this is its licence.)
//   hipcc --offload-arch=gfx950 -O3 -I include tools/ubench/rcp_accuracy.hip -o tools/ubench/bin/rcp_accuracy && tools/ubench/bin/rcp_accuracy
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include "../../diffqcqp_amd/csrc/common.h"

__global__ void k(const double* x, double* r, double* s, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { r[i] = dqq::fast_rcp(x[i]); s[i] = dqq::fast_rsqrt(x[i]); }
}
static int64_t ulps(double a, double b)
{
    int64_t ia, ib; std::memcpy(&ia, &a, 8); std::memcpy(&ib, &b, 8);
    return ia > ib ? ia - ib : ib - ia;
}
int main()
{
    const int n = 1 << 22;
    std::vector<double> x(n), r(n), s(n);
    uint64_t st = 88172645463325252ull;
    for (int i = 0; i < n; ++i) {
        st ^= st << 13; st ^= st >> 7; st ^= st << 17;
        const double u = (double)(st >> 11) / 9007199254740992.0;          // [0, 1)
        x[i] = std::ldexp(1.0 + u, (int)((st >> 3) % 120) - 60);            // 2^-60 .. 2^60, every mantissa
    }
    double *dx, *dr, *ds;
    hipMalloc(&dx, n * 8); hipMalloc(&dr, n * 8); hipMalloc(&ds, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, dr, ds, n);
    hipMemcpy(r.data(), dr, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(s.data(), ds, n * 8, hipMemcpyDeviceToHost);
    int64_t mr = 0, ms = 0; long exact_r = 0, exact_s = 0;
    for (int i = 0; i < n; ++i) {
        const int64_t a = ulps(r[i], 1.0 / x[i]), b = ulps(s[i], (double)(1.0L / sqrtl((long double)x[i])));
        mr = a > mr ? a : mr; ms = b > ms ? b : ms;
        exact_r += a == 0; exact_s += b == 0;
    }
    std::printf("fast_rcp:   max %lld ulp from the correctly rounded 1/x, %.4f %% exact, over %d values\n", (long long)mr, 100.0 * exact_r / n, n);
    std::printf("fast_rsqrt: max %lld ulp from the correctly rounded 1/sqrt(x), %.4f %% exact\n", (long long)ms, 100.0 * exact_s / n);
    return (mr <= 1 && ms <= 1) ? 0 : 1;
}
