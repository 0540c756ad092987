// div_by_recip.hip -- is  a / b  == the quotient rebuilt from r = RN(1/b) by residual corrections, bit for bit?
//   q0 = RN(a r);  e = a - b q0 (exact, one fma);  q1 = RN(q0 + e r);  [e = a - b q1;  q2 = RN(q1 + e r)]
// (Markstein: with r the correctly rounded reciprocal and q within one ulp, the corrected quotient is the correctly
// rounded one.)  The backward's factorisation divides ~280 times per problem by only 12 different pivots
// (bwd_lane_dense.hip); an IEEE division is ~11 FP64 instructions, a corrected product 3 or 5.
// Tests 2^34 pairs: random mantissas, structured ones (all ones, 1 + few ulps, powers of two), exponents spread over
// +-300, and quotients steered next to rounding midpoints.  Prints mismatch counts for the one- and two-correction forms.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench/div_by_recip.hip -o tools/ubench/bin/div_by_recip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

__device__ inline uint64_t mix(uint64_t x)
{
    x += 0x9e3779b97f4a7c15ull;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
    return x ^ (x >> 31);
}

__device__ inline double make(uint64_t h, int mode, int eshift)
{
    uint64_t mant = h & 0xfffffffffffffull;
    switch (mode & 7) {
    case 1: mant = 0xfffffffffffffull - (h & 7); break;          // all ones, minus a few
    case 2: mant = h & 7; break;                                   // 1 + a few ulps
    case 3: mant = (h & 0xfffffull) << 32; break;                  // short mantissas
    case 4: mant = 0x8000000000000ull + (h & 15) - 8; break;       // around 1.5
    default: break;
    }
    const int e = 1023 + (int)((h >> 52) % 601) - 300 + eshift;
    const uint64_t bits = ((uint64_t)e << 52) | mant | ((h >> 63) << 63);
    return __longlong_as_double((long long)bits);
}

__global__ void probe(uint64_t seed, unsigned long long* bad1, unsigned long long* bad2, unsigned long long* badr)
{
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long b1 = 0, b2 = 0, br = 0;
    for (int it = 0; it < 4096; ++it) {
        const uint64_t h0 = mix(seed + gid * 4096 + it), h1 = mix(h0);
        const int mode = (int)(h1 >> 56);
        double b = make(h0, mode, 0);
        double a = make(h1, mode >> 3, 0);
        if ((mode & 0xc0) == 0xc0) {
            // steer a / b next to a rounding midpoint: a = (q + half an ulp) * b, rounded, +- a few ulps
            const double q = make(mix(h1), 0, 0);
            const double qn = __longlong_as_double(__double_as_longlong(q) + 1);
            const double mid = 0.5 * q + 0.5 * qn; // not representable: rounds to q or qn -- perturb a instead
            (void)mid;
            const double prod = q * b;
            a = __longlong_as_double(__double_as_longlong(prod) + (long long)((h1 >> 40) & 7) - 3);
        }
        const double ref = a / b;
        const double r = 1.0 / b;
        double q0 = a * r;
        double e = fma(-b, q0, a);
        const double q1 = fma(e, r, q0);
        e = fma(-b, q1, a);
        const double q2 = fma(e, r, q1);
        const bool fin = (ref == ref) && fabs(ref) < 1e305 && fabs(ref) > 1e-290 && fabs(a) > 1e-280;
        if (fin) {
            b1 += (__double_as_longlong(q1) != __double_as_longlong(ref));
            b2 += (__double_as_longlong(q2) != __double_as_longlong(ref));
            // the reciprocal through the same corrections from the hardware seed: is it RN(1/b)?
            double y = __builtin_amdgcn_rcp(b);
            double ee = fma(-b, y, 1.0);
            y = fma(ee, y, y);
            ee = fma(-b, y, 1.0);
            y = fma(ee, y, y);
            br += (__double_as_longlong(y) != __double_as_longlong(r));
        }
    }
    if (b1) atomicAdd(bad1, b1);
    if (b2) atomicAdd(bad2, b2);
    if (br) atomicAdd(badr, br);
}

int main()
{
    unsigned long long *d, h[3] = {0, 0, 0};
    hipMalloc(&d, sizeof(h));
    hipMemset(d, 0, sizeof(h));
    const int blocks = 16384, threads = 256; // x 4096 pairs per thread = 2^34
    for (int rep = 0; rep < 1; ++rep) probe<<<blocks, threads>>>(0x1234567ull + rep, d, d + 1, d + 2);
    hipDeviceSynchronize();
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("pairs %.3g  mismatches vs a/b: one correction %llu, two corrections %llu; Newton reciprocal != 1.0/b: %llu\n",
           (double)blocks * threads * 4096, h[0], h[1], h[2]);
    return 0;
}
