// Semantics probe for v_permlane32_swap_b32 (gfx950): prints what each lane holds after the swap.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* a, int* b)
{
    int x = 100 + threadIdx.x, y = 200 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    a[threadIdx.x] = r[0];
    b[threadIdx.x] = r[1];
}
int main()
{
    int *a, *b, ha[64], hb[64];
    hipMalloc(&a, 256); hipMalloc(&b, 256);
    k<<<1, 64>>>(a, b);
    hipMemcpy(ha, a, 256, hipMemcpyDeviceToHost); hipMemcpy(hb, b, 256, hipMemcpyDeviceToHost);
    printf("r0: lane0=%d lane31=%d lane32=%d lane63=%d\n", ha[0], ha[31], ha[32], ha[63]);
    printf("r1: lane0=%d lane31=%d lane32=%d lane63=%d\n", hb[0], hb[31], hb[32], hb[63]);
    return 0;
}
