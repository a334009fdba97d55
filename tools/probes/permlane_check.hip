// What v_permlane32_swap_b32 does, empirically: a = lane, b = 100 + lane.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* p) {
    unsigned a = threadIdx.x, b = 100 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    p[threadIdx.x] = r[0]; p[threadIdx.x + 64] = r[1];
}
int main() {
    unsigned* d; unsigned h[128];
    (void)hipMalloc(&d, 512);
    k<<<1, 64>>>(d);
    (void)hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    printf("r[0]: lane0 %u lane1 %u lane31 %u lane32 %u lane63 %u\n", h[0], h[1], h[31], h[32], h[63]);
    printf("r[1]: lane0 %u lane1 %u lane31 %u lane32 %u lane63 %u\n", h[64], h[65], h[95], h[96], h[127]);
    return 0;
}
