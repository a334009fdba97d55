// Micro-benchmark: aggregate read rate when the working set fits every XCD's 4 MiB L2 (all hits after the first pass), when it
// fits the 256 MiB Infinity Cache, and when it does not (HBM) -- the ceilings a gather kernel's L1 fills run against.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ __launch_bounds__(256) void rd(const uint4* __restrict__ x, int64_t n_vec, int reps, uint32_t* out) {
    uint32_t acc = 0;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int r = 0; r < reps; ++r) {
        // every block walks the WHOLE buffer (offset by its id so that neighbours do not read the same line at once)
        int64_t i = ((int64_t)blockIdx.x * 256 * 7 + threadIdx.x) % n_vec;
        for (int64_t k = 0; k < n_vec; k += 256 * 4) {
            uint4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { int64_t p = i + j * 256; if (p >= n_vec) p -= n_vec; v[j] = x[p]; }
#pragma unroll
            for (int j = 0; j < 4; ++j) acc += v[j].x ^ v[j].w;
            i += 256 * 4; if (i >= n_vec) i -= n_vec;
        }
    }
    (void)stride;
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    uint32_t* out; (void)hipMalloc(&out, 4);
    for (double mb : {1.0, 2.0, 3.0, 6.0, 16.0, 64.0, 192.0}) {
        const int64_t n_vec = (int64_t)(mb * 1024 * 1024) / 16 / 1024 * 1024;
        uint4* x; (void)hipMalloc(&x, n_vec * 16); (void)hipMemset(x, 1, n_vec * 16);
        const int blocks = 2048;
        int reps = (int)(2048.0 / mb / 2048 * 8); if (reps < 1) reps = 1;      // each block reads reps * mb MB
        reps = mb <= 16 ? (int)(64 / mb) : 1;
        hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        rd<<<blocks, 256>>>(x, n_vec, reps, out);
        (void)hipEventRecord(a);
        rd<<<blocks, 256>>>(x, n_vec, reps, out);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        const double bytes = (double)blocks * reps * n_vec * 16;
        printf("working set %6.1f MB, %d blocks x %d passes: %.3f ms, %.1f TB/s delivered to the CUs\n", mb, blocks, reps, ms, bytes / 1e9 / ms);
        (void)hipFree(x);
    }
    return 0;
}
