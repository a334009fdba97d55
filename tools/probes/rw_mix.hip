// Micro-benchmark: what a plain streaming kernel reaches on this part for the transform's traffic MIX -- two [M, 256] bf16
// operands read, one [M, 256] bf16 result written (2R : 1W), next to 1R : 1W (copy) and read-only, full-line accesses,
// non-persistent grid, no arithmetic to speak of.  The fused transform's ceiling is the 2R:1W line, not the 8 TB/s pin rate.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/rw_mix.hip -o tools/probes/rw_mix.bin && tools/probes/rw_mix.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int MODE, int NT>
__global__ __launch_bounds__(256) void k(const uint4* __restrict__ a, const uint4* __restrict__ b, uint4* __restrict__ c, int64_t n, uint32_t* out) {
    const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x);
    const int64_t stride = (int64_t)gridDim.x * 256;
    uint32_t acc = 0;
#pragma unroll 4
    for (int64_t i = i0; i < n; i += stride) {
        uint4 va = a[i];
        if (MODE == 2) { uint4 vb = b[i]; va.x ^= vb.x; va.y += vb.y; va.z ^= vb.z; va.w += vb.w; }
        if (MODE == 0) acc += va.x ^ va.w;
        else if (NT) { typedef unsigned int u4 __attribute__((ext_vector_type(4))); u4 t = {va.x, va.y, va.z, va.w}; __builtin_nontemporal_store(t, (u4*)&c[i]); }
        else c[i] = va;
    }
    if (MODE == 0 && acc == 0x12345678u) out[0] = acc;
}

// persistent grid, tiles of 4 KB per workgroup step handed out by an atomic counter (DYN = 1) or statically strided (DYN = 0),
// TILE = consecutive 4 KB pieces per grab
template <int DYN, int TILE>
__global__ __launch_bounds__(256) void kp(const uint4* __restrict__ a, const uint4* __restrict__ b, uint4* __restrict__ c, int64_t n,
                                          unsigned int* counter) {
    __shared__ unsigned int s_t;
    const int64_t n_tiles = n / (256 * TILE);
    unsigned int t = blockIdx.x;
    for (;;) {
        if (DYN) {
            if (threadIdx.x == 0) s_t = atomicAdd(counter, 1u);
            __syncthreads();
            t = s_t;
            __syncthreads();
        }
        if (t >= n_tiles) break;
        const int64_t i0 = (int64_t)t * 256 * TILE + threadIdx.x;
        uint4 va[TILE], vb[TILE];
#pragma unroll
        for (int j = 0; j < TILE; ++j) { va[j] = a[i0 + j * 256]; vb[j] = b[i0 + j * 256]; }
#pragma unroll
        for (int j = 0; j < TILE; ++j) {
            va[j].x ^= vb[j].x; va[j].y += vb[j].y; va[j].z ^= vb[j].z; va[j].w += vb[j].w;
            c[i0 + j * 256] = va[j];
        }
        if (!DYN) t += gridDim.x;
    }
}

template <int DYN, int TILE>
void runp(const char* name, const uint4* a, const uint4* b, uint4* c, int64_t n, unsigned int* counter, int blocks, double bytes) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipMemsetAsync(counter, 0, 4); kp<DYN, TILE><<<blocks, 256>>>(a, b, c, n, counter);
    float tot = 0;
    for (int i = 0; i < 5; ++i) {
        hipMemsetAsync(counter, 0, 4);
        hipEventRecord(e0);
        kp<DYN, TILE><<<blocks, 256>>>(a, b, c, n, counter);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); tot += ms;
    }
    printf("  %-44s grid %7d: %.3f ms  %.2f TB/s\n", name, blocks, tot / 5, bytes / 1e9 / (tot / 5));
}

template <int MODE, int NT>
void run(const char* name, const uint4* a, const uint4* b, uint4* c, int64_t n, uint32_t* out, int blocks, double bytes) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, NT><<<blocks, 256>>>(a, b, c, n, out);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) k<MODE, NT><<<blocks, 256>>>(a, b, c, n, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("  %-34s grid %7d: %.3f ms  %.2f TB/s\n", name, blocks, ms, bytes / 1e9 / ms);
}

int main() {
    const int64_t M = 2449029 / 32 * 32, n = M * 32;          // uint4 elements per operand
    uint4 *a, *b, *c; uint32_t* out;
    hipMalloc(&a, n * 16); hipMalloc(&b, n * 16); hipMalloc(&c, n * 16); hipMalloc(&out, 4);
    hipMemset(a, 1, n * 16); hipMemset(b, 2, n * 16);
    const double gb = (double)n * 16;
    printf("operands of %.2f GB\n", gb / 1e9);
    for (int blocks : {256 * 8, 256 * 32, (int)((n + 255) / 256 / 4)}) {
        run<0, 0>("read only (1R)", a, b, c, n, out, blocks, gb);
        run<1, 0>("copy (1R:1W)", a, b, c, n, out, blocks, 2 * gb);
        run<1, 1>("copy (1R:1W), streaming stores", a, b, c, n, out, blocks, 2 * gb);
        run<2, 0>("add (2R:1W)", a, b, c, n, out, blocks, 3 * gb);
        run<2, 1>("add (2R:1W), streaming stores", a, b, c, n, out, blocks, 3 * gb);
    }
    unsigned int* counter; hipMalloc(&counter, 4);
    for (int blocks : {256 * 2, 256 * 8}) {
        runp<0, 1>("persistent add, static stride, 4 KB", a, b, c, n, counter, blocks, 3 * gb);
        runp<1, 1>("persistent add, atomic hand-out, 4 KB", a, b, c, n, counter, blocks, 3 * gb);
        runp<0, 4>("persistent add, static stride, 16 KB", a, b, c, n, counter, blocks, 3 * gb);
        runp<1, 4>("persistent add, atomic hand-out, 16 KB", a, b, c, n, counter, blocks, 3 * gb);
    }
    return 0;
}
