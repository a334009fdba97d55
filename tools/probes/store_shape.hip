// Micro-benchmark: write rate of the transform's output pattern -- a persistent grid, every wave owns 32 rows x 256 bytes of
// a [M, 512-byte] matrix per step -- for three store shapes:
//   S64:  16 rows x  64 B per instruction (what the epilogue's 32 x 32 LDS tile gives)
//   S128:  8 rows x 128 B per instruction (would need a 64-column staging tile)
//   S32:  32 rows x  32 B per instruction (two 16-byte pieces per row: the permlane variant)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(4))) unsigned int u4;

template <int SHAPE>
__global__ __launch_bounds__(512) void wr(char* __restrict__ y, int64_t M) {
    extern __shared__ char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int share = (blockIdx.x >> 3) & 1, pair = (blockIdx.x >> 4) * 8 + (blockIdx.x & 7), n_pairs = gridDim.x / 2;
    const int64_t n_blocks = M / 256;
    u4 v = {(unsigned)lane, 1u, 2u, 3u};
    for (int64_t blk = pair; blk < n_blocks; blk += n_pairs) {
        char* base = y + (blk * 256 + wave * 32) * 512 + share * 256;
        if (SHAPE == 64) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int idx = it * 64 + lane;
                    *reinterpret_cast<u4*>(base + (idx >> 2) * 512 + t * 64 + (idx & 3) * 16) = v;
                }
        } else if (SHAPE == 128) {
#pragma unroll
            for (int tp = 0; tp < 2; ++tp)
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int idx = it * 64 + lane;
                    *reinterpret_cast<u4*>(base + (idx >> 3) * 512 + tp * 128 + (idx & 7) * 16) = v;
                }
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int it = 0; it < 2; ++it)
                    *reinterpret_cast<u4*>(base + (lane & 31) * 512 + t * 64 + (lane >> 5) * 32 + it * 16) = v;
        }
        v.x += 1;
    }
    if (v.y == 0x12345678u) y[0] = lds[0];
}

template <int SHAPE>
void run(const char* name, char* y, int64_t M) {
    const size_t lds = 100 * 1024;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wr<SHAPE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    wr<SHAPE><<<256, 512, lds>>>(y, M);
    (void)hipEventRecord(a);
    for (int i = 0; i < 5; ++i) wr<SHAPE><<<256, 512, lds>>>(y, M);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); ms /= 5;
    printf("  %-40s %.3f ms  %.2f TB/s\n", name, ms, (double)M * 512 / 1e9 / ms);
}

int main() {
    const int64_t M = 2449029 / 256 * 256;
    char* y; (void)hipMalloc(&y, M * 512);
    printf("write-only, [%lld, 512 B], persistent grid of 256 x 8 waves\n", (long long)M);
    run<64>("16 rows x 64 B per instruction", y, M);
    run<128>("8 rows x 128 B per instruction", y, M);
    run<32>("32 rows x 2 x 16 B per instruction", y, M);
    return 0;
}
