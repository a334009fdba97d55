// Micro-benchmark: HBM read rate of a tall [M, 256] bf16 matrix under the two access patterns an MFMA transform can use
// for its activation operand.  Pattern F ("fragment-shaped"): lane (row = lane % 32, half = lane / 32) reads 16 bytes at
// column (2*kk + half) * 16 of ITS row -- a wave instruction touches 32 rows, 32 bytes of each.  Pattern L ("full lines"):
// lane reads piece lane % 8 of row lane / 8 (+ 8 j) -- a wave instruction covers 8 whole 128-byte lines.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/read_pattern.hip -o /tmp/read_pattern && /tmp/read_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int PATTERN, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void read_kernel(const uint4* __restrict__ x, int64_t M, int row_vecs, uint32_t* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t m0 = ((int64_t)blockIdx.x * WAVES + wave) * 32;
    if (m0 + 32 > M) return;
    uint32_t acc = 0;
    for (int c = 0; c < row_vecs / 8; ++c) {          // 128-byte chunks of the row
        uint4 v[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            int64_t row; int piece;
            if (PATTERN == 0) { row = m0 + (lane & 31); piece = 2 * kk + (lane >> 5); }
            else { row = m0 + (lane >> 3) + 8 * kk; piece = lane & 7; }
            v[kk] = x[row * row_vecs + c * 8 + piece];
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) acc += v[kk].x ^ v[kk].y ^ v[kk].z ^ v[kk].w;
    }
    if (acc == 0x12345678u) out[0] = acc;             // keep the loads alive
}

template <int PATTERN, int WAVES>
float run(const uint4* x, int64_t M, int row_vecs, uint32_t* out) {
    dim3 grid((unsigned)((M / 32 + WAVES - 1) / WAVES));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    read_kernel<PATTERN, WAVES><<<grid, 64 * WAVES>>>(x, M, row_vecs, out);
    hipEventRecord(a);
    for (int i = 0; i < 5; ++i) read_kernel<PATTERN, WAVES><<<grid, 64 * WAVES>>>(x, M, row_vecs, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / 5;
}

int main() {
    const int64_t M = 2449029 / 32 * 32;
    for (int row_vecs : {32, 16}) {                   // 512-byte and 256-byte rows
        uint4* x; uint32_t* out;
        hipMalloc(&x, M * row_vecs * 16); hipMalloc(&out, 4);
        hipMemset(x, 1, M * row_vecs * 16);
        const double gb = (double)M * row_vecs * 16 / 1e9;
        printf("rows of %d bytes, %.2f GB\n", row_vecs * 16, gb);
        printf("  fragment-shaped, 4 waves/block: %.3f ms  %.2f TB/s\n", run<0, 4>(x, M, row_vecs, out), gb / run<0, 4>(x, M, row_vecs, out));
        printf("  full lines,      4 waves/block: %.3f ms  %.2f TB/s\n", run<1, 4>(x, M, row_vecs, out), gb / run<1, 4>(x, M, row_vecs, out));
        printf("  fragment-shaped, 8 waves/block: %.3f ms  %.2f TB/s\n", run<0, 8>(x, M, row_vecs, out), gb / run<0, 8>(x, M, row_vecs, out));
        printf("  full lines,      8 waves/block: %.3f ms  %.2f TB/s\n", run<1, 8>(x, M, row_vecs, out), gb / run<1, 8>(x, M, row_vecs, out));
        hipFree(x); hipFree(out);
    }
    return 0;
}
