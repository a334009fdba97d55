// Micro-benchmark: HBM read rate of a persistent grid (one 8-wave workgroup per CU) streaming two [M, 256] bf16 operands the
// way the resident-weights transform does -- a ring of DEPTH 4-load chunks per wave, refilled one chunk per step, waited for
// with a hand-placed s_waitcnt -- varying ONE factor at a time:
//   PATTERN 0 "fragment": lane (row = lane % 32, half = lane / 32) reads 16 bytes of ITS row per load (32 rows x 32 bytes / instr)
//   PATTERN 1 "lines":    lane reads piece lane % 8 of row lane / 8 (+ 8 per load): 8 whole 128-byte lines per instruction
//   PATTERN 2 "halves":   lane reads piece lane % 4 of a 64-byte half line of row lane / 4 (+ 16): 16 half lines per instruction
//   PATTERN 3 "mfma16":   the v_mfma_f32_16x16x32 operand shape: lane reads piece lane / 16 of a half line of row lane % 16 (+ 16)
//   DUP 1 / 2: every row block is read by DUP workgroups (on the same XCD), as the two column shares of the transform do
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(4))) unsigned int u4;

__device__ __forceinline__ void gload(u4& d, const void* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(p) : "memory"); }
template <int N> __device__ __forceinline__ void waitn(u4 (&r)[4]) {
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : "i"(N) : "memory");
}

template <int PATTERN, int DEPTH, int DUP>
__global__ __launch_bounds__(512) void rd(const char* __restrict__ x0, const char* __restrict__ x1, int64_t M, uint32_t* out) {
    extern __shared__ char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bid = blockIdx.x;
    const int pair = DUP == 1 ? bid : (bid >> 3) / DUP * 8 + (bid & 7);
    const int n_pairs = gridDim.x / DUP;
    const int64_t n_blocks = M / 256;
    // chunk sequence of this wave: block b, chunk c (0..7: 4 of x0 then 4 of x1), 128 bytes of 32 rows each
    auto addr = [&](int64_t blk, int c, int kk) -> const char* {
        const char* x = c < 4 ? x0 : x1;
        const int64_t r0 = blk * 256 + wave * 32;
        if (PATTERN == 0) return x + (r0 + (lane & 31)) * 512 + (c & 3) * 128 + kk * 32 + (lane >> 5) * 16;
        if (PATTERN == 1) return x + (r0 + (lane >> 3) + 8 * kk) * 512 + (c & 3) * 128 + (lane & 7) * 16;
        if (PATTERN == 2) return x + (r0 + (lane >> 2) + 16 * (kk & 1)) * 512 + (c & 3) * 128 + (kk >> 1) * 64 + (lane & 3) * 16;   // half lines
        return x + (r0 + (lane & 15) + 16 * (kk & 1)) * 512 + (c & 3) * 128 + (kk >> 1) * 64 + (lane >> 4) * 16;   // 16x16x32 MFMA operand
    };
    u4 A[DEPTH][4];
    uint32_t acc = 0;
    int64_t blk = pair;                      // position of the NEXT chunk to issue
    int c = 0;
    auto issue = [&](u4 (&slot)[4]) {
        const int64_t b = blk < n_blocks ? blk : pair;       // past the end: harmless re-reads
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) gload(slot[kk], addr(b, c, kk));
        if (++c == 8) { c = 0; blk += n_pairs; }
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) issue(A[d]);
    const int64_t my_blocks = (n_blocks - pair + n_pairs - 1) / n_pairs;
    const int64_t steps = my_blocks * 8;
    for (int64_t s = 0; s < steps; s += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            waitn<4 * (DEPTH - 1)>(A[d]);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) acc += A[d][kk].x ^ A[d][kk].w;
            issue(A[d]);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 0x12345678u) out[0] = acc + lds[0];
}

template <int PATTERN, int DEPTH, int DUP>
void run(const char* x0, const char* x1, int64_t M, uint32_t* out) {
    const size_t lds = 100 * 1024;                                   // one workgroup per CU
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rd<PATTERN, DEPTH, DUP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    rd<PATTERN, DEPTH, DUP><<<256, 512, lds>>>(x0, x1, M, out);
    (void)hipEventRecord(a);
    for (int i = 0; i < 5; ++i) rd<PATTERN, DEPTH, DUP><<<256, 512, lds>>>(x0, x1, M, out);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); ms /= 5;
    printf("  %-9s depth %d (%2d KB in flight per wave), every block read by %d workgroup(s): %.3f ms  %.2f TB/s unique, %.2f TB/s at L1\n",
           PATTERN == 3 ? "mfma16" : PATTERN == 2 ? "halves" : PATTERN ? "lines" : "fragment", DEPTH, DEPTH * 4, DUP, ms, (double)M * 1024 / 1e9 / ms, (double)M * 1024 * DUP / 1e9 / ms);
}

int main() {
    const int64_t M = 2449029 / 256 * 256;
    char *x0, *x1; uint32_t* out;
    (void)hipMalloc(&x0, M * 512); (void)hipMalloc(&x1, M * 512); (void)hipMalloc(&out, 4);
    (void)hipMemset(x0, 1, M * 512); (void)hipMemset(x1, 2, M * 512);
    run<3, 1, 1>(x0, x1, M, out); run<3, 2, 1>(x0, x1, M, out); run<3, 4, 1>(x0, x1, M, out);
    run<3, 1, 2>(x0, x1, M, out); run<3, 2, 2>(x0, x1, M, out); run<3, 4, 2>(x0, x1, M, out);
    run<2, 1, 2>(x0, x1, M, out); run<2, 2, 2>(x0, x1, M, out); run<0, 1, 2>(x0, x1, M, out); run<0, 2, 2>(x0, x1, M, out);
    return 0;
}
