// Micro-benchmark: cycles per ds_read_b128 / ds_write_b128 for the address patterns of the transform kernel's LDS tile
// (one wave, nothing else running): which lane groups the hardware serves together decides what "conflict-free" means.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(4))) unsigned int u4;

__global__ void k(const int* __restrict__ offs, int n_pat, int write, long long* out) {
    __shared__ __attribute__((aligned(16))) char lds[16384];
    const int lane = threadIdx.x;
    for (int i = lane; i < 4096; i += 64) reinterpret_cast<int*>(lds)[i] = i;
    __syncthreads();
    for (int p = 0; p < n_pat; ++p) {
        const int off = offs[p * 64 + lane];
        u4 acc = {0, 0, 0, 0};
        long long t0 = __builtin_readcyclecounter();
        for (int it = 0; it < 256; ++it) {
            if (write) { *reinterpret_cast<u4*>(lds + off) = acc; acc.x += it; }
            else { u4 v = *reinterpret_cast<const u4*>(lds + off); acc.x += v.x; acc.y ^= v.w; }
            asm volatile("" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        long long t1 = __builtin_readcyclecounter();
        if (lane == 0) out[p] = t1 - t0;
        if (acc.x == 0x12345678u) out[63] = acc.y;
    }
}

int main() {
    const char* names[] = {"linear lane*16", "fragment read, tile swizzle q^((r>>1)&3), e=0", "same, e=1", "tile write (row 16j+lane/4)",
                           "fragment read, no swizzle", "weights read slot^((r>>1)&7)", "fragment read, swizzle q^(r&3)",
                           "fragment read, swizzle q^((r>>2)&3)", "fragment read 128B rows, slot^(r&7)"};
    const int NP = 9;
    int h[NP * 64];
    for (int lane = 0; lane < 64; ++lane) {
        const int l32 = lane & 31, half = lane >> 5;
        h[0 * 64 + lane] = lane * 16;
        h[1 * 64 + lane] = l32 * 64 + (((0 + half) ^ ((l32 >> 1) & 3)) * 16);
        h[2 * 64 + lane] = l32 * 64 + (((2 + half) ^ ((l32 >> 1) & 3)) * 16);
        { const int r = lane >> 2; h[3 * 64 + lane] = r * 64 + (((lane & 3) ^ ((r >> 1) & 3)) * 16); }
        h[4 * 64 + lane] = l32 * 64 + (0 + half) * 16;
        h[5 * 64 + lane] = l32 * 128 + (((0 + half) ^ ((l32 >> 1) & 7)) * 16);
        h[6 * 64 + lane] = l32 * 64 + (((0 + half) ^ (l32 & 3)) * 16);
        h[7 * 64 + lane] = l32 * 64 + (((0 + half) ^ ((l32 >> 2) & 3)) * 16);
        h[8 * 64 + lane] = l32 * 128 + (((0 + half) ^ (l32 & 7)) * 16);
    }
    int* d; long long* out;
    (void)hipMalloc(&d, sizeof(h)); (void)hipMalloc(&out, 64 * 8);
    (void)hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    for (int write = 0; write < 2; ++write) {
        k<<<1, 64>>>(d, NP, write, out);
        k<<<1, 64>>>(d, NP, write, out);
        long long r[64];
        (void)hipMemcpy(r, out, 64 * 8, hipMemcpyDeviceToHost);
        for (int p = 0; p < NP; ++p) printf("%s b128 %-52s %.1f cycles per instruction\n", write ? "ds_write" : "ds_read ", names[p], r[p] / 256.0);
    }
    return 0;
}
