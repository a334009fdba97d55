// Probe for VERDICT round 4, item 4: would SOURCE-COLUMN BLOCKING raise the L2 hit rate of the F = 256 gather?
//
// The shipped SpMM walks a row's edges in CSR order: every row sweeps its whole community (~38 k source rows = 19 MB at 512 B,
// five times one XCD's 4 MB L2), so only the hub rows stay resident (hit rate 0.54).  The blocked form would let a workgroup own
// R consecutive output rows (accumulators in LDS) and walk the edges of ALL of them in ascending SOURCE order: the workgroups of an
// XCD, started together on neighbouring row groups, then sweep the same few MB of source rows at the same time.
//
// This kernel measures only what that order does to the GATHER (the upper bound of the gain; the LDS accumulation is not paid):
// workgroup g gathers the 512-byte rows idx[ptr[g] .. ptr[g+1]) -- 32 lanes x 16 B per row, two row slots per wavefront, U = 4
// loads in flight per lane, indices from coalesced 64-entry batches, the four wavefronts taking alternate batches -- and writes R
// rows at the end (the Y write).  tools/colblock_probe.py feeds it the bench graph's edges in CSR order and in per-group
// source-sorted order and times both (and reads FETCH_SIZE under rocprofv3 --pmc).
#include <hip/hip_runtime.h>
#include <stdint.h>

__device__ __forceinline__ float lo(uint32_t p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float hi(uint32_t p) { return __uint_as_float(p & 0xffff0000u); }

__device__ __forceinline__ uint32_t xcd_remap(uint32_t bid, uint32_t nblocks) {      // XCD x gets a contiguous range of groups
    const uint32_t q = nblocks / 8, r = nblocks % 8;
    const uint32_t xcd = bid % 8, i = bid / 8;
    return xcd * q + (xcd < r ? xcd : r) + i;
}

template <int U>
__global__ __launch_bounds__(256) void gather_groups(const uint4* __restrict__ table, const int32_t* __restrict__ idx,
                                                     const int64_t* __restrict__ ptr, int n_groups, int rows_per_group,
                                                     uint4* __restrict__ out, int remap) {
    const int lane = threadIdx.x & 63, slot = lane >> 5, l32 = lane & 31, wave = threadIdx.x >> 6;
    const uint32_t g = remap ? xcd_remap(blockIdx.x, (uint32_t)n_groups) : blockIdx.x;
    if (g >= (uint32_t)n_groups) return;
    const int64_t e0 = ptr[g], e1 = ptr[g + 1];
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int64_t base = e0 + 64 * wave; base < e1; base += 256) {
        const int cnt = (int)((e1 - base) < 64 ? (e1 - base) : 64);
        const int mine = __builtin_nontemporal_load(idx + base + (lane < cnt ? lane : cnt - 1));
#pragma unroll 1
        for (int k = 0; k < cnt; k += 2 * U) {
            uint4 v[U];
            bool ok[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = k + 2 * u + slot;
                ok[u] = j < cnt;
                const int id = __shfl(mine, ok[u] ? j : cnt - 1);
                v[u] = table[(int64_t)id * 32 + l32];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float m = ok[u] ? 1.0f : 0.0f;
                acc[0] += m * lo(v[u].x); acc[1] += m * hi(v[u].x); acc[2] += m * lo(v[u].y); acc[3] += m * hi(v[u].y);
                acc[4] += m * lo(v[u].z); acc[5] += m * hi(v[u].z); acc[6] += m * lo(v[u].w); acc[7] += m * hi(v[u].w);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] += __shfl_xor(acc[i], 32);
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 ov = {(__float_as_uint(acc[0]) >> 16) | (__float_as_uint(acc[1]) & 0xffff0000u),
                (__float_as_uint(acc[2]) >> 16) | (__float_as_uint(acc[3]) & 0xffff0000u),
                (__float_as_uint(acc[4]) >> 16) | (__float_as_uint(acc[5]) & 0xffff0000u),
                (__float_as_uint(acc[6]) >> 16) | (__float_as_uint(acc[7]) & 0xffff0000u)};
    for (int r = wave * 2 + slot; r < rows_per_group; r += 8)      // the group's R output rows, written once
        __builtin_nontemporal_store(ov, reinterpret_cast<u32x4*>(out + ((int64_t)g * rows_per_group + r) * 32 + l32));
}

extern "C" int colblock_gather(void* stream, const void* table, const int32_t* idx, const int64_t* ptr, int n_groups, int rows_per_group,
                               void* out, int remap, int u) {
    const dim3 grid((uint32_t)n_groups);
    if (u == 8)
        hipLaunchKernelGGL((gather_groups<8>), grid, dim3(256), 0, (hipStream_t)stream, (const uint4*)table, idx, ptr, n_groups, rows_per_group,
                           (uint4*)out, remap);
    else
        hipLaunchKernelGGL((gather_groups<4>), grid, dim3(256), 0, (hipStream_t)stream, (const uint4*)table, idx, ptr, n_groups, rows_per_group,
                           (uint4*)out, remap);
    return (int)hipGetLastError();
}
