// gather.hip -- feature-row gather for the FeatureCache / mini-batch queues and the halo packer.
//
//   out[i, :] = cache[slot[idx[i]], :]   if slot[idx[i]] >= 0      (hot node: row lives in the HBM cache)
//             = host[idx[i], :]           otherwise                  (miss: read straight from pinned host memory)
//
// Reference: GraphCacheServer.fetch_data, /root/reference/dgll/FeatureCache/storage.py:151-198 -- mask the batch into
// cached / uncached ids, gather the cached rows on the GPU (:176-181), gather the rest on the CPU and copy them over
// (:183-188), merge in place; and `self.features[nodes]`, dgll/data/dgraph.py:105.  Here the split, both gathers and the
// merge are ONE kernel: the miss rows are pulled over PCIe by the GPU itself from the pinned (device-mapped) host
// array, 16 bytes per lane, while the hit rows come from HBM; the per-batch miss count is accumulated for the
// miss-rate log (storage.py:213-220).  HBM/PCIe-bound, no LDS, no MFMA.
#include <algorithm>

#include "common.hpp"

namespace dgll {

struct GatherArgs {
    const void* cache;      // [n_cached, ldc] device rows (may be NULL when slot is NULL)
    const void* host;       // [n_nodes, ldh] rows: pinned host memory or a plain device matrix
    const int64_t* idx;     // [n] requested node ids
    const int64_t* slot;    // [n_nodes] cache slot of every node, -1 = not cached; NULL = everything from `host`
    void* out;              // [n, ldo]
    int64_t ldc, ldh, ldo, n;
    int row_bytes;          // bytes actually copied per row
    unsigned long long* miss_count;   // optional
    const int64_t* host_map; // optional [n_nodes]: row of `host` that holds node i (local -> full-graph id, storage.py:27)
};

template <int VEC>  // bytes per lane per step: 16, 4 or 2
__global__ __launch_bounds__(kBlock) void gather_rows_kernel(const GatherArgs a, int esz) {
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int steps = (a.row_bytes + VEC * kWave - 1) / (VEC * kWave);
    int64_t misses = 0;
    for (int64_t i = (int64_t)blockIdx.x * kWavesPerBlock + wave; i < a.n; i += (int64_t)gridDim.x * kWavesPerBlock) {
        const int64_t node = a.idx[i];
        const int64_t s = a.slot ? a.slot[node] : -1;
        const char* src = (s >= 0) ? static_cast<const char*>(a.cache) + s * a.ldc * esz
                                   : static_cast<const char*>(a.host) + (a.host_map ? a.host_map[node] : node) * a.ldh * esz;
        char* dst = static_cast<char*>(a.out) + i * a.ldo * esz;
        misses += (a.slot && s < 0) ? 1 : 0;
        for (int st = 0; st < steps; ++st) {
            const int off = (st * kWave + lane) * VEC;
            if (off < a.row_bytes) {
                if (VEC == 16) *reinterpret_cast<uint4*>(dst + off) = *reinterpret_cast<const uint4*>(src + off);
                else if (VEC == 4) *reinterpret_cast<uint32_t*>(dst + off) = *reinterpret_cast<const uint32_t*>(src + off);
                else *reinterpret_cast<uint16_t*>(dst + off) = *reinterpret_cast<const uint16_t*>(src + off);
            }
        }
    }
    if (a.miss_count && lane == 0 && misses) atomicAdd(a.miss_count, (unsigned long long)misses);
}

}  // namespace dgll

using namespace dgll;

static int gather_rows_impl(void* stream, const void* cache, int64_t ldc, const void* host, int64_t ldh,
                            const int64_t* idx, const int64_t* slot, void* out, int64_t ldo, int64_t n, int feat,
                            int dtype, unsigned long long* miss_count, const int64_t* host_map);

DGLL_API int dgll_hip_gather_rows(void* stream, const void* cache, int64_t ldc, const void* host, int64_t ldh,
                                  const int64_t* idx, const int64_t* slot, void* out, int64_t ldo, int64_t n, int feat,
                                  int dtype, unsigned long long* miss_count) {
    return gather_rows_impl(stream, cache, ldc, host, ldh, idx, slot, out, ldo, n, feat, dtype, miss_count, nullptr);
}

DGLL_API int dgll_hip_gather_rows_mapped(void* stream, const void* cache, int64_t ldc, const void* host, int64_t ldh,
                                         const int64_t* idx, const int64_t* slot, const int64_t* host_map, void* out,
                                         int64_t ldo, int64_t n, int feat, int dtype, unsigned long long* miss_count) {
    return gather_rows_impl(stream, cache, ldc, host, ldh, idx, slot, out, ldo, n, feat, dtype, miss_count, host_map);
}

static int gather_rows_impl(void* stream, const void* cache, int64_t ldc, const void* host, int64_t ldh,
                            const int64_t* idx, const int64_t* slot, void* out, int64_t ldo, int64_t n, int feat,
                            int dtype, unsigned long long* miss_count, const int64_t* host_map) {
    if (n <= 0 || feat <= 0) return DGLL_OK;
    DGLL_REQUIRE(host && idx && out, "NULL argument");
    DGLL_REQUIRE(!slot || cache, "a slot map needs a cache matrix");
    DGLL_REQUIRE(dtype == DGLL_F32 || dtype == DGLL_BF16, "dtype");
    const int esz = dtype == DGLL_BF16 ? 2 : 4;
    DGLL_REQUIRE(ldh >= feat && ldo >= feat && (!slot || ldc >= feat), "leading dimension smaller than feat");
    GatherArgs a{};
    a.cache = cache; a.host = host; a.idx = idx; a.slot = slot; a.out = out;
    a.ldc = ldc; a.ldh = ldh; a.ldo = ldo; a.n = n; a.row_bytes = feat * esz; a.miss_count = miss_count; a.host_map = host_map;
    auto ok16 = [&](const void* p, int64_t ld) { return !p || (aligned16(p) && (ld * esz) % 16 == 0); };
    const bool v16 = a.row_bytes % 16 == 0 && ok16(cache, ldc) && ok16(host, ldh) && ok16(out, ldo);
    const bool v4 = a.row_bytes % 4 == 0 && (ldh * esz) % 4 == 0 && (ldo * esz) % 4 == 0 && (!slot || (ldc * esz) % 4 == 0);
    const int64_t blocks = std::min<int64_t>((n + kWavesPerBlock - 1) / kWavesPerBlock, 256 * 16);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (v16) hipLaunchKernelGGL(gather_rows_kernel<16>, dim3((uint32_t)blocks), dim3(kBlock), 0, s, a, esz);
    else if (v4) hipLaunchKernelGGL(gather_rows_kernel<4>, dim3((uint32_t)blocks), dim3(kBlock), 0, s, a, esz);
    else hipLaunchKernelGGL(gather_rows_kernel<2>, dim3((uint32_t)blocks), dim3(kBlock), 0, s, a, esz);
    DGLL_HIP_TRY(hipGetLastError());
    return DGLL_OK;
}
