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
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <type_traits>

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
    // optional: rows of uncached nodes fetched into HBM beforehand (stage_rows_kernel below): stage_map[node] = (serial << 32) | staged row
    const unsigned long long* stage_map; const void* stage_rows; int64_t ld_stage; uint32_t stage_serial, stage_cap;
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
        const char* src;
        if (s >= 0) src = static_cast<const char*>(a.cache) + s * a.ldc * esz;
        else {
            const unsigned long long sv = a.stage_map ? a.stage_map[node] : 0ull;
            if ((uint32_t)(sv >> 32) == a.stage_serial && (uint32_t)sv < a.stage_cap)
                src = static_cast<const char*>(a.stage_rows) + (int64_t)(uint32_t)sv * a.ld_stage * esz;
            else src = static_cast<const char*>(a.host) + (a.host_map ? a.host_map[node] : node) * a.ldh * esz;
        }
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

// ---- aggregate straight out of the cache: out[i, :] = reduce_{k in [rowptr[i], rowptr[i+1])} row(idx[k]) ---------------------------------
// The outermost hop of a sampled mini-batch enters the model through its MEAN only (sageconv.py:33-36 on neighbor_node_features
// of the last hop): gathering its fan-out x batch rows into a matrix (fetch_data) only to reduce them in the next launch writes
// and re-reads the largest tensor of the batch.  Here the reduction reads the HBM cache / the pinned host rows directly -- one
// wavefront per destination row, neighbours NB at a time, fp32 accumulation -- and only the reduced [n_rows, F] rows are written.
struct AggregateArgs {
    const void* cache; const void* host;
    const int64_t* idx; const int64_t* slot; const int64_t* host_map; const int64_t* rowptr;
    void* out;
    int64_t ldc, ldh, ldo, n_rows;
    int row_bytes, mean;
    unsigned long long* miss_count;
    // optional: rows of uncached nodes fetched into HBM beforehand (stage_misses below): stage_map[node] = (serial << 32) | staged row
    const unsigned long long* stage_map; const void* stage_rows; int64_t ld_stage; uint32_t stage_serial, stage_cap;
};

template <typename T, int VEC> struct AggIO;
template <> struct AggIO<bf16_t, 16> {
    typedef uint4 raw_t;
    static __device__ __forceinline__ void add(const raw_t& r, float (&acc)[8]) {
        const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) { acc[2 * q] += __uint_as_float(w[q] << 16); acc[2 * q + 1] += __uint_as_float(w[q] & 0xffff0000u); }
    }
    static __device__ __forceinline__ raw_t pack(const float (&acc)[8], float sc) {
        return make_uint4(pack_bf16x2(acc[0] * sc, acc[1] * sc), pack_bf16x2(acc[2] * sc, acc[3] * sc),
                          pack_bf16x2(acc[4] * sc, acc[5] * sc), pack_bf16x2(acc[6] * sc, acc[7] * sc));
    }
};
template <> struct AggIO<bf16_t, 4> {
    typedef uint32_t raw_t;
    static __device__ __forceinline__ void add(const raw_t& r, float (&acc)[2]) {
        acc[0] += __uint_as_float(r << 16); acc[1] += __uint_as_float(r & 0xffff0000u);
    }
    static __device__ __forceinline__ raw_t pack(const float (&acc)[2], float sc) { return pack_bf16x2(acc[0] * sc, acc[1] * sc); }
};
template <> struct AggIO<float, 16> {
    typedef float4 raw_t;
    static __device__ __forceinline__ void add(const raw_t& r, float (&acc)[4]) { acc[0] += r.x; acc[1] += r.y; acc[2] += r.z; acc[3] += r.w; }
    static __device__ __forceinline__ raw_t pack(const float (&acc)[4], float sc) { return make_float4(acc[0] * sc, acc[1] * sc, acc[2] * sc, acc[3] * sc); }
};
template <> struct AggIO<float, 4> {
    typedef float raw_t;
    static __device__ __forceinline__ void add(const raw_t& r, float (&acc)[1]) { acc[0] += r; }
    static __device__ __forceinline__ raw_t pack(const float (&acc)[1], float sc) { return acc[0] * sc; }
};

template <typename T, int VEC>
__global__ __launch_bounds__(kBlock) void aggregate_rows_kernel(const AggregateArgs a) {
    typedef AggIO<T, VEC> IO;
    typedef typename IO::raw_t raw_t;
    constexpr int EL = VEC / (int)sizeof(T);
    constexpr int STEPS = VEC == 16 ? 2 : 5;       // bytes of a row handled per column block: STEPS * VEC * 64 (2048 / 1280)
    constexpr int NB = VEC == 16 ? 4 : 2;          // neighbour rows in flight
    constexpr int kBlockBytes = STEPS * VEC * kWave;
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int esz = (int)sizeof(T);
    unsigned long long misses = 0;
    for (int64_t i = (int64_t)blockIdx.x * kWavesPerBlock + wave; i < a.n_rows; i += (int64_t)gridDim.x * kWavesPerBlock) {
        const int64_t b = a.rowptr[i], e = a.rowptr[i + 1];
        const float scale = (a.mean && e > b) ? 1.0f / (float)(e - b) : 1.0f;
        char* dst = static_cast<char*>(a.out) + i * a.ldo * esz;
        for (int c0 = 0; c0 < a.row_bytes; c0 += kBlockBytes) {
            float acc[STEPS][EL];
#pragma unroll
            for (int st = 0; st < STEPS; ++st)
#pragma unroll
                for (int q = 0; q < EL; ++q) acc[st][q] = 0.0f;
            for (int64_t k = b; k < e; k += NB) {
                raw_t v[NB][STEPS];
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const int64_t kk = k + u < e ? k + u : e - 1;                    // the tail repeats the last neighbour (not added)
                    const int64_t node = a.idx[kk];
                    const int64_t s = a.slot ? a.slot[node] : -1;
                    const char* src;
                    if (s >= 0) src = static_cast<const char*>(a.cache) + s * a.ldc * esz;
                    else {                                                           // (wave-uniform: one node per wavefront and slot)
                        const unsigned long long sv = a.stage_map ? a.stage_map[node] : 0ull;
                        if ((uint32_t)(sv >> 32) == a.stage_serial && (uint32_t)sv < a.stage_cap)
                            src = static_cast<const char*>(a.stage_rows) + (int64_t)(uint32_t)sv * a.ld_stage * esz;
                        else src = static_cast<const char*>(a.host) + (a.host_map ? a.host_map[node] : node) * a.ldh * esz;
                    }
                    if (c0 == 0 && k + u < e && a.slot && s < 0) ++misses;
#pragma unroll
                    for (int st = 0; st < STEPS; ++st) {
                        const int off = c0 + (st * kWave + lane) * VEC;
                        if (off < a.row_bytes) v[u][st] = *reinterpret_cast<const raw_t*>(src + off);
                        else v[u][st] = raw_t{};
                    }
                }
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    if (k + u >= e) continue;                                        // wave-uniform
#pragma unroll
                    for (int st = 0; st < STEPS; ++st) IO::add(v[u][st], acc[st]);
                }
            }
#pragma unroll
            for (int st = 0; st < STEPS; ++st) {
                const int off = c0 + (st * kWave + lane) * VEC;
                if (off < a.row_bytes) *reinterpret_cast<raw_t*>(dst + off) = IO::pack(acc[st], scale);
            }
        }
    }
    if (a.miss_count && lane == 0 && misses) atomicAdd(a.miss_count, misses);
}

// ---- the misses of the outermost hop, fetched ahead of its reduction -------------------------------------------------------------------
// aggregate_rows_kernel on a partly cached store mixes two kinds of reads: 98.5 % of its rows come from HBM, 1.5 % over PCIe (Reddit
// shape, half of the nodes cached) -- and every wavefront that meets a miss sits on its CU for the link's latency.  The kernel then
// takes as long as the link needs for the misses (0.84 ms for 46 MB), with a chip-filling grid resident next to the training step all
// that time (the step's kernels ran 30 % slower beside it: tools/minibatch_timeline.py).  Split: (1) list_misses_kernel walks the hop's ids
// once and gives every DISTINCT uncached node a row of a staging buffer (a claim per node in stage_map: duplicates of a batch cross the
// link once); (2) stage_rows_kernel copies those rows from the pinned store into HBM with a SMALL grid -- one workgroup per CU already
// saturates the link (tools/probes/pcie_probe.py); 20 workgroups here, because what slows the training step beside a zero-copy kernel is the
// DEPTH of its read queue on the link, not the CUs it holds: at 32 workgroups and more (330 KB of reads in flight) the step's own kernels
// ran 1.4-1.6x longer, at 16-20 hardly (profiles/r06_minibatch_stage_sweep.log); (3) the reduction reads staged rows: HBM only.
// stage_map entries carry the batch's serial number in their upper half, so the map is never cleared.
constexpr unsigned long long kStageClaimed = 0xffffffffull, kStageOverflow = 0xfffffffeull;

constexpr int kListPerBlock = 2048;     // ids a workgroup of list_misses_kernel walks: its claims fit an LDS list, ONE counter update per workgroup

__global__ __launch_bounds__(kBlock) void list_misses_kernel(const int64_t* __restrict__ idx, int64_t n, const int64_t* __restrict__ slot,
                                                             unsigned long long* __restrict__ stage_map, uint32_t serial, uint32_t cap,
                                                             int64_t* __restrict__ list, unsigned int* __restrict__ count) {
    // (one atomicAdd per claim on the one counter took 0.2 ms for the 30 k claims of a Reddit batch: same-address atomics serialise)
    __shared__ int64_t claimed[kListPerBlock];
    __shared__ unsigned int n_claimed, base;
    if (threadIdx.x == 0) n_claimed = 0;
    __syncthreads();
    const unsigned long long tag = (unsigned long long)serial << 32;
    const int64_t lo = (int64_t)blockIdx.x * kListPerBlock, hi = lo + kListPerBlock < n ? lo + kListPerBlock : n;
    for (int64_t k = lo + threadIdx.x; k < hi; k += kBlock) {
        const int64_t node = idx[k];
        if (slot[node] >= 0) continue;
        const unsigned long long cur = __hip_atomic_load(stage_map + node, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t)(cur >> 32) == serial) continue;                              // already claimed for this batch
        if (atomicCAS(stage_map + node, cur, tag | kStageClaimed) != cur) continue;  // somebody else's claim came first
        claimed[atomicAdd(&n_claimed, 1u)] = node;
    }
    __syncthreads();
    if (threadIdx.x == 0) base = n_claimed ? atomicAdd(count, n_claimed) : 0u;
    __syncthreads();
    for (unsigned int j = threadIdx.x; j < n_claimed; j += kBlock) {
        const unsigned int row = base + j;
        const int64_t node = claimed[j];
        if (row < cap) list[row] = node;
        __hip_atomic_store(stage_map + node, tag | (row < cap ? (unsigned long long)row : kStageOverflow), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);                                // past the buffer: that node stays a zero-copy read
    }
}

template <int VEC>   // bytes per lane and load: 16 or 4
__global__ __launch_bounds__(kBlock) void stage_rows_kernel(const char* __restrict__ host, int64_t ldh_bytes, const int64_t* __restrict__ host_map,
                                                            const int64_t* __restrict__ list, const unsigned int* __restrict__ count,
                                                            uint32_t cap, char* __restrict__ stage, int64_t lds_bytes, int row_bytes) {
    typedef typename std::conditional<VEC == 16, uint4, uint32_t>::type raw_t;
    constexpr int STEPS = VEC == 16 ? 2 : 5, ROWS = 2;      // ROWS rows x STEPS loads of a wavefront in flight (2.5 KB at the Reddit width)
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t n = min(*count, cap);
    for (int64_t i = ((int64_t)blockIdx.x * kWavesPerBlock + wave) * ROWS; i < n; i += (int64_t)gridDim.x * kWavesPerBlock * ROWS) {
        const char* src[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int64_t node = list[i + r < n ? i + r : n - 1];
            src[r] = host + (host_map ? host_map[node] : node) * ldh_bytes;
        }
        for (int c0 = 0; c0 < row_bytes; c0 += STEPS * VEC * kWave) {
            raw_t v[ROWS][STEPS];
#pragma unroll
            for (int r = 0; r < ROWS; ++r)
#pragma unroll
                for (int st = 0; st < STEPS; ++st) {
                    const int off = c0 + (st * kWave + lane) * VEC;
                    if (off < row_bytes) v[r][st] = *reinterpret_cast<const raw_t*>(src[r] + off);
                }
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                if (i + r >= n) continue;
#pragma unroll
                for (int st = 0; st < STEPS; ++st) {
                    const int off = c0 + (st * kWave + lane) * VEC;
                    if (off < row_bytes) *reinterpret_cast<raw_t*>(stage + (i + r) * lds_bytes + off) = v[r][st];
                }
            }
        }
    }
}

}  // namespace dgll

using namespace dgll;

struct MissStage;
static int gather_rows_impl(void* stream, const void* cache, int64_t ldc, const void* host, int64_t ldh,
                            const int64_t* idx, const int64_t* slot, void* out, int64_t ldo, int64_t n, int feat,
                            int dtype, unsigned long long* miss_count, const int64_t* host_map, const MissStage* stage = nullptr);
struct MissStage {      // dgll_batch_load's stage_* members
    unsigned long long* map; void* rows; int64_t ld, cap; int64_t* list; unsigned int* count; uint32_t serial; int blocks;
};
static int aggregate_rows_impl(void* stream, const void* cache, int64_t ldc, const void* host, int64_t ldh, const int64_t* idx,
                               const int64_t* slot, const int64_t* host_map, const int64_t* rowptr, void* out, int64_t ldo, int64_t n_rows,
                               int feat, int dtype, int reduce, unsigned long long* miss_count, const MissStage* stage);
// list_misses_kernel over every id list, then stage_rows_kernel; *usable = false (nothing launched) when the rows are not 4-byte granular
static int stage_misses(hipStream_t s, const MissStage& stage, const void* host, int64_t ldh, const int64_t* host_map, const int64_t* slot,
                        int feat, int dtype, const int64_t* const* lists, const int64_t* counts, int n_lists, bool* usable);

// dgll_hip_debug_tune(12, n): cap the grids of the feature-loading kernels below at n workgroups per CU (0 = their defaults, 16 and
// 32).  They are grid-stride loops issued on a mini-batch pipeline's LOADING stream next to the training kernels: a smaller
// grid leaves wavefront slots of every CU to the compute stream instead of filling the chip.
int g_tune_loader_blocks_per_cu = 0;

namespace dgll {
// positions -> neighbour ids of one sampled hop: one wavefront per seed row, lanes over its kept neighbours
template <typename PT>
__global__ __launch_bounds__(kBlock) void translate_positions_kernel(const int64_t* __restrict__ indptr, const int64_t* __restrict__ indices,
                                                                     const int64_t* __restrict__ seeds, const int64_t* __restrict__ rowptr,
                                                                     int64_t n_rows, const PT* __restrict__ pos, int64_t* __restrict__ out) {
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    for (int64_t r = (int64_t)blockIdx.x * kWavesPerBlock + wave; r < n_rows; r += (int64_t)gridDim.x * kWavesPerBlock) {
        const int64_t b = rowptr[r], e = rowptr[r + 1];
        if (e <= b) continue;
        const int64_t* nb = indices + indptr[seeds[r]];
        for (int64_t k = b + lane; k < e; k += kWave) out[k] = nb[(int64_t)pos[k]];
    }
}
}  // namespace dgll

DGLL_API int dgll_hip_translate_positions(void* stream, const int64_t* indptr, const int64_t* indices, const int64_t* seeds,
                                          const int64_t* rowptr, int64_t n_rows, const void* positions, int pos_bytes, int64_t* out_ids) {
    DGLL_REQUIRE(n_rows >= 0, "negative row count");
    if (n_rows == 0) return DGLL_OK;
    DGLL_REQUIRE(indptr && indices && seeds && rowptr && positions && out_ids, "NULL argument");
    DGLL_REQUIRE(pos_bytes == 2 || pos_bytes == 4 || pos_bytes == 8, "positions are int16, int32 or int64");
    const dim3 grid((uint32_t)std::min<int64_t>((n_rows + kWavesPerBlock - 1) / kWavesPerBlock, 65536));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (pos_bytes == 2)
        hipLaunchKernelGGL(translate_positions_kernel<int16_t>, grid, dim3(kBlock), 0, s, indptr, indices, seeds, rowptr, n_rows,
                           static_cast<const int16_t*>(positions), out_ids);
    else if (pos_bytes == 4)
        hipLaunchKernelGGL(translate_positions_kernel<int32_t>, grid, dim3(kBlock), 0, s, indptr, indices, seeds, rowptr, n_rows,
                           static_cast<const int32_t*>(positions), out_ids);
    else
        hipLaunchKernelGGL(translate_positions_kernel<int64_t>, grid, dim3(kBlock), 0, s, indptr, indices, seeds, rowptr, n_rows,
                           static_cast<const int64_t*>(positions), out_ids);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "translate_positions_kernel launch");
    return DGLL_OK;
}

namespace dgll {
// out[k, :] = scale_r * g[r, :] for k in [rowptr[r], rowptr[r + 1]); the rows behind rowptr[n_rows] are zeroed.  One wavefront per
// destination row (then per tail row); VEC = 16: 16-byte lanes over rows on 16-byte pitches, else element by element.
template <typename T, int VEC>
__global__ __launch_bounds__(kBlock) void expand_rows_kernel(const int64_t* __restrict__ rowptr, int64_t n_rows, const T* __restrict__ g,
                                                             int64_t ldg, T* __restrict__ out, int64_t ldo, int64_t n_out_rows, int feat,
                                                             int mean, int accumulate) {
    constexpr int EL = VEC / (int)sizeof(T);
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t used = rowptr[n_rows];
    const int64_t items = n_rows + ((!accumulate && n_out_rows > used) ? n_out_rows - used : 0);
    const int vecs = (feat + EL - 1) / EL;
    for (int64_t it = (int64_t)blockIdx.x * kWavesPerBlock + wave; it < items; it += (int64_t)gridDim.x * kWavesPerBlock) {
        if (it >= n_rows) {                                   // a row of the unused tail: zeros
            T* dst = out + (used + (it - n_rows)) * ldo;
            for (int v = lane; v < vecs; v += kWave) {
                if (VEC == 16) *reinterpret_cast<uint4*>(dst + v * EL) = make_uint4(0, 0, 0, 0);
                else dst[v] = (T)0;
            }
            continue;
        }
        const int64_t b = rowptr[it], e = rowptr[it + 1];
        if (e <= b) continue;
        const float sc = mean ? 1.0f / (float)(e - b) : 1.0f;
        const T* src = g + it * ldg;
        for (int v = lane; v < vecs; v += kWave) {
            if constexpr (VEC == 16 && sizeof(T) == 2) {
                const uint4 r = *reinterpret_cast<const uint4*>(src + v * EL);
                const uint4 o = make_uint4(pack_bf16x2(bf16_lo(r.x) * sc, bf16_hi(r.x) * sc), pack_bf16x2(bf16_lo(r.y) * sc, bf16_hi(r.y) * sc),
                                           pack_bf16x2(bf16_lo(r.z) * sc, bf16_hi(r.z) * sc), pack_bf16x2(bf16_lo(r.w) * sc, bf16_hi(r.w) * sc));
                if (!accumulate) {
                    for (int64_t k = b; k < e; ++k) *reinterpret_cast<uint4*>(out + k * ldo + v * EL) = o;
                } else {
                    const float a8[8] = {bf16_lo(o.x), bf16_hi(o.x), bf16_lo(o.y), bf16_hi(o.y), bf16_lo(o.z), bf16_hi(o.z), bf16_lo(o.w), bf16_hi(o.w)};
                    for (int64_t k = b; k < e; ++k) {
                        uint4* d = reinterpret_cast<uint4*>(out + k * ldo + v * EL);
                        const uint4 p = *d;
                        *d = make_uint4(pack_bf16x2(bf16_lo(p.x) + a8[0], bf16_hi(p.x) + a8[1]), pack_bf16x2(bf16_lo(p.y) + a8[2], bf16_hi(p.y) + a8[3]),
                                        pack_bf16x2(bf16_lo(p.z) + a8[4], bf16_hi(p.z) + a8[5]), pack_bf16x2(bf16_lo(p.w) + a8[6], bf16_hi(p.w) + a8[7]));
                    }
                }
            } else if constexpr (VEC == 16) {
                float4 r = *reinterpret_cast<const float4*>(src + v * EL);
                r.x *= sc; r.y *= sc; r.z *= sc; r.w *= sc;
                for (int64_t k = b; k < e; ++k) {
                    float4* d = reinterpret_cast<float4*>(out + k * ldo + v * EL);
                    if (accumulate) { const float4 p = *d; *d = make_float4(p.x + r.x, p.y + r.y, p.z + r.z, p.w + r.w); }
                    else *d = r;
                }
            } else if constexpr (sizeof(T) == 2) {
                const bf16_t o = f32_to_bf16(bf16_to_f32(src[v]) * sc);
                for (int64_t k = b; k < e; ++k) out[k * ldo + v] = accumulate ? f32_to_bf16(bf16_to_f32(out[k * ldo + v]) + bf16_to_f32(o)) : o;
            } else {
                const float o = src[v] * sc;
                for (int64_t k = b; k < e; ++k) out[k * ldo + v] = accumulate ? out[k * ldo + v] + o : o;
            }
        }
    }
}
}  // namespace dgll

DGLL_API int dgll_hip_expand_rows(void* stream, const int64_t* rowptr, int64_t n_rows, const void* g, int64_t ldg, void* out, int64_t ldo,
                                  int64_t n_out_rows, int feat, int dtype, int mean, int accumulate) {
    DGLL_REQUIRE(n_rows >= 0 && n_out_rows >= 0 && feat >= 0, "negative size");
    if (n_out_rows == 0 || feat == 0) return DGLL_OK;
    DGLL_REQUIRE(rowptr && out && (g || n_rows == 0), "NULL argument");
    DGLL_REQUIRE(dtype == DGLL_F32 || dtype == DGLL_BF16, "dtype");
    DGLL_REQUIRE(ldg >= feat && ldo >= feat, "leading dimension smaller than feat");
    const int esz = dtype == DGLL_BF16 ? 2 : 4;
    const int el = 16 / esz;
    const int64_t padded = (int64_t)(feat + el - 1) / el * el;
    const bool vec = aligned16(g) && aligned16(out) && (ldg * esz) % 16 == 0 && (ldo * esz) % 16 == 0 && ldg >= padded && ldo >= padded;
    const int64_t items = n_rows + n_out_rows;          // an upper bound of the work items (the kernel reads the real count)
    const dim3 grid((uint32_t)std::min<int64_t>((items + kWavesPerBlock - 1) / kWavesPerBlock, 16384));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == DGLL_BF16) {
        if (vec) hipLaunchKernelGGL((expand_rows_kernel<bf16_t, 16>), grid, dim3(kBlock), 0, s, rowptr, n_rows, static_cast<const bf16_t*>(g), ldg,
                                    static_cast<bf16_t*>(out), ldo, n_out_rows, feat, mean, accumulate);
        else hipLaunchKernelGGL((expand_rows_kernel<bf16_t, 2>), grid, dim3(kBlock), 0, s, rowptr, n_rows, static_cast<const bf16_t*>(g), ldg,
                                static_cast<bf16_t*>(out), ldo, n_out_rows, feat, mean, accumulate);
    } else {
        if (vec) hipLaunchKernelGGL((expand_rows_kernel<float, 16>), grid, dim3(kBlock), 0, s, rowptr, n_rows, static_cast<const float*>(g), ldg,
                                    static_cast<float*>(out), ldo, n_out_rows, feat, mean, accumulate);
        else hipLaunchKernelGGL((expand_rows_kernel<float, 4>), grid, dim3(kBlock), 0, s, rowptr, n_rows, static_cast<const float*>(g), ldg,
                                static_cast<float*>(out), ldo, n_out_rows, feat, mean, accumulate);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "expand_rows_kernel launch");
    return DGLL_OK;
}

namespace dgll {
// dst[i] = src[i] for i <= n, src[n] (the edge count) for n < i <= cap: a batch's row pointers on a static block's shape
__global__ __launch_bounds__(kBlock) void pad_rowptr_kernel(const int64_t* __restrict__ src, int64_t n, int64_t* __restrict__ dst, int64_t cap) {
    const int64_t last = src[n];
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i <= cap; i += (int64_t)gridDim.x * kBlock) dst[i] = i <= n ? src[i] : last;
}
// dst[i] = labels[ids[i]] for i < n, fill for n <= i < cap
__global__ __launch_bounds__(kBlock) void gather_labels_kernel(const int64_t* __restrict__ labels, const int64_t* __restrict__ ids, int64_t n,
                                                               int64_t* __restrict__ dst, int64_t cap, int64_t fill) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < cap; i += (int64_t)gridDim.x * kBlock) dst[i] = i < n ? labels[ids[i]] : fill;
}
}  // namespace dgll

// Upload of a pinned (device-visible) host buffer by a KERNEL on the loading stream: 16-byte loads over PCIe, stores to HBM.
// hipMemcpyAsync does the same through the copy engines -- but when the stream's last command is a not-yet-resolved cross-stream
// wait (the loading stage waits for the replay that read the input set it is about to overwrite), the call BLOCKED ON THE HOST until
// that dependency had resolved: 7-19 ms, about once in thirty batches (DGLL_LOADER_STAMPS=1), each time draining the loaded-batch queue
// behind it.  A kernel is simply ordered behind the wait on the device.  Few workgroups (16; DGLL_LOADER_UPLOAD_BLOCKS): the link, not
// the CUs, bounds it, and a deep queue of reads on the link slows the kernels of the training step beside it (see stage_rows_kernel).
__global__ __launch_bounds__(kBlock) void upload_kernel(const char* __restrict__ src, char* __restrict__ dst, size_t bytes) {
    const size_t vecs = bytes / 16;
    const uint4* __restrict__ s4 = reinterpret_cast<const uint4*>(src);
    uint4* __restrict__ d4 = reinterpret_cast<uint4*>(dst);
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < vecs; i += (size_t)gridDim.x * kBlock) d4[i] = s4[i];
    const size_t tail = vecs * 16;
    for (size_t i = tail + (size_t)blockIdx.x * kBlock + threadIdx.x; i < bytes; i += (size_t)gridDim.x * kBlock) dst[i] = src[i];
}

static int upload_async(hipStream_t s, void* dst, const void* src, size_t bytes, int blocks_wanted = 0) {
    static const bool by_copy_engine = []() { const char* v = std::getenv("DGLL_LOADER_UPLOAD"); return v && std::string(v) == "memcpy"; }();
    if (bytes == 0) return DGLL_OK;
    const bool aligned = aligned16(dst) && aligned16(src);
    bool visible = false;                       // only page-locked, device-mapped host memory may be read by a kernel
    if (!by_copy_engine && aligned) {
        hipPointerAttribute_t attr{};
        if (hipPointerGetAttributes(&attr, src) == hipSuccess) visible = attr.type == hipMemoryTypeHost && attr.devicePointer != nullptr;
        else (void)hipGetLastError();          // an unregistered (pageable) pointer: clear the error, take the copy engine
    }
    if (by_copy_engine || !aligned || !visible) {
        DGLL_HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s));
        return DGLL_OK;
    }
    static const int max_blocks = []() { const char* v = std::getenv("DGLL_LOADER_UPLOAD_BLOCKS"); const int n = v ? std::atoi(v) : 0; return n > 0 ? n : 16; }();
    const int blocks = (int)std::min<size_t>((bytes / 16 + kBlock - 1) / kBlock + 1, (size_t)(blocks_wanted > 0 ? blocks_wanted : max_blocks));
    hipPointerAttribute_t attr{};
    (void)hipPointerGetAttributes(&attr, src);
    const char* dsrc = static_cast<const char*>(attr.devicePointer ? attr.devicePointer : src);
    hipLaunchKernelGGL(upload_kernel, dim3(blocks), dim3(kBlock), 0, s, dsrc, static_cast<char*>(dst), bytes);
    DGLL_HIP_TRY(hipGetLastError());
    return DGLL_OK;
}

DGLL_API int dgll_hip_load_sampled_batch(void* stream, const dgll_batch_load* b) {
    DGLL_REQUIRE(b != nullptr, "NULL batch");
    const int L = b->n_hops;
    DGLL_REQUIRE(L >= 1 && L <= 8, "1 .. 8 hops");
    DGLL_REQUIRE(b->staged_host && b->staged_dev && b->staged_entries > 0, "staging buffer");
    DGLL_REQUIRE(b->n_outer == 0 || (b->pos_host && b->pos_dev && b->ids_out && b->indptr && b->indices), "outermost hop arguments");
    hipStream_t s = static_cast<hipStream_t>(stream);
    // DGLL_LOADER_STAMPS=1 (diagnostics): host time of every enqueue of this call; any that takes more than a millisecond is reported
    static const bool stamps = std::getenv("DGLL_LOADER_STAMPS") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto stamp = [&](const char* what) {
        if (!stamps) return;
        const auto now = std::chrono::steady_clock::now();
        const double ms = std::chrono::duration<double, std::milli>(now - t_prev).count();
        if (ms > 1.0) std::fprintf(stderr, "[dgll loader] %s took %.2f ms on the host\n", what, ms);
        t_prev = now;
    };
    // staged_host / pos_host must be pinned, device-visible memory (hipHostMalloc / torch pin_memory): they are read by a kernel
    int up = upload_async(s, b->staged_dev, b->staged_host, (size_t)b->staged_entries * 8, b->upload_blocks);
    if (up != DGLL_OK) return up;
    stamp("upload of the staged arrays");
    if (b->n_outer > 0) {
        up = upload_async(s, b->pos_dev, b->pos_host, (size_t)b->n_outer * (size_t)b->pos_bytes, b->upload_blocks);
        if (up != DGLL_OK) return up;
    }
    stamp("upload of the positions");
    const int64_t* st = b->staged_dev;
    const int64_t* ids_of[8];
    ids_of[0] = st + b->seeds_off;
    for (int h = 1; h < L; ++h) ids_of[h] = st + b->src_off[h - 1];           // rows of hop h = the sources around hop h - 1
    int code = DGLL_OK;
    if (b->n_outer > 0) {
        code = dgll_hip_translate_positions(stream, b->indptr, b->indices, ids_of[L - 1], st + b->ptr_off[L - 1], b->rows[L - 1], b->pos_dev,
                                            b->pos_bytes, b->ids_out);
        if (code != DGLL_OK) return code;
        stamp("translate launch");
    }
    // the uncached rows of every hop of the batch, each distinct node once, into HBM ahead of the gathers and the reduction
    MissStage stage{reinterpret_cast<unsigned long long*>(b->stage_map), b->stage_rows, b->ld_stage, b->stage_cap, b->stage_list,
                    b->stage_count, b->stage_serial, b->stage_blocks};
    bool staged = b->stage_map && b->slot;
    if (staged) {
        DGLL_REQUIRE(b->stage_rows && b->stage_list && b->stage_count && b->stage_cap > 0 && b->stage_cap < 0xfffffffell &&
                     b->stage_serial != 0 && b->ld_stage >= b->feat, "staging of the uncached rows: buffers, capacity, serial");
        const int64_t* lists[9]; int64_t counts[9]; int n_lists = 0;
        for (int h = 0; h < L; ++h)
            if (b->rows[h] > 0 && b->feat_out[h]) { lists[n_lists] = ids_of[h]; counts[n_lists++] = b->rows[h]; }
        if (b->reduced_out && b->rows[L - 1] > 0 && b->n_outer > 0) { lists[n_lists] = b->ids_out; counts[n_lists++] = b->n_outer; }
        code = stage_misses(s, stage, b->host, b->ldh, b->host_map, b->slot, b->feat, b->dtype, lists, counts, n_lists, &staged);
        if (code != DGLL_OK) return code;
        stamp("staging launches");
    }
    for (int h = 0; h < L; ++h) {
        if (b->rows[h] <= 0 || !b->feat_out[h]) continue;
        code = gather_rows_impl(stream, b->cache, b->ldc, b->host, b->ldh, ids_of[h], b->slot, b->feat_out[h], b->ld_feat, b->rows[h], b->feat,
                                b->dtype, b->miss_count, b->host_map, staged ? &stage : nullptr);
        if (code != DGLL_OK) return code;
        stamp("gather launch");
    }
    if (b->reduced_out && b->rows[L - 1] > 0) {
        code = aggregate_rows_impl(stream, b->cache, b->ldc, b->host, b->ldh, b->ids_out, b->slot, b->host_map, st + b->ptr_off[L - 1],
                                   b->reduced_out, b->ld_reduced, b->rows[L - 1], b->feat, b->dtype, b->reduce, b->miss_count,
                                   staged ? &stage : nullptr);
        if (code != DGLL_OK) return code;
        stamp("aggregate launch");
    }
    for (int h = 0; h + 1 < L; ++h) {
        if (!b->rowptr_out[h]) continue;
        DGLL_REQUIRE(b->rows[h] <= b->rowptr_cap[h], "a hop with more rows than the static block holds");
        const int blocks = (int)std::min<int64_t>((b->rowptr_cap[h] + kBlock) / kBlock, 1024);
        hipLaunchKernelGGL(pad_rowptr_kernel, dim3(blocks), dim3(kBlock), 0, s, st + b->ptr_off[h], b->rows[h], b->rowptr_out[h], b->rowptr_cap[h]);
    }
    if (b->labels && b->labels_out && b->labels_cap > 0) {
        DGLL_REQUIRE(b->rows[0] <= b->labels_cap, "more seeds than the label buffer holds");
        const int blocks = (int)std::min<int64_t>((b->labels_cap + kBlock - 1) / kBlock, 1024);
        hipLaunchKernelGGL(gather_labels_kernel, dim3(blocks), dim3(kBlock), 0, s, b->labels, ids_of[0], b->rows[0], b->labels_out, b->labels_cap,
                           b->label_fill);
    }
    stamp("row-pointer / label launches");
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "dgll_hip_load_sampled_batch launches");
    return DGLL_OK;
}

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

static int stage_misses(hipStream_t s, const MissStage& stage, const void* host, int64_t ldh, const int64_t* host_map, const int64_t* slot,
                        int feat, int dtype, const int64_t* const* lists, const int64_t* counts, int n_lists, bool* usable) {
    const int esz = dtype == DGLL_BF16 ? 2 : 4;
    const int64_t row_bytes = (int64_t)feat * esz, whole = ((row_bytes + 15) / 16) * 16;
    auto ok = [&](const void* p, int64_t ld, int vec) { return (reinterpret_cast<uintptr_t>(p) % vec) == 0 && (ld * esz) % vec == 0; };
    const bool v16 = ok(host, ldh, 16) && ok(stage.rows, stage.ld, 16) && ldh * esz >= whole && stage.ld * esz >= whole;
    const bool v4 = row_bytes % 4 == 0 && ok(host, ldh, 4) && ok(stage.rows, stage.ld, 4);
    *usable = v16 || v4;
    if (!*usable || n_lists == 0) { *usable = *usable && n_lists > 0; return DGLL_OK; }
    DGLL_HIP_TRY(hipMemsetAsync(stage.count, 0, sizeof(unsigned int), s));
    for (int i = 0; i < n_lists; ++i) {
        const int64_t lb = (counts[i] + kListPerBlock - 1) / kListPerBlock;
        hipLaunchKernelGGL(list_misses_kernel, dim3((uint32_t)lb), dim3(kBlock), 0, s, lists[i], counts[i], slot, stage.map, stage.serial,
                           (uint32_t)stage.cap, stage.list, stage.count);
    }
    // default grid: ~192 KB of reads in flight on the link (two rows, or two 2048 / 1280-byte pieces of them, per wavefront): 20 workgroups
    // at the Reddit row of 1204 bytes -- the measured optimum (16: 744, 20: 783-795, 24: 725, 32: 630 batches/s)
    const int64_t piece = std::min<int64_t>(v16 ? whole : row_bytes, v16 ? 2048 : 1280);
    const int sb = stage.blocks > 0 ? stage.blocks
                                    : (int)std::min<int64_t>(std::max<int64_t>(196608 / (2 * piece) / kWavesPerBlock, 4), 256);
    if (v16) hipLaunchKernelGGL(stage_rows_kernel<16>, dim3(sb), dim3(kBlock), 0, s, static_cast<const char*>(host), ldh * esz, host_map, stage.list,
                                stage.count, (uint32_t)stage.cap, static_cast<char*>(stage.rows), stage.ld * esz, (int)whole);
    else hipLaunchKernelGGL(stage_rows_kernel<4>, dim3(sb), dim3(kBlock), 0, s, static_cast<const char*>(host), ldh * esz, host_map, stage.list,
                            stage.count, (uint32_t)stage.cap, static_cast<char*>(stage.rows), stage.ld * esz, (int)row_bytes);
    DGLL_HIP_TRY(hipGetLastError());
    return DGLL_OK;
}

static int gather_rows_impl(void* stream, const void* cache, int64_t ldc, const void* host, int64_t ldh,
                            const int64_t* idx, const int64_t* slot, void* out, int64_t ldo, int64_t n, int feat,
                            int dtype, unsigned long long* miss_count, const int64_t* host_map, const MissStage* stage) {
    if (n <= 0 || feat <= 0) return DGLL_OK;
    DGLL_REQUIRE(host && idx && out, "NULL argument");
    DGLL_REQUIRE(!slot || cache, "a slot map needs a cache matrix");
    DGLL_REQUIRE(dtype == DGLL_F32 || dtype == DGLL_BF16, "dtype");
    const int esz = dtype == DGLL_BF16 ? 2 : 4;
    DGLL_REQUIRE(ldh >= feat && ldo >= feat && (!slot || ldc >= feat), "leading dimension smaller than feat");
    GatherArgs a{};
    a.cache = cache; a.host = host; a.idx = idx; a.slot = slot; a.out = out;
    a.ldc = ldc; a.ldh = ldh; a.ldo = ldo; a.n = n; a.row_bytes = feat * esz; a.miss_count = miss_count; a.host_map = host_map;
    if (stage) {
        a.stage_map = stage->map; a.stage_rows = stage->rows; a.ld_stage = stage->ld; a.stage_serial = stage->serial;
        a.stage_cap = (uint32_t)stage->cap;
    }
    auto ok16 = [&](const void* p, int64_t ld) { return !p || (aligned16(p) && (ld * esz) % 16 == 0); };
    const bool v16 = a.row_bytes % 16 == 0 && ok16(cache, ldc) && ok16(host, ldh) && ok16(out, ldo) && (!stage || ok16(stage->rows, stage->ld));
    const bool v4 = a.row_bytes % 4 == 0 && (ldh * esz) % 4 == 0 && (ldo * esz) % 4 == 0 && (!slot || (ldc * esz) % 4 == 0) &&
                    (!stage || (stage->ld * esz) % 4 == 0);
    const int64_t blocks = std::min<int64_t>((n + kWavesPerBlock - 1) / kWavesPerBlock, 256 * (g_tune_loader_blocks_per_cu > 0 ? g_tune_loader_blocks_per_cu : 16));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (v16) hipLaunchKernelGGL(gather_rows_kernel<16>, dim3((uint32_t)blocks), dim3(kBlock), 0, s, a, esz);
    else if (v4) hipLaunchKernelGGL(gather_rows_kernel<4>, dim3((uint32_t)blocks), dim3(kBlock), 0, s, a, esz);
    else hipLaunchKernelGGL(gather_rows_kernel<2>, dim3((uint32_t)blocks), dim3(kBlock), 0, s, a, esz);
    DGLL_HIP_TRY(hipGetLastError());
    return DGLL_OK;
}


DGLL_API int dgll_hip_aggregate_rows_mapped(void* stream, const void* cache, int64_t ldc, const void* host, int64_t ldh,
                                            const int64_t* idx, const int64_t* slot, const int64_t* host_map, const int64_t* rowptr,
                                            void* out, int64_t ldo, int64_t n_rows, int feat, int dtype, int reduce,
                                            unsigned long long* miss_count) {
    return aggregate_rows_impl(stream, cache, ldc, host, ldh, idx, slot, host_map, rowptr, out, ldo, n_rows, feat, dtype, reduce, miss_count,
                               nullptr);
}

// stage: the uncached rows were fetched by stage_misses on this stream
static int aggregate_rows_impl(void* stream, const void* cache, int64_t ldc, const void* host, int64_t ldh, const int64_t* idx,
                               const int64_t* slot, const int64_t* host_map, const int64_t* rowptr, void* out, int64_t ldo, int64_t n_rows,
                               int feat, int dtype, int reduce, unsigned long long* miss_count, const MissStage* stage) {
    if (n_rows <= 0 || feat <= 0) return DGLL_OK;
    DGLL_REQUIRE(host && idx && rowptr && out, "NULL argument");
    DGLL_REQUIRE(!slot || cache, "a slot map needs a cache matrix");
    DGLL_REQUIRE(dtype == DGLL_F32 || dtype == DGLL_BF16, "dtype");
    DGLL_REQUIRE(reduce == DGLL_REDUCE_SUM || reduce == DGLL_REDUCE_MEAN, "reduce");
    const int esz = dtype == DGLL_BF16 ? 2 : 4;
    DGLL_REQUIRE(ldh >= feat && ldo >= feat && (!slot || ldc >= feat), "leading dimension smaller than feat");
    AggregateArgs a{};
    a.cache = cache; a.host = host; a.idx = idx; a.slot = slot; a.host_map = host_map; a.rowptr = rowptr; a.out = out;
    a.ldc = ldc; a.ldh = ldh; a.ldo = ldo; a.n_rows = n_rows; a.row_bytes = feat * esz; a.mean = reduce == DGLL_REDUCE_MEAN;
    a.miss_count = miss_count;
    auto ok = [&](const void* p, int64_t ld, int vec) {
        return !p || ((reinterpret_cast<uintptr_t>(p) % vec) == 0 && (ld * esz) % vec == 0);
    };
    const void* srows = stage ? stage->rows : nullptr;
    const int64_t lds = stage ? stage->ld : 0;
    // 16-byte lanes when every row of every source starts on a 16-byte boundary (the row's last vector may reach into its padding:
    // pitches are whole vectors), else 4-byte lanes; rows that are not even 4-byte granular are the caller's to fetch and reduce
    const int64_t whole = ((a.row_bytes + 15) / 16) * 16;
    const bool v16 = ok(cache, ldc, 16) && ok(host, ldh, 16) && ok(out, ldo, 16) && ok(srows, lds, 16) && (!slot || ldc * esz >= whole) &&
                     ldh * esz >= whole && ldo * esz >= whole && (!stage || lds * esz >= whole);
    const bool v4 = a.row_bytes % 4 == 0 && ok(cache, ldc, 4) && ok(host, ldh, 4) && ok(out, ldo, 4) && ok(srows, lds, 4);
    DGLL_REQUIRE(v16 || v4, "rows must be at least 4-byte granular (even bf16 width, 4-byte aligned pitches)");
    if (v16) a.row_bytes = (int)whole;       // whole vectors: the padding columns are summed and written too
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (stage) {
        a.stage_map = stage->map; a.stage_rows = stage->rows; a.ld_stage = stage->ld; a.stage_serial = stage->serial;
        a.stage_cap = (uint32_t)stage->cap;
    }
    const int64_t blocks = std::min<int64_t>((n_rows + kWavesPerBlock - 1) / kWavesPerBlock, 256 * (g_tune_loader_blocks_per_cu > 0 ? g_tune_loader_blocks_per_cu : 32));
    if (dtype == DGLL_BF16) {
        if (v16) hipLaunchKernelGGL((aggregate_rows_kernel<bf16_t, 16>), dim3((uint32_t)blocks), dim3(kBlock), 0, s, a);
        else hipLaunchKernelGGL((aggregate_rows_kernel<bf16_t, 4>), dim3((uint32_t)blocks), dim3(kBlock), 0, s, a);
    } else {
        if (v16) hipLaunchKernelGGL((aggregate_rows_kernel<float, 16>), dim3((uint32_t)blocks), dim3(kBlock), 0, s, a);
        else hipLaunchKernelGGL((aggregate_rows_kernel<float, 4>), dim3((uint32_t)blocks), dim3(kBlock), 0, s, a);
    }
    DGLL_HIP_TRY(hipGetLastError());
    return DGLL_OK;
}
