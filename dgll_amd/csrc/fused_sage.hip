// fused_sage.hip -- aggregate -> transform in ONE kernel: the MI355X redesign of the reference's only native kernel.
//
//   out[i, :] = act( A1[i, :] . W1  +  reduce_{j in row i} X[j, :] . W2  + bias )          (bf16 storage, fp32 accumulation)
//
// Reference: dgll/FusedKernel/gcn_fused_kernel.cu:5-74 computes relu(A . X . W) in one launch (one thread per output element,
// recomputing X.W per neighbour); sageConv (sageconv.py:33-41,70-83) is the same shape with a second, un-aggregated operand.
// Here a workgroup (4 wavefronts) owns a tile of 64 destination rows:
//   1. gather: the wavefronts take the tile's rows one at a time (LDS ticket) and aggregate each exactly as spmm.hip does
//      (gather.hpp: 16-byte lanes, 8 loads in flight, fp32 accumulation, no atomics), scale by 1/deg for the mean and park the
//      bf16 row in the LDS tile -- and, when the caller trains, also write it to `agg_out` (the weight gradient agg^T . g
//      needs it);
//   2. transform: v_mfma_f32_32x32x16_bf16 with the WEIGHTS as the MFMA "A" operand (16-byte loads from L2: the weight
//      matrices are a few hundred KiB and shared by every tile; each fragment feeds the tile's two 32-row groups) and the
//      ACTIVATIONS as "B" -- the aggregate straight from the LDS tile, the self operand straight from global memory.
//      Wavefront w produces output columns [64 w, 64 w + 64);
//   3. epilogue: bias / ReLU, the tile is reused to transpose the accumulators, whole 16-byte row segments are stored.
// The aggregated matrix is never READ back from HBM and the separate transform launch disappears.
//
// MEASURED (MI355X, products-sized graph, F = K = N = 256, tools/fused_probe.py): 6.70 ms, against 4.35 ms (SpMM) + 0.88 ms
// (MFMA transform) for the two launches it replaces; 5.70 ms without the self operand (the reference kernel's relu(A.X.W)).
// Every variant kept the same verdict (32-row tiles / 6 wavefronts per SIMD: 7.5 ms; fixed rows per wavefront: the same):
// the gather needs all the CU's wavefront slots and registers to keep enough loads in flight, and the transform phase's
// sixteen dependent L2 round trips for weight fragments -- each queued behind the gathers that saturate the memory system --
// come straight out of that.  The layers therefore run SpMM + MFMA transform (fused_layers.FUSE_AGGREGATE_TRANSFORM = False);
// this kernel stays as the single-launch form of the reference's fused op, correct and tested, for callers that want it.
// Rows longer than the plan's threshold are aggregated beforehand by spmm.hip's chunk path (dgll_spmm_csr_impl, only_long)
// into `agg_out`; the tile then just loads those rows.
// W2 == NULL: the aggregate is ADDED to the output instead of transformed (feat == N): the narrowing "transform first" layer.
#include <algorithm>

#include "common.hpp"
#include "gather.hpp"

namespace dgll {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

constexpr int kTileRows = 64;          // two 32-row MFMA row groups per tile: every weight fragment feeds two MFMAs
constexpr int kRowsPerWave = kTileRows / kWavesPerBlock;

struct FusedSageArgs {
    const int64_t* rowptr;
    const int32_t* col;
    const float* val;
    const bf16_t* X;          // gathered matrix [n_cols, ldx]
    int64_t ldx;
    int feat, kagg;           // aggregate width, and rounded up to 64 (what the MFMA reads; the tile's tail columns are zero)
    int reduce, threshold;
    bf16_t* agg_out;          // optional [n_rows, ldagg]
    int64_t ldagg;
    const bf16_t* A1;         // optional self operand [n_rows, K1]
    int64_t lda1;
    int K1;
    const bf16_t* Wt1;        // [N padded to 64 | 128 | 256 rows, K1 padded to 64] (W1 transposed)
    int64_t ldw1;
    const bf16_t* Wt2;        // [same rows, kagg] or NULL (aggregate added, not transformed)
    int64_t ldw2;
    const float* bias;
    bf16_t* out;
    int64_t ldo;
    int N, relu;
    int64_t n_rows;
    int pitch;                // bytes per tile row in LDS
};

__device__ __forceinline__ uint4 keep_first(uint4 v, int valid) {  // zero all but the first `valid` (0..8) bf16 of a vector
    uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int d = 0; d < 4; ++d) w[d] &= (2 * d < valid ? 0x0000ffffu : 0u) | (2 * d + 1 < valid ? 0xffff0000u : 0u);
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// (the occupancy bound makes hipcc share registers between the two phases: without it the accumulators go to AGPRs ON TOP of
// the gather phase's registers although the phases never overlap)
template <int LPR, bool HAS_VAL, int NTW, int U>
__global__ __launch_bounds__(kBlock, 4) void fused_sage_kernel(const FusedSageArgs a) {
    extern __shared__ __attribute__((aligned(16))) char tile[];
    constexpr int EPV = 8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t row0 = (int64_t)blockIdx.x * kTileRows;
    __shared__ int next_row;
    if (tid == 0) next_row = 0;
    __syncthreads();

    // ---------------------------------------------------------------- 1. gather: 8 rows per wavefront into the LDS tile
    {
        const int sub = lane % LPR;
        const int c0 = sub * EPV;
        const bool col_ok = c0 < a.feat;
        const bf16_t* xcol = a.X + (col_ok ? c0 : 0);
        const bool full = c0 + EPV <= a.feat;
        // the tile's rows are handed out one at a time from an LDS ticket: the workgroup meets at a barrier after the gather, and
        // with a fixed 16 rows per wavefront the power-law row lengths leave three wavefronts waiting for the fourth
        for (;;) {
            int tr = 0;
            if (lane == 0) tr = atomicAdd(&next_row, 1);
            tr = __builtin_amdgcn_readfirstlane(tr);
            if (tr >= kTileRows) break;
            const int64_t row = row0 + tr;
            uint4 packed = make_uint4(0, 0, 0, 0);
            if (row < a.n_rows) {
                const int64_t b = uniform64(a.rowptr[row]), e = uniform64(a.rowptr[row + 1]);
                if (a.threshold > 0 && e - b > a.threshold) {            // aggregated beforehand (chunk path): already scaled
                    if (lane < LPR && col_ok) {
                        const bf16_t* p = a.agg_out + row * a.ldagg + c0;
                        if (full) packed = *reinterpret_cast<const uint4*>(p);
                        else {
                            uint32_t w[4] = {0u, 0u, 0u, 0u};
                            for (int i = 0; i < EPV && c0 + i < a.feat; ++i) w[i >> 1] |= (uint32_t)p[i] << ((i & 1) * 16);
                            packed = make_uint4(w[0], w[1], w[2], w[3]);
                        }
                    }
                } else {
                    float acc[EPV];
#pragma unroll
                    for (int i = 0; i < EPV; ++i) acc[i] = 0.0f;
                    gather_edges<bf16_t, EPV, LPR, HAS_VAL, U>(a.col, a.val, xcol, a.ldx, b, e, lane, acc);
                    const float scale = (a.reduce == DGLL_REDUCE_MEAN && e > b) ? 1.0f / (float)(e - b) : 1.0f;
                    packed = make_uint4(pack_bf16x2(acc[0] * scale, acc[1] * scale), pack_bf16x2(acc[2] * scale, acc[3] * scale),
                                        pack_bf16x2(acc[4] * scale, acc[5] * scale), pack_bf16x2(acc[6] * scale, acc[7] * scale));
                    if (!full) packed = keep_first(packed, col_ok ? a.feat - c0 : 0);   // the row's padding never reaches the MFMA
                    if (a.agg_out && lane < LPR && col_ok) {
                        bf16_t* p = a.agg_out + row * a.ldagg + c0;
                        if (full) *reinterpret_cast<uint4*>(p) = packed;
                        else {
                            const uint32_t w[4] = {packed.x, packed.y, packed.z, packed.w};
                            for (int i = 0; i < EPV && c0 + i < a.feat; ++i) p[i] = (bf16_t)((w[i >> 1] >> ((i & 1) * 16)) & 0xffffu);
                        }
                    }
                }
            }
            if (lane < LPR && c0 < a.kagg) *reinterpret_cast<uint4*>(tile + tr * a.pitch + c0 * 2) = packed;
        }
    }
    __syncthreads();

    // ---------------------------------------------------------------- 2. transform: 64 rows x NTW * 32 columns per wavefront
    const int half = lane >> 5, l32 = lane & 31;
    const int n0 = wave * NTW * 32;
    const int n_pad = (a.N + 31) & ~31;
    const bool active = n0 < n_pad;
    constexpr int RG = kTileRows / 32;                    // row groups
    f32x16_t acc[RG][NTW];
#pragma unroll
    for (int g = 0; g < RG; ++g)
#pragma unroll
        for (int t = 0; t < NTW; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[g][t][r] = 0.0f;
    if (active) {
        if (a.A1) {        // self operand: activations from global memory (rows clamped, the K tail masked)
            const bf16_t* ap[RG];
#pragma unroll
            for (int g = 0; g < RG; ++g) {
                const int64_t arow = row0 + g * 32 + l32 < a.n_rows ? row0 + g * 32 + l32 : a.n_rows - 1;
                ap[g] = a.A1 + arow * a.lda1;
            }
            const bf16_t* wp = a.Wt1 + (int64_t)(n0 + l32) * a.ldw1;
            for (int k0 = 0; k0 < a.K1; k0 += 32) {
                uint4 bv[RG][2], wv[NTW][2];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int k = k0 + ks * 16 + half * 8;
#pragma unroll
                    for (int g = 0; g < RG; ++g) {
                        bv[g][ks] = make_uint4(0, 0, 0, 0);
                        if (k < a.K1) {
                            bv[g][ks] = *reinterpret_cast<const uint4*>(ap[g] + k);
                            if (k + 8 > a.K1) bv[g][ks] = keep_first(bv[g][ks], a.K1 - k);
                        }
                    }
#pragma unroll
                    for (int t = 0; t < NTW; ++t)
                        wv[t][ks] = n0 + t * 32 < n_pad ? *reinterpret_cast<const uint4*>(wp + (int64_t)t * 32 * a.ldw1 + k) : make_uint4(0, 0, 0, 0);
                }
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int t = 0; t < NTW; ++t)
#pragma unroll
                        for (int g = 0; g < RG; ++g)
                            acc[g][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wv[t][ks]),
                                                                                __builtin_bit_cast(bf16x8_t, bv[g][ks]), acc[g][t], 0, 0, 0);
            }
        }
        if (a.Wt2) {       // aggregate: activations from the LDS tile
            const char* tp = tile + l32 * a.pitch + half * 16;
            const bf16_t* wp = a.Wt2 + (int64_t)(n0 + l32) * a.ldw2 + half * 8;
            for (int k0 = 0; k0 < a.kagg; k0 += 32) {
                uint4 bv[RG][2], wv[NTW][2];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                    for (int g = 0; g < RG; ++g) bv[g][ks] = *reinterpret_cast<const uint4*>(tp + g * 32 * a.pitch + (k0 + ks * 16) * 2);
#pragma unroll
                    for (int t = 0; t < NTW; ++t)
                        wv[t][ks] = n0 + t * 32 < n_pad ? *reinterpret_cast<const uint4*>(wp + (int64_t)t * 32 * a.ldw2 + k0 + ks * 16) : make_uint4(0, 0, 0, 0);
                }
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int t = 0; t < NTW; ++t)
#pragma unroll
                        for (int g = 0; g < RG; ++g)
                            acc[g][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wv[t][ks]),
                                                                                __builtin_bit_cast(bf16x8_t, bv[g][ks]), acc[g][t], 0, 0, 0);
            }
        }
    }
    if (a.Wt2) __syncthreads();          // every wavefront is done reading the aggregate: the tile becomes the output stage

    // ---------------------------------------------------------------- 3. epilogue
    // D[i][j]: j = lane % 32 = the row inside its group, i = (r & 3) + 8 (r >> 2) + 4 half = the output column inside a 32-column tile
    if (active) {
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) {
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
                if (n0 + t * 32 >= n_pad) continue;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = n0 + t * 32 + g * 8 + half * 4;
                    char* tp = tile + (rg * 32 + l32) * a.pitch + n * 2;
                    float v[4];
                    uint2 prev = make_uint2(0u, 0u);
                    if (!a.Wt2) prev = *reinterpret_cast<const uint2*>(tp);   // the aggregate itself (feat == N): added, not transformed
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float x = acc[rg][t][g * 4 + i];
                        if (!a.Wt2) x += i < 2 ? ((i & 1) ? bf16_hi(prev.x) : bf16_lo(prev.x)) : ((i & 1) ? bf16_hi(prev.y) : bf16_lo(prev.y));
                        if (a.bias && n + i < a.N) x += a.bias[n + i];
                        if (a.relu) x = fmaxf(x, 0.0f);
                        v[i] = x;
                    }
                    *reinterpret_cast<uint2*>(tp) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
                }
            }
        }
    }
    __syncthreads();
    const int vecs = (a.N + 7) / 8;
    const bool vec_rows = (a.ldo & 7) == 0 && (reinterpret_cast<uintptr_t>(a.out) & 15u) == 0;
    for (int idx = tid; idx < kTileRows * vecs; idx += kBlock) {
        const int r = idx / vecs, n = (idx % vecs) * 8;
        const int64_t grow = row0 + r;
        if (grow >= a.n_rows) continue;
        const uint4 d = *reinterpret_cast<const uint4*>(tile + r * a.pitch + n * 2);
        bf16_t* o = a.out + grow * a.ldo + n;
        if (n + 8 <= a.N && vec_rows) {
            *reinterpret_cast<uint4*>(o) = d;
        } else {
            const uint32_t w[4] = {d.x, d.y, d.z, d.w};
            for (int i = 0; i < 8 && n + i < a.N; ++i) o[i] = (bf16_t)((w[i >> 1] >> ((i & 1) * 16)) & 0xffffu);
        }
    }
}

template <int LPR, int NTW>
static hipError_t launch_fused(const FusedSageArgs& a, dim3 grid, size_t lds, hipStream_t s) {
    if (a.val) hipLaunchKernelGGL((fused_sage_kernel<LPR, true, NTW, 8>), grid, dim3(kBlock), lds, s, a);
    else hipLaunchKernelGGL((fused_sage_kernel<LPR, false, NTW, 8>), grid, dim3(kBlock), lds, s, a);
    return hipGetLastError();
}

}  // namespace dgll

using namespace dgll;

DGLL_API int dgll_hip_sage_fused_forward(void* stream, const dgll_csr_plan* plan, const int64_t* rowptr, const int32_t* col,
                                         const float* val, const void* X, int64_t ldx, int feat, int reduce,
                                         const void* A1, int64_t lda1, int K1, const void* Wt1, int64_t ldw1,
                                         const void* Wt2, int64_t ldw2, int w_rows, const float* bias, int relu, void* out,
                                         int64_t ldo, int N, void* agg_out, int64_t ldagg, int64_t n_rows, int64_t n_cols,
                                         void* workspace, size_t workspace_bytes) {
    DGLL_REQUIRE(n_rows >= 0 && n_cols >= 0, "negative size");
    if (n_rows == 0) return DGLL_OK;
    DGLL_REQUIRE(rowptr && col && X && out, "NULL argument");
    DGLL_REQUIRE(feat > 0 && feat <= 256 && N > 0 && N <= 256, "the fused kernel handles up to 256 aggregate / output columns");
    DGLL_REQUIRE(reduce == DGLL_REDUCE_SUM || reduce == DGLL_REDUCE_MEAN, "reduce");
    DGLL_REQUIRE(A1 || Wt2, "nothing to transform: give a self operand and / or weights for the aggregate");
    DGLL_REQUIRE(!A1 || (Wt1 && K1 > 0 && lda1 >= K1), "self operand needs its weights");
    DGLL_REQUIRE(Wt2 || feat == N, "without weights the aggregate is added to the output: feat must equal N");
    DGLL_REQUIRE(w_rows >= ((N + 31) & ~31), "weight matrices must be zero-padded to whole 32-row tiles of N");
    const int kagg = (feat + 63) & ~63;
    DGLL_REQUIRE(!Wt1 || ldw1 >= ((K1 + 63) & ~63), "Wt1 must be zero-padded to a multiple of 64 columns");
    DGLL_REQUIRE(!Wt2 || ldw2 >= kagg, "Wt2 must be zero-padded to a multiple of 64 columns");
    DGLL_REQUIRE(aligned16(X) && (ldx * 2) % 16 == 0 && ldx >= feat && ldx < ((int64_t)1 << 31) && n_cols < ((int64_t)1 << 31),
                 "X rows must be 16-byte aligned");
    DGLL_REQUIRE(!A1 || (aligned16(A1) && (lda1 * 2) % 16 == 0), "self operand rows must be 16-byte aligned");
    DGLL_REQUIRE((!Wt1 || (aligned16(Wt1) && (ldw1 * 2) % 16 == 0)) && (!Wt2 || (aligned16(Wt2) && (ldw2 * 2) % 16 == 0)),
                 "weights must be 16-byte aligned");
    DGLL_REQUIRE(!agg_out || (aligned16(agg_out) && (ldagg * 2) % 16 == 0 && ldagg >= feat), "agg_out rows must be 16-byte aligned");
    DGLL_REQUIRE(ldo >= N, "leading dimension smaller than N");
    hipStream_t s = static_cast<hipStream_t>(stream);
    FusedSageArgs a{};
    a.rowptr = rowptr; a.col = col; a.val = val; a.X = static_cast<const bf16_t*>(X); a.ldx = ldx; a.feat = feat; a.kagg = kagg;
    a.reduce = reduce; a.threshold = 0;
    a.agg_out = static_cast<bf16_t*>(agg_out); a.ldagg = ldagg;
    a.A1 = static_cast<const bf16_t*>(A1); a.lda1 = lda1; a.K1 = A1 ? K1 : 0;
    a.Wt1 = static_cast<const bf16_t*>(Wt1); a.ldw1 = ldw1; a.Wt2 = static_cast<const bf16_t*>(Wt2); a.ldw2 = ldw2;
    a.bias = bias; a.out = static_cast<bf16_t*>(out); a.ldo = ldo; a.N = N; a.relu = relu; a.n_rows = n_rows;
    if (plan && plan->n_long > 0) {
        // rows above the plan's threshold: chunked over many wavefronts by the SpMM's long-row path, straight into agg_out
        DGLL_REQUIRE(plan->n_rows == n_rows, "plan was built for a different CSR");
        DGLL_REQUIRE(agg_out, "this graph has rows longer than the plan's threshold: the fused kernel needs agg_out for them");
        int rc = dgll_spmm_csr_impl(stream, plan, rowptr, col, val, X, ldx, DGLL_BF16, agg_out, ldagg, DGLL_BF16, n_rows, n_cols, feat,
                                    reduce, 0, nullptr, workspace, workspace_bytes, nullptr, 0, nullptr, 0, 1);
        if (rc != DGLL_OK) return rc;
        a.threshold = plan->threshold;
    }
    const int n_pad = (N + 31) & ~31;
    a.pitch = std::max(kagg, n_pad) * 2 + 16;
    const size_t lds = (size_t)kTileRows * a.pitch;
    const int vecs = (feat + 7) / 8;
    int lpr = 8;
    while (lpr < 32 && lpr < vecs) lpr <<= 1;
    dim3 grid((uint32_t)((n_rows + kTileRows - 1) / kTileRows));
    const int ntw = n_pad > 128 ? 2 : 1;
    hipError_t err;
#define DGLL_FUSED(L) (ntw == 2 ? launch_fused<L, 2>(a, grid, lds, s) : launch_fused<L, 1>(a, grid, lds, s))
    if (lpr == 8) err = DGLL_FUSED(8);
    else if (lpr == 16) err = DGLL_FUSED(16);
    else err = DGLL_FUSED(32);
#undef DGLL_FUSED
    if (err != hipSuccess) return hip_fail(err, "fused_sage_kernel launch");
    return DGLL_OK;
}
