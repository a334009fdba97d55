// dense.hip -- dense transforms next to the aggregation: C[M,N] = act(A[M,K] . B[K,N] + bias).
//
// Reference call sites: F.mm(x, weight) gcnconv.py:30, F.matmul sageconv.py:41,72, F.mm gatconv.py:31,117, and the
// X.W inside FusedKernel/gcn_fused_kernel.cu:46-54.  This file holds the exact-fp32 kernel used by the C-ABI-only
// entry points (the fused GCN launcher); the Python layers use library GEMMs for forward/dX.
#include "common.hpp"

namespace dgll {

constexpr int TM = 64, TN = 64, TK = 16;   // 64x64 output tile per 256-thread block, 4x4 outputs per thread

// fp32 LDS-tiled GEMM, fmaf accumulation in k order (bit-stable), optional bias and ReLU epilogue.
// trans bit 0: A is stored [K, M] (use A^T); bit 1: B is stored [N, K] (use B^T).
__global__ __launch_bounds__(kBlock) void gemm_f32_kernel(const float* __restrict__ A, int64_t lda,
                                                          const float* __restrict__ B, int64_t ldb,
                                                          float* __restrict__ C, int64_t ldc, int64_t M, int N, int K,
                                                          const float* __restrict__ bias, int relu, int trans) {
    __shared__ float sA[TK][TM + 4];   // stored k-major so the inner product reads are conflict-free
    __shared__ float sB[TK][TN + 4];
    const int tx = threadIdx.x % 16, ty = threadIdx.x / 16;
    const int64_t m0 = (int64_t)blockIdx.x * TM;
    const int n0 = blockIdx.y * TN;
    float acc[4][4] = {};
    for (int k0 = 0; k0 < K; k0 += TK) {
        for (int i = threadIdx.x; i < TM * TK; i += kBlock) {   // A tile: TM rows x TK cols
            const int r = i / TK, c = i % TK;
            const int64_t gm = m0 + r;
            sA[c][r] = (gm < M && k0 + c < K) ? ((trans & 1) ? A[(int64_t)(k0 + c) * lda + gm] : A[gm * lda + k0 + c]) : 0.0f;
        }
        for (int i = threadIdx.x; i < TK * TN; i += kBlock) {   // B tile: TK rows x TN cols
            const int r = i / TN, c = i % TN;
            sB[r][c] = (k0 + r < K && n0 + c < N) ? ((trans & 2) ? B[(int64_t)(n0 + c) * ldb + k0 + r] : B[(int64_t)(k0 + r) * ldb + n0 + c]) : 0.0f;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < TK; ++k) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] = sA[k][ty * 4 + i]; b[i] = sB[k][tx * 4 + i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t gm = m0 + ty * 4 + i;
        if (gm >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gn = n0 + tx * 4 + j;
            if (gn >= N) continue;
            float v = acc[i][j];
            if (bias) v += bias[gn];
            if (relu) v = fmaxf(v, 0.0f);
            C[gm * ldc + gn] = v;
        }
    }
}

int launch_gemm_f32(hipStream_t s, const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc,
                    int64_t M, int N, int K, const float* bias, int relu, int trans) {
    if (M <= 0 || N <= 0) return DGLL_OK;
    dim3 grid((uint32_t)((M + TM - 1) / TM), (uint32_t)((N + TN - 1) / TN));
    hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(kBlock), 0, s, A, lda, B, ldb, C, ldc, M, N, K, bias, relu, trans);
    DGLL_HIP_TRY(hipGetLastError());
    return DGLL_OK;
}

}  // namespace dgll

using namespace dgll;

DGLL_API int dgll_hip_gemm_f32(void* stream, const float* A, int64_t lda, const float* B, int64_t ldb, float* C,
                               int64_t ldc, int64_t M, int N, int K, const float* bias, int relu) {
    DGLL_REQUIRE(M >= 0 && N >= 0 && K >= 0, "negative size");
    if (M == 0 || N == 0) return DGLL_OK;
    DGLL_REQUIRE(A && B && C, "NULL matrix");
    DGLL_REQUIRE(lda >= K && ldb >= N && ldc >= N, "leading dimension too small");
    return launch_gemm_f32(static_cast<hipStream_t>(stream), A, lda, B, ldb, C, ldc, M, N, K, bias, relu, 0);
}

// =====================================================================================================================
// bf16 MFMA transform:  out[M, N] = act( sum_s A_s[M, K_s] . Wt_s[N, K_s]^T )      (s = 1 or 2 operand pairs)
//
// The post-aggregation dense W-transform of the layers on matrix cores:
//   sageConv  act(src.W_s + agg.W_n)   sageconv.py:71-82   -> two operand pairs, ReLU fused, one pass over the outputs
//   gcnConv   x.W                      gcnconv.py:30       -> one pair
//   GAT       h = x.W                  gatconv.py:31,117   -> one pair
//   backward  dX = g.W^T               (autograd of the above) -> one pair with Wt := W, optional ReLU mask fused on g
//
// Shapes are tall-skinny (M ~ 1e6 rows, N, K <= a few hundred): HBM-bound (128 flop/B at N = K = 256 < the 400 flop/B
// MFMA ridge), so the kernel is organised around reading every activation row exactly once:
//   * workgroup = 4 waves = 128 rows x ALL N columns; wave w owns rows 32w..32w+31 and keeps its 32 x N fp32 tile in
//     accumulators (16 registers per 32x32 tile);
//   * v_mfma_f32_32x32x16_bf16 with the WEIGHT as the MFMA "A" operand and the ACTIVATIONS as "B": lane l then feeds 8
//     consecutive k of activation row l%32 -- a straight 16-byte global load, no LDS -- and receives, per tile, 4 x 4
//     consecutive output COLUMNS of that row, so the epilogue stores 8-byte bf16x4 pieces of a row, not a scatter;
//   * the weight chunk (N x 64 k, <= 32 KiB) is staged through LDS, double buffered, rows padded to 144 bytes
//     (conflict-free ds_read_b128); it is re-read from L2 per 128-row block (the weights are tiny and stay resident);
//   * the reduction index inside a 64-k chunk is permuted (lane half h takes k = 32h + 8kk + j) so each lane's four
//     16-byte activation loads per chunk are one contiguous 64-byte run; A and B use the same permutation.
// Wt must be zero-padded by the host to [32*NT rows (NT = 2, 4 or 8: 64 / 128 / 256), ceil(K/64)*64 columns]; activations need no padding (tail
// vectors are masked element-wise, so uninitialised pad columns can never inject NaNs).
namespace dgll {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;   // native vector: stays in registers across the loop

struct MfmaGemmArgs {
    const bf16_t* A[2];
    int64_t lda[2];
    int K[2];
    const bf16_t* Wt[2];
    int64_t ldw[2];
    const bf16_t* mask;     // optional [M, K[0]]: A[0] elements are zeroed where mask <= 0 (fused ReLU backward)
    int64_t ldm;
    void* out;
    int64_t ldo;
    int64_t M;
    int N, relu, out_f32, pairs;
    const float* bias;
    const bf16_t* out_gate; // optional [M, N]: outputs are zeroed where out_gate <= 0 (the consumer's ReLU backward)
    int64_t ldgate;
    const float* row_scale; // optional fp32 [M]: out = act(row_scale[m] * (A.W) + bias)
    int kperm;              // diagnostics: 0 = natural k order (shipped), 1 = the two halves of a row 64 bytes apart
};

constexpr int kChunkK = 64;                 // k per LDS stage
constexpr int kWPitch = kChunkK * 2 + 16;   // bytes per weight row in LDS (144: bank-conflict-free b128 reads)

__device__ __forceinline__ uint4 mask_tail(uint4 v, int valid) {  // keep the first `valid` (0..8) bf16 of a vector
    uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const int lo = 2 * d, hi = 2 * d + 1;
        uint32_t m = (lo < valid ? 0x0000ffffu : 0u) | (hi < valid ? 0xffff0000u : 0u);
        w[d] &= m;
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

__device__ __forceinline__ uint4 relu_mask(uint4 v, uint4 m) {  // zero bf16 lanes of v where m <= 0 (or NaN-free assumption)
    uint32_t a[4] = {v.x, v.y, v.z, v.w}, b[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const uint32_t lo_pos = ((b[d] & 0x8000u) == 0u && (b[d] & 0x7fffu) != 0u) ? 0x0000ffffu : 0u;
        const uint32_t hi_pos = ((b[d] & 0x80000000u) == 0u && (b[d] & 0x7fff0000u) != 0u) ? 0xffff0000u : 0u;
        a[d] &= (lo_pos | hi_pos);
    }
    return make_uint4(a[0], a[1], a[2], a[3]);
}

// Global loads of chunk c: the weight slab (NT vectors per thread) and this lane's four activation vectors.
// (kernel-argument arrays are selected with ternaries: a runtime index would move them to scratch)
template <int NT>
__device__ __forceinline__ void mfma_load_chunk(const MfmaGemmArgs& a, int c, int chunks0, int tid, int half, int64_t row_ld,
                                                u32x4_t (&wreg)[NT], uint4 (&areg)[4]) {
    const bool second = c >= chunks0;
    const int k0 = (second ? c - chunks0 : c) * kChunkK;
    const bf16_t* w = second ? a.Wt[1] : a.Wt[0];
    const int64_t ldw = second ? a.ldw[1] : a.ldw[0];
    const int64_t lda = second ? a.lda[1] : a.lda[0];
    const int K = second ? a.K[1] : a.K[0];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int v = tid + i * kBlock;               // 16-byte vector index inside the chunk: row n = v/8, part = v%8
        wreg[i] = *reinterpret_cast<const u32x4_t*>(w + (int64_t)(v >> 3) * ldw + k0 + (v & 7) * 8);
    }
    const bf16_t* x = (second ? a.A[1] : a.A[0]) + row_ld * lda;
    // natural MFMA k order: step kk takes k = kk*16 + half*8 .. +8, so the two lanes of a row read ADJACENT 16-byte pieces
    // (32 contiguous bytes per row per load instruction instead of two pieces 64 bytes apart)
    const int kb = k0 + half * (a.kperm ? 32 : 8);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const int k = kb + kk * (a.kperm ? 8 : 16);
        uint4 v = make_uint4(0, 0, 0, 0);
        if (k < K) {
            v = *reinterpret_cast<const uint4*>(x + k);
            if (!second && a.mask) v = relu_mask(v, *reinterpret_cast<const uint4*>(a.mask + row_ld * a.ldm + k));
            if (k + 8 > K) v = mask_tail(v, K - k);
        }
        areg[kk] = v;
    }
}

template <int NT>
__device__ __forceinline__ void mfma_stage_chunk(char* base, int tid, const u32x4_t (&wreg)[NT]) {
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int v = tid + i * kBlock;
        *reinterpret_cast<u32x4_t*>(base + (v >> 3) * kWPitch + (v & 7) * 16) = wreg[i];
    }
}

// two workgroups per CU (<= 256 registers per lane, 2 x 73 KiB of LDS): the second one's loads overlap the first one's MFMAs
template <int NT>
__global__ __launch_bounds__(kBlock, 2) void gemm_bf16_nt_kernel(const MfmaGemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int kBufBytes = NT * 32 * kWPitch;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l32 = lane & 31;
    const int64_t m0 = (int64_t)blockIdx.x * 128 + wave * 32;
    const int64_t row = m0 + l32;
    const int64_t row_ld = row < a.M ? row : a.M - 1;    // clamp loads, mask stores

    f32x16_t acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    const int chunks0 = (a.K[0] + kChunkK - 1) / kChunkK;
    const int chunks1 = a.pairs > 1 ? (a.K[1] + kChunkK - 1) / kChunkK : 0;
    const int n_chunks = chunks0 + chunks1;

    u32x4_t wreg[NT];   // weight chunk staging: NT x 16 bytes per thread
    uint4 areg[4];    // this lane's activations for one chunk (4 k-steps x 8 bf16)
    mfma_load_chunk<NT>(a, 0, chunks0, tid, half, row_ld, wreg, areg);
    mfma_stage_chunk<NT>(smem, tid, wreg);
    __syncthreads();
    for (int c = 0; c < n_chunks; ++c) {
        uint4 cur[4] = {areg[0], areg[1], areg[2], areg[3]};
        if (c + 1 < n_chunks) mfma_load_chunk<NT>(a, c + 1, chunks0, tid, half, row_ld, wreg, areg);   // in flight during the MFMAs
        const char* base = smem + (c & 1) * kBufBytes + l32 * kWPitch + half * (a.kperm ? 64 : 16);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const bf16x8_t xf = __builtin_bit_cast(bf16x8_t, cur[kk]);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const uint4 wv = *reinterpret_cast<const uint4*>(base + t * 32 * kWPitch + kk * (a.kperm ? 16 : 32));
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wv), xf, acc[t], 0, 0, 0);
            }
        }
        if (c + 1 < n_chunks) {
            mfma_stage_chunk<NT>(smem + ((c + 1) & 1) * kBufBytes, tid, wreg);   // the other buffer: idle during this chunk
            __syncthreads();
        }
    }

    // D[i][j]: j = lane%32 = activation row, i = (r&3) + 8*(r>>2) + 4*half = output column inside the tile
    const float rs = a.row_scale ? a.row_scale[row_ld] : 1.0f;
    if (!a.out_f32) {
        // bf16 output: each lane holds 4-column pieces of ONE row, so direct stores would scatter 8-byte pieces over 32
        // rows per instruction.  Transpose through LDS (the weight buffers are idle now): every wavefront parks its
        // 32 x (NT*32) tile, then writes -- and reads the optional gate -- as whole 16-byte vectors along the rows.
        constexpr int kOPitch = NT * 64 + 16;               // bytes per staged row (16-byte multiple, bank-skewed)
        __syncthreads();                                    // every wave is done with the weight buffers
        char* mine = smem + wave * 32 * kOPitch;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = t * 32 + g * 8 + half * 4;
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float x = acc[t][g * 4 + i] * rs;
                    if (a.bias && n + i < a.N) x += a.bias[n + i];
                    if (a.relu) x = fmaxf(x, 0.0f);
                    v[i] = x;
                }
                *reinterpret_cast<uint2*>(mine + l32 * kOPitch + n * 2) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
            }
        }
        __syncthreads();
        constexpr int kVecs = NT * 4;                       // 16-byte vectors per staged row
        const bool vec_rows = (a.ldo & 7) == 0 && (reinterpret_cast<uintptr_t>(a.out) & 15u) == 0;
#pragma unroll 4
        for (int it = 0; it < 32 * kVecs / kWave; ++it) {
            const int idx = it * kWave + lane;
            const int r = idx / kVecs, n = (idx % kVecs) * 8;
            const int64_t grow = m0 + r;
            if (grow >= a.M || n >= a.N) continue;
            uint4 d = *reinterpret_cast<const uint4*>(mine + r * kOPitch + n * 2);
            bf16_t* o = static_cast<bf16_t*>(a.out) + grow * a.ldo + n;
            if (n + 8 <= a.N && vec_rows) {
                if (a.out_gate) d = relu_mask(d, *reinterpret_cast<const uint4*>(a.out_gate + grow * a.ldgate + n));
                *reinterpret_cast<uint4*>(o) = d;
            } else {
                const uint32_t w[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if (n + e >= a.N) continue;
                    bf16_t b = (bf16_t)((w[e >> 1] >> ((e & 1) * 16)) & 0xffffu);
                    if (a.out_gate && !(bf16_to_f32(a.out_gate[grow * a.ldgate + n + e]) > 0.0f)) b = 0;
                    o[e] = b;
                }
            }
        }
        return;
    }
    if (row >= a.M) return;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = t * 32 + g * 8 + half * 4;
            if (n >= a.N) continue;
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float x = acc[t][g * 4 + i] * rs;
                if (a.bias && n + i < a.N) x += a.bias[n + i];
                if (a.relu) x = fmaxf(x, 0.0f);
                v[i] = x;
            }
            if (a.out_gate) {
                const bf16_t* gp = a.out_gate + row * a.ldgate + n;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (n + i < a.N && !(bf16_to_f32(gp[i]) > 0.0f)) v[i] = 0.0f;
            }
            float* o = static_cast<float*>(a.out) + row * a.ldo + n;
            if (n + 4 <= a.N && (a.ldo & 3) == 0) *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
            else
                for (int i = 0; i < 4; ++i) if (n + i < a.N) o[i] = v[i];
        }
    }
}

template <int NT>
static hipError_t launch_mfma(const MfmaGemmArgs& a, hipStream_t s) {
    const size_t lds = 2 * (size_t)NT * 32 * kWPitch;
    if (lds > 48 * 1024) {
        static hipError_t raised = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_nt_kernel<NT>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);   // once: keeps
        if (raised != hipSuccess) return raised;                            // launches free of non-stream calls (graph capture)
    }
    dim3 grid((uint32_t)((a.M + 127) / 128));
    hipLaunchKernelGGL((gemm_bf16_nt_kernel<NT>), grid, dim3(kBlock), lds, s, a);
    return hipGetLastError();
}

}  // namespace dgll

int g_tune_mfma_kperm = 0;   // set through dgll_hip_debug_tune(4, .)

static int transform_bf16_impl(void* stream, const void* A1, int64_t lda1, int K1, const void* Wt1, int64_t ldw1,
                               const void* A2, int64_t lda2, int K2, const void* Wt2, int64_t ldw2, int wt_rows,
                               const void* relu_mask, int64_t ldm, void* out, int64_t ldo, int out_dtype, int64_t M,
                               int N, int relu, const float* bias, const void* out_gate, int64_t ldgate,
                               const float* row_scale);

DGLL_API int dgll_hip_transform_bf16(void* stream, const void* A1, int64_t lda1, int K1, const void* Wt1, int64_t ldw1,
                                     const void* A2, int64_t lda2, int K2, const void* Wt2, int64_t ldw2, int wt_rows,
                                     const void* relu_mask, int64_t ldm, void* out, int64_t ldo, int out_dtype, int64_t M,
                                     int N, int relu, const float* bias) {
    return transform_bf16_impl(stream, A1, lda1, K1, Wt1, ldw1, A2, lda2, K2, Wt2, ldw2, wt_rows, relu_mask, ldm, out, ldo,
                               out_dtype, M, N, relu, bias, nullptr, 0, nullptr);
}

DGLL_API int dgll_hip_transform_bf16_gated(void* stream, const void* A1, int64_t lda1, int K1, const void* Wt1, int64_t ldw1,
                                           const void* A2, int64_t lda2, int K2, const void* Wt2, int64_t ldw2, int wt_rows,
                                           const void* relu_mask, int64_t ldm, void* out, int64_t ldo, int out_dtype,
                                           int64_t M, int N, int relu, const float* bias, const void* out_gate,
                                           int64_t ldgate, const float* row_scale) {
    return transform_bf16_impl(stream, A1, lda1, K1, Wt1, ldw1, A2, lda2, K2, Wt2, ldw2, wt_rows, relu_mask, ldm, out, ldo,
                               out_dtype, M, N, relu, bias, out_gate, ldgate, row_scale);
}

static int transform_bf16_impl(void* stream, const void* A1, int64_t lda1, int K1, const void* Wt1, int64_t ldw1,
                               const void* A2, int64_t lda2, int K2, const void* Wt2, int64_t ldw2, int wt_rows,
                               const void* relu_mask, int64_t ldm, void* out, int64_t ldo, int out_dtype, int64_t M,
                               int N, int relu, const float* bias, const void* out_gate, int64_t ldgate,
                               const float* row_scale) {
    DGLL_REQUIRE(M >= 0 && N >= 0 && K1 >= 0 && K2 >= 0, "negative size");
    if (M == 0 || N == 0) return DGLL_OK;
    DGLL_REQUIRE(A1 && Wt1 && out && K1 > 0, "NULL operand");
    DGLL_REQUIRE(N <= 256, "dgll_hip_transform_bf16 keeps all N <= 256 output columns of a row block in accumulators");
    DGLL_REQUIRE(wt_rows >= (N <= 64 ? 64 : N <= 128 ? 128 : 256),
                 "Wt must be zero-padded to 64 / 128 / 256 rows (the kernel stages whole 32-row tiles of it)");
    DGLL_REQUIRE(out_dtype == DGLL_F32 || out_dtype == DGLL_BF16, "out_dtype");
    MfmaGemmArgs a{};
    a.A[0] = static_cast<const bf16_t*>(A1); a.lda[0] = lda1; a.K[0] = K1; a.Wt[0] = static_cast<const bf16_t*>(Wt1); a.ldw[0] = ldw1;
    a.pairs = 1;
    if (A2) {
        DGLL_REQUIRE(Wt2 && K2 > 0, "second operand pair incomplete");
        a.A[1] = static_cast<const bf16_t*>(A2); a.lda[1] = lda2; a.K[1] = K2; a.Wt[1] = static_cast<const bf16_t*>(Wt2); a.ldw[1] = ldw2;
        a.pairs = 2;
    }
    for (int s = 0; s < a.pairs; ++s) {
        DGLL_REQUIRE(aligned16(a.A[s]) && (a.lda[s] * 2) % 16 == 0 && a.lda[s] >= ((a.K[s] + 7) / 8) * 8,
                     "activations must be 16-byte aligned with the leading dimension padded to 8 elements");
        DGLL_REQUIRE(aligned16(a.Wt[s]) && (a.ldw[s] * 2) % 16 == 0 && a.ldw[s] >= ((a.K[s] + 63) / 64) * 64,
                     "Wt must be zero-padded to a multiple of 64 columns");
    }
    if (relu_mask) DGLL_REQUIRE(aligned16(relu_mask) && (ldm * 2) % 16 == 0, "mask alignment");
    a.mask = static_cast<const bf16_t*>(relu_mask); a.ldm = ldm;
    a.out = out; a.ldo = ldo; a.M = M; a.N = N; a.relu = relu; a.out_f32 = out_dtype == DGLL_F32; a.bias = bias;
    DGLL_REQUIRE(!out_gate || (ldgate >= N && aligned16(out_gate) && (ldgate * 2) % 16 == 0),
                 "out_gate: bf16 [M, ldgate >= N], 16-byte aligned rows");
    a.out_gate = static_cast<const bf16_t*>(out_gate); a.ldgate = ldgate;
    a.row_scale = row_scale;
    a.kperm = g_tune_mfma_kperm;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nt = (N + 31) / 32;
    hipError_t e;
    if (nt <= 2) e = launch_mfma<2>(a, s);
    else if (nt <= 4) e = launch_mfma<4>(a, s);
    else e = launch_mfma<8>(a, s);
    if (e != hipSuccess) return hip_fail(e, "gemm_bf16_nt_kernel launch");
    return DGLL_OK;
}
