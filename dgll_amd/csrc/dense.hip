// dense.hip -- dense transforms next to the aggregation: C[M,N] = act(A[M,K] . B[K,N] + bias).
//
// Reference call sites: F.mm(x, weight) gcnconv.py:30, F.matmul sageconv.py:41,72, F.mm gatconv.py:31,117, and the
// X.W inside FusedKernel/gcn_fused_kernel.cu:46-54.  Every dense product of the Python layers runs here or in gradw.hip /
// gemm_f32.hip: the bf16 MFMA transforms (resident-weights persistent kernel and the 4-wavefront one), their gated / addend
// epilogues, the weight pack, and the exact-fp32 kernel behind the C-ABI-only fused GCN launcher.  No library GEMM is called on any
// GPU path (no Cijk_* kernel in profiles/r0*_bench_kernel_stats.csv).
#include <algorithm>

#include "common.hpp"

int g_tune_res_per_cu = 0;    // dgll_hip_debug_tune(11, v): workgroups per CU of the resident-weights transform (0 = default)

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
    int pad_store;          // bf16 output whose row padding [N, ldo) belongs to the output: whole 16-byte vectors are stored, zeros past N
    const float* bias;
    const bf16_t* out_gate; // optional [M, N]: outputs are zeroed where out_gate <= 0 (the consumer's ReLU backward)
    int64_t ldgate;
    const float* row_scale; // optional fp32 [M]: out = act(row_scale[m] * (A.W) + bias)
    int no_rotate;          // diagnostics: 1 = every block walks the reduction from chunk 0
    const bf16_t* addend;   // optional bf16 [M, N] added before the activation (the aggregated term of a transform-first layer)
    int64_t ldadd;
    void* out2;             // dual form (resident-weights kernel, COLSPLIT = 2): out = A[0].Wt[0]^T and out2 = A[0].Wt[1]^T, A[0] read once
    int64_t ldo2;
    // One bit per output element instead of a bf16 gate matrix (32 bytes per 256-column row against 512): word w of row i, bit b
    // = column 32 w + b is positive.  bits_out: written by the epilogue from the values it stores (the producer of an activation);
    // gate_bits: read INSTEAD of out_gate by the resident-weights kernel (the consumer's ReLU backward).
    const uint32_t* gate_bits;
    int64_t ld_gate_bits;   // words per row, a multiple of 4 (16-byte rows)
    uint32_t* bits_out;
    int64_t ld_bits_out;
#ifdef DGLL_RES_TRACE
    unsigned long long* trace;   // probe builds only (tools/probes/res_trace.hip): [wg < 16][wave 8][block < 4][16 events]
#endif
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

// the keep-condition of relu_mask as one bit per element: bit e (of the low byte) = element e of v is positive -- sign clear and
// magnitude non-zero, i.e. (int16) v > 0: packed max with 0 then packed min with 1 leaves the flag at bits 0 and 16 of a dword.
// (Inline asm: written with builtins, hipcc turns both helpers back into compare + select per element -- 3-4x the instructions,
// with wait states between the v_cmp and the v_cndmask; tools/probes/res_epi.hip counts the epilogue's cycles.)
__device__ __forceinline__ uint32_t pos_bits8(uint4 v) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    const uint32_t ones = 0x00010001u;                           // (a packed inline constant is not replicated into the high half)
    uint32_t f4 = 0;                                             // low elements' flags at bits 0 2 4 6, high elements' at 16 18 20 22
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        uint32_t t, f;
        asm("v_pk_max_i16 %0, %1, 0" : "=v"(t) : "v"(w[d]));
        asm("v_pk_min_u16 %0, %1, %2" : "=v"(f) : "v"(t), "s"(ones));
        f4 |= f << (2 * d);
    }
    return (f4 & 0x55u) | ((f4 >> 15) & 0xaau);                  // bit 16 + 2 d -> 2 d + 1
}

// zero element e of v where bit e of `bits` is clear (higher bits ignored): relu_mask with the gate as bits.  A signed one-bit field
// extract gives 0 / ~0; v_bfi merges the two halves' masks
__device__ __forceinline__ uint4 keep_bits8(uint4 v, uint32_t bits) {
    uint32_t a[4] = {v.x, v.y, v.z, v.w};
    const uint32_t low_half = 0x0000ffffu;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        uint32_t lo, hi, m;
        asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(lo) : "v"(bits), "n"(2 * d));
        asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(hi) : "v"(bits), "n"(2 * d + 1));
        asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(m) : "s"(low_half), "v"(lo), "v"(hi));
        a[d] &= m;
    }
    return make_uint4(a[0], a[1], a[2], a[3]);
}

__device__ __forceinline__ uint32_t quad_or(uint32_t v) {   // OR over the four lanes 4 q .. 4 q + 3 (two DPP quad permutations, no LDS)
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false);    // quad_perm [1, 0, 3, 2]
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false);    // quad_perm [2, 3, 0, 1]
    return v;
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
    const int kb = k0 + half * 8;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const int k = kb + kk * 16;
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
        const char* base = smem + (c & 1) * kBufBytes + l32 * kWPitch + half * 16;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const bf16x8_t xf = __builtin_bit_cast(bf16x8_t, cur[kk]);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const uint4 wv = *reinterpret_cast<const uint4*>(base + t * 32 * kWPitch + kk * 32);
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
                    if (a.addend && n + i < a.N) x += bf16_to_f32(a.addend[row_ld * a.ldadd + n + i]);
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
            const int n_store = a.pad_store ? (int)a.ldo : a.N;      // as in the resident-weights kernel's epilogue
            if (grow >= a.M || n >= n_store) continue;
            uint4 d = *reinterpret_cast<const uint4*>(mine + r * kOPitch + n * 2);
            bf16_t* o = static_cast<bf16_t*>(a.out) + grow * a.ldo + n;
            const bool pad_vec = a.pad_store && n + 8 > a.N && n + 8 <= a.ldo;
            if (pad_vec) d = mask_tail(d, a.N > n ? a.N - n : 0);
            if ((n + 8 <= a.N || pad_vec) && vec_rows) {
                if (a.out_gate) d = relu_mask(d, *reinterpret_cast<const uint4*>(a.out_gate + grow * a.ldgate + n));
                *reinterpret_cast<uint4*>(o) = d;
            } else if (n < a.N) {
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
                if (a.addend && n + i < a.N) x += bf16_to_f32(a.addend[row * a.ldadd + n + i]);
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

// ---- helpers of the persistent kernel below -------------------------------------------------------------------------
// Every vector-memory LOAD of its main loop is issued from inline assembly and waited for BY HAND: hipcc's own s_waitcnt
// insertion merges the pending-load state at the loop header and ends up waiting for (almost) everything in front of the
// first MFMA of an iteration, which serialises any prefetch.  An asm load is invisible to that pass; its destination
// registers are only touched again after the hand-placed wait that names them as read-write operands
// (cdna_hip_programming.md section 5.7, forms (ii)/(iii)).  (Round 2 also carried an 8-wave kernel that re-staged the
// weights per chunk behind barriers; the resident-weights kernel replaced it everywhere and it was deleted.)
typedef __attribute__((ext_vector_type(4))) int i32x4_t;

__device__ __forceinline__ bool g_rot_enabled(const MfmaGemmArgs& a) { return a.no_rotate == 0; }

// Buffer descriptor (V#) over `bytes` bytes at `base`: raw dword access, hardware range check -- a load past the end
// returns zeros, so ragged tails need no address clamping.  gfx9-family word 3: DATA_FORMAT = 32 (0x00020000).
__device__ __forceinline__ i32x4_t make_rsrc(const void* base, uint32_t bytes) {
    const uint64_t p = reinterpret_cast<uint64_t>(base);
    return i32x4_t{(int)(uint32_t)p, (int)(uint32_t)((p >> 32) & 0xffffu), (int)bytes, 0x00020000};
}

// 16 bytes from descriptor `rsrc` at byte offset voff (per lane) + soff (scalar) + IMM: one VGPR of address, no 64-bit
// pointer arithmetic in vector registers (a flat pointer per load costs two VGPRs each and the compiler hoists all of
// them -- one per chunk -- out of the persistent loop).
template <int IMM>
__device__ __forceinline__ void asm_bufload16(u32x4_t& dst, uint32_t voff, i32x4_t rsrc, uint32_t soff) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4" : "=v"(dst) : "v"(voff), "s"(rsrc), "s"(soff), "i"(IMM) : "memory");
}

// =====================================================================================================================
// Resident-weights variant ("res"): the weights of the WHOLE reduction live in LDS for the lifetime of a persistent
// workgroup (128 KiB, 128-byte rows XOR-swizzled: no padding), staged once; the main loop has no barrier and no weight
// traffic.  A wave owns RG row groups of 32 rows x NTW column tiles of 32 (RG = 2, NTW = 4: 64 rows x 128 columns, 128
// accumulator registers; every weight fragment read from LDS feeds RG MFMAs).  128 KiB of LDS holds NC x NWG x 128 bytes:
// all 256 columns for NC <= 4 chunks (two waves per row group, CS = 2), 128 columns for NC = 8 (COLSPLIT = 2: two
// workgroups, on the same XCD and walking the same row blocks, each produce half of the columns; the partner's activation
// reads hit that XCD's L2).  What measurement settled, in the order it was found (tools/probes/):
//   * the epilogue must not issue a vector-memory LOAD the compiler can see (bias, row scale, gate, addend): hipcc waits
//     for it with vmcnt(0) -- it cannot see the inline-asm activation loads queued ahead of it -- and drains the prefetch of
//     the next block (19 k of a block's 34 k cycles).  PLAIN epilogues read the bias from LDS and load nothing;
//   * activation loads in the MFMA fragment shape (lane = row: 32 rows x 32 bytes per instruction, four instructions per
//     128-byte line) cap the L1 path at 6.5-7 TB/s and get slower with every chunk in flight; every row block is read twice
//     here (two workgroups or two waves), hence the 3.3 TB/s plateau of every earlier variant.  Loads are "half-line shaped"
//     now (16 rows x 64 contiguous bytes per instruction, 10-12 TB/s at L1) and a wave-private 2 KiB LDS tile turns two
//     of them into the fragments;
//   * ring depth: one or two chunks.  More in flight only queues (D = 1 / 2 / 4 / 8: 0.95 / 1.07 / 1.09 / 1.11 ms with
//     fragment loads; 0.92 / 0.87 / 0.88 with half-line loads);
//   * LDS bandwidth is the next limit (weights 16 KiB + tile 8 KiB per wave and chunk at RG = 1: 83 % of 128 B/clk), which
//     is what RG = 2 is for.
// The epilogue transposes one 32 x 32 tile at a time through the same wave-private LDS tile: no barrier either.
// registers of a chunk (per ring slot): index 4 g + 2 h + j -- row group g, 32-k half h, rows 16 j ..
template <int OUTSTANDING, int NR>
__device__ __forceinline__ void res_wait(u32x4_t (&r)[NR]) {
    static_assert(NR == 4 || NR == 8, "one or two row groups per wave");
    static_assert(OUTSTANDING >= 0 && OUTSTANDING < 64, "vmcnt is a 6-bit counter");
    if constexpr (NR == 4)
        asm volatile("s_waitcnt vmcnt(%4)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : "i"(OUTSTANDING) : "memory");
    else
        asm volatile("s_waitcnt vmcnt(%8)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
                     : "i"(OUTSTANDING) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// ---- the ring ---------------------------------------------------------------------------------------------------------
// Every phase consumes one 64-k chunk from ring slot P % D and refills that slot with the chunk D phases ahead -- ACROSS row
// blocks: the tail phases of a block already fetch the head of the next one.  Slot arithmetic stays compile-time because the
// loop body is a "super-iteration" of Q x NC phases with D | Q x NC.
struct ResCtx {                 // wave-uniform (SGPRs) except a_voff
    const bf16_t* A[2];
    int64_t lda[2];
    uint32_t ldab[2], tail[2];  // bytes per row; bytes of a row that are read (K rounded up to 8 elements)
    uint32_t a_voff[2][4];      // per lane [operand][2 g + j]: row 32 g + 16 j + lane / 4 of the wave's rows, piece lane % 4 of a half line
    int64_t M, n_blocks, stride;
    int chunks0, K0, K1, rotate;
};

template <int ROWS>
__device__ __forceinline__ i32x4_t res_rsrc(const ResCtx& x, bool second, int64_t blk) {
    const bf16_t* A = second ? x.A[1] : x.A[0];
    const int64_t lda = second ? x.lda[1] : x.lda[0];
    const uint32_t ldab = second ? x.ldab[1] : x.ldab[0], tail = second ? x.tail[1] : x.tail[0];
    const bool live = blk < x.n_blocks;                 // past the end: a zero-byte descriptor -- the loads return zeros, no traffic
    const int64_t r0 = live ? blk * ROWS : 0;
    const int64_t left = x.M - r0;
    const uint32_t rows = (uint32_t)(left < ROWS ? left : ROWS);
    return make_rsrc(A + r0 * lda, live ? (rows - 1) * ldab + tail : 0u);
}

template <int NC>
__device__ __forceinline__ int res_rot(const ResCtx& x, int64_t blk, int c) {   // block blk walks the reduction from chunk blk % NC
    c += x.rotate ? (int)((uint32_t)blk % (uint32_t)NC) : 0;
    return c >= NC ? c - NC : c;
}

// register 4 g + 2 h + j of a chunk: for lane l, the 16-byte piece l % 4 of the 64-byte half h of row 32 g + 16 j + l / 4
template <int NR>
__device__ __forceinline__ void res_trim(const ResCtx& x, int cc, int lane, u32x4_t (&areg)[NR]) {   // zero what lies past K
    const bool second = cc >= x.chunks0;
    const int k0 = (second ? cc - x.chunks0 : cc) * kChunkK;
    const int K = second ? x.K1 : x.K0;
    if (k0 + kChunkK <= K) return;                    // wave-uniform: only an operand's last chunk can be ragged
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const uint4 v = mask_tail(make_uint4(areg[i][0], areg[i][1], areg[i][2], areg[i][3]), K - (k0 + ((i >> 1) & 1) * 32 + (lane & 3) * 8));
        areg[i] = u32x4_t{v.x, v.y, v.z, v.w};
    }
}

struct ResIssue { i32x4_t r; uint32_t voff[4], k0b; };
struct ResLane {             // per-lane coordinates + the wave's LDS tile (transposition buffer of the main loop, staging of the epilogue)
    int lane, half, l32, rgroup, n_col0;
    int tw_off[2], tr_off[2];   // tile byte offsets: this lane's two writes (j) and two fragment reads (e) of a 32-k half
    char* scratch;
    const float* bias_w;
};

template <int NC, int ROWS>
__device__ __forceinline__ ResIssue res_target(const ResCtx& x, int64_t blk, int c) {
    const int cc = res_rot<NC>(x, blk, c);
    const bool second = cc >= x.chunks0;
    ResIssue t;
    t.r = res_rsrc<ROWS>(x, second, blk);
#pragma unroll
    for (int i = 0; i < 4; ++i) t.voff[i] = second ? x.a_voff[1][i] : x.a_voff[0][i];
    t.k0b = (uint32_t)((second ? cc - x.chunks0 : cc) * kChunkK * 2);
    return t;
}

__device__ __forceinline__ void res_issue1(const ResIssue& t, int i, u32x4_t& dst) {   // register i = 4 g + 2 h + j
    const uint32_t voff = t.voff[(i >> 2) * 2 + (i & 1)];
    if ((i & 2) == 0) asm_bufload16<0>(dst, voff, t.r, t.k0b);
    else asm_bufload16<64>(dst, voff, t.r, t.k0b);
}

#ifdef DGLL_RES_TRACE
#define RES_TRACE(EV) do { if (a.trace && blockIdx.x < 16 && t_blk < 4 && lane == 0 && wave < 8)                                   \
        a.trace[(((int64_t)blockIdx.x * 8 + wave) * 4 + t_blk) * 16 + (EV)] = __builtin_readcyclecounter(); } while (0)
#else
#define RES_TRACE(EV) do { } while (0)
#endif

// One 32-k half of a chunk through the wave's LDS tile ([32 rows][64 bytes], 16-byte slot q of row r at q ^ ((r >> 2) & 3):
// conflict-free both ways -- tools/probes/lds_pattern.hip; (r >> 1) & 3 costs the b128 reads 4 extra cycles): the two landed registers of every row group go in, two MFMA fragments (lane = row) come out, and
// the registers are refilled by the loads of the chunk D phases ahead.  One wave, in-order LDS: write -> read -> next write
// needs no wait and no barrier.
template <int RG, int H>
__device__ __forceinline__ void res_transpose(const ResLane& L, u32x4_t (&areg)[4 * RG], const ResIssue& nxt, uint4 (&f)[RG][2]) {
#pragma unroll
    for (int g = 0; g < RG; ++g) {
        *reinterpret_cast<u32x4_t*>(L.scratch + L.tw_off[0]) = areg[4 * g + 2 * H];
        *reinterpret_cast<u32x4_t*>(L.scratch + L.tw_off[1]) = areg[4 * g + 2 * H + 1];
        f[g][0] = *reinterpret_cast<const uint4*>(L.scratch + L.tr_off[0]);
        f[g][1] = *reinterpret_cast<const uint4*>(L.scratch + L.tr_off[1]);
        res_issue1(nxt, 4 * g + 2 * H, areg[4 * g + 2 * H]);
        res_issue1(nxt, 4 * g + 2 * H + 1, areg[4 * g + 2 * H + 1]);
    }
}

template <int NTW, int RG, int H>
__device__ __forceinline__ void res_mfmas(const char* base, const int (&slot_off)[4], const uint4 (&f)[RG][2], f32x16_t (&acc)[RG][NTW]) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const bf16x8_t wv = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(base + t * 32 * 128 + slot_off[2 * H + e]));
#pragma unroll
            for (int g = 0; g < RG; ++g)
#if defined(DGLL_RES_ABL) && (DGLL_RES_ABL & 2)
                if (t == 0) acc[g][t][e] += __builtin_bit_cast(float, f[g][e].x) + __builtin_bit_cast(float, ((const uint4*)(base + t * 32 * 128 + slot_off[2 * H + e]))->x);
#else
                acc[g][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wv, __builtin_bit_cast(bf16x8_t, f[g][e]), acc[g][t], 0, 0, 0);
#endif
        }
    }
}

// phase P of the super-iteration that starts at row block blk0: chunk P % NC of block blk0 + (P / NC) stride.
// (Measured and dropped: software-pipelining the halves -- second half transposed before the first half's MFMAs, the next
// chunk's first half before the second half's -- was 5-7 % SLOWER; the two extra scheduling fences cost more than the LDS
// round trips they hide, which the other waves of the SIMD already cover.)
template <int NTW, int RG, int NC, int D, int ROWS, int P, bool WAIT = true>
__device__ __forceinline__ void res_phase(const MfmaGemmArgs& a, const ResCtx& x, int64_t blk0, const ResLane& L, const char* wlane,
                                          int chunk_bytes, f32x16_t (&acc)[RG][NTW], u32x4_t (&A)[D][4 * RG], const int (&slot_off)[4],
                                          int t_blk, int wave) {
    constexpr int S = P % D;
    const int lane = L.lane;
    const int cc = res_rot<NC>(x, blk0 + (P / NC) * x.stride, P % NC);
    res_trim<4 * RG>(x, cc, lane, A[S]);
    const ResIssue nxt = res_target<NC, ROWS>(x, blk0 + ((P + D) / NC) * x.stride, (P + D) % NC);
    const char* base = wlane + cc * chunk_bytes;
    uint4 f[RG][2];
    res_transpose<RG, 0>(L, A[S], nxt, f);
    res_mfmas<NTW, RG, 0>(base, slot_off, f, acc);
    res_transpose<RG, 1>(L, A[S], nxt, f);
    res_mfmas<NTW, RG, 1>(base, slot_off, f, acc);
    if constexpr (WAIT) {
        res_wait<4 * RG * (D - 1), 4 * RG>(A[(P + 1) % D]);   // the next phase's chunk has landed; D - 1 younger sets may still fly
        RES_TRACE(2 + P % NC);
    }
    (void)t_blk; (void)wave; (void)a;
}

template <int NTW, int RG, int NC, int D, int ROWS, int P, int PEND>
__device__ __forceinline__ void res_phases(const MfmaGemmArgs& a, const ResCtx& x, int64_t blk0, const ResLane& L, const char* wlane,
                                           int chunk_bytes, f32x16_t (&acc)[RG][NTW], u32x4_t (&A)[D][4 * RG], const int (&slot_off)[4],
                                           int t_blk, int wave) {
    if constexpr (P < PEND) {
        res_phase<NTW, RG, NC, D, ROWS, P>(a, x, blk0, L, wlane, chunk_bytes, acc, A, slot_off, t_blk, wave);
        res_phases<NTW, RG, NC, D, ROWS, P + 1, PEND>(a, x, blk0, L, wlane, chunk_bytes, acc, A, slot_off, t_blk, wave);
    }
}

// epilogue of one wave: 32 rows x NTW*32 columns from global column n0, one 32 x 32 tile at a time through `scratch`
// (wave-private: 32 rows x 80 bytes)
// PLAIN (bf16 output, no row scale / addend / gate): the epilogue issues NO vector-memory load.  That matters more than the
// arithmetic it saves: a load the compiler can see is waited for with the `s_waitcnt vmcnt` IT computes, and it knows nothing
// of the inline-asm prefetch of the next block queued ahead of that load -- so it emits vmcnt(0) and the whole prefetch is
// drained before the epilogue starts (measured with cycle stamps, tools/probes/res_trace.hip: 19 k of a block's 34 k cycles
// sat in that wait).  The bias therefore comes from LDS (staged once), never from global memory.
// EPI: 0 = the general epilogue, 1 = PLAIN, 2 = PLAIN + an output gate given as BITS (`gb`: per `it`, the four words of this wave's
// tiles for row (it * 64 + lane) / 4 of the row group -- fetched by the caller BEFORE the block's last phase, so nothing is waited
// for here).  Any of them also writes the sign bits of what it stores when a.bits_out is set (stores only).
template <int NTW, int EPI>
__device__ __forceinline__ void res_epilogue(const MfmaGemmArgs& a, char* scratch, const float* bias_w, f32x16_t (&acc)[NTW],
                                             int64_t m0, int64_t row, int64_t row_ld, int n0, int lane, int half, int l32,
                                             const uint4 (&gb)[2]) {
    constexpr bool PLAIN = EPI != 0;
    uint32_t my_word[2] = {0u, 0u};                             // bits_out: lane l keeps the word of tile l % 4 for its row
    float rs = 1.0f;
    if constexpr (!PLAIN) rs = a.row_scale ? a.row_scale[row_ld] : 1.0f;
    constexpr int kPitch = 80;                                  // 64 bytes of bf16 + 16: conflict-free 8-byte writes
    const bool vec_rows = (a.ldo & 7) == 0 && (reinterpret_cast<uintptr_t>(a.out) & 15u) == 0;
    // Non-plain epilogues: fetch what TWO tiles need from global memory up front -- their gate vectors
    // (row-store layout) and the addend pieces (accumulator layout, 8 bytes per lane and column group) -- with
    // unconditional, address-clamped loads: one wait for the lot instead of one exposed round trip per tile (and no
    // branch around a load, which would make hipcc wait per element).  Columns past N but inside the padded leading
    // dimension are read and never stored.
    bool gate_fast = false, add_fast = false;
    if constexpr (!PLAIN) {
        gate_fast = !a.out_f32 && a.out_gate && (a.ldgate & 7) == 0 && (reinterpret_cast<uintptr_t>(a.out_gate) & 15u) == 0;
        add_fast = !a.out_f32 && a.addend && (a.ldadd & 3) == 0 && (reinterpret_cast<uintptr_t>(a.addend) & 7u) == 0;
    }
    constexpr int TP = NTW >= 2 ? 2 : 1;                        // tiles fetched for at a time (registers: 2 x (8 + 8) per lane)
#pragma unroll
    for (int tp = 0; tp < NTW; tp += TP) {
    uint4 gate_v[TP][2];
    uint2 add_v[TP][4];
    if constexpr (!PLAIN) {
        if (gate_fast) {
#pragma unroll
            for (int u = 0; u < TP; ++u)
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int idx = it * kWave + lane;
                    const int64_t grow = m0 + (idx >> 2);
                    const int n = n0 + (tp + u) * 32 + (idx & 3) * 8;
                    gate_v[u][it] = *reinterpret_cast<const uint4*>(a.out_gate + (grow < a.M ? grow : a.M - 1) * a.ldgate +
                                                                    (n + 8 <= a.ldgate ? n : 0));
                }
        }
        if (add_fast) {
#pragma unroll
            for (int u = 0; u < TP; ++u)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = n0 + (tp + u) * 32 + g * 8 + half * 4;
                    add_v[u][g] = *reinterpret_cast<const uint2*>(a.addend + row_ld * a.ldadd + (n + 4 <= a.ldadd ? n : 0));
                }
        }
    }
#pragma unroll
    for (int u = 0; u < TP; ++u) {
        const int t = tp + u;
        const int nt0 = n0 + t * 32;
        // columns that may be written: with pad_store the row padding too (zeros) -- a 47-column row on a 128-byte pitch is then
        // written as whole lines with 16-byte stores (256 -> 47: 3.6 -> see DESIGN 4.3; seven 2-byte stores per row before)
        const int n_store = a.pad_store ? (int)a.ldo : a.N;
        if (nt0 >= n_store) continue;                           // wave-uniform
        if (!PLAIN && a.out_f32) {
            if (row < a.M) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = nt0 + g * 8 + half * 4;
                    if (n >= a.N) continue;
                    const float4 bv = *reinterpret_cast<const float4*>(bias_w + t * 32 + g * 8 + half * 4);
                    const float b4[4] = {bv.x, bv.y, bv.z, bv.w};
                    float v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float xv = acc[t][g * 4 + i] * rs + b4[i];
                        if (a.addend && n + i < a.N) xv += bf16_to_f32(a.addend[row * a.ldadd + n + i]);
                        if (a.relu) xv = fmaxf(xv, 0.0f);
                        if (a.out_gate && n + i < a.N && !(bf16_to_f32(a.out_gate[row * a.ldgate + n + i]) > 0.0f)) xv = 0.0f;
                        v[i] = xv;
                    }
                    float* o = static_cast<float*>(a.out) + row * a.ldo + n;
                    if (n + 4 <= a.N && (a.ldo & 3) == 0) *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
                    else
                        for (int i = 0; i < 4; ++i) if (n + i < a.N) o[i] = v[i];
                }
            }
            continue;
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int nl = g * 8 + half * 4;
            const int n = nt0 + nl;
            const float4 bv = *reinterpret_cast<const float4*>(bias_w + t * 32 + nl);
            const float b4[4] = {bv.x, bv.y, bv.z, bv.w};
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float xv = acc[t][g * 4 + i] * rs + b4[i];
                if constexpr (!PLAIN) {
                    if (add_fast) {
                        const uint32_t w = i < 2 ? add_v[u][g].x : add_v[u][g].y;
                        xv += __builtin_bit_cast(float, (i & 1) ? (w & 0xffff0000u) : (w << 16));
                    } else if (a.addend && n + i < a.N) {
                        xv += bf16_to_f32(a.addend[row_ld * a.ldadd + n + i]);
                    }
                }
                if (a.relu) xv = fmaxf(xv, 0.0f);
                v[i] = xv;
            }
            *reinterpret_cast<uint2*>(scratch + l32 * kPitch + nl * 2) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
        }
        __builtin_amdgcn_wave_barrier();                       // compiler fence only: one wave's LDS operations execute in order,
                                                               // so the reads below see the whole tile without a wait
        if constexpr (EPI == 0) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {                       // 32 rows x 4 vectors of 16 bytes = 128 vectors, two per lane
            const int idx = it * kWave + lane;
            const int r = idx >> 2, nl = (idx & 3) * 8;
            const int n = nt0 + nl;
            const int64_t grow = m0 + r;
            if (grow >= a.M || n >= n_store) continue;
            uint4 d = *reinterpret_cast<const uint4*>(scratch + r * kPitch + nl * 2);
            bf16_t* o = static_cast<bf16_t*>(a.out) + grow * a.ldo + n;
#if defined(DGLL_RES_ABL) && (DGLL_RES_ABL & 1)
            if (d.x != 0x12345678u) continue;                          // probe: no stores
#endif
            const bool pad_vec = a.pad_store && n + 8 > a.N && n + 8 <= a.ldo;    // a vector that reaches into the row padding
            if (pad_vec) d = mask_tail(d, a.N > n ? a.N - n : 0);
            if ((n + 8 <= a.N || pad_vec) && vec_rows) {
                if constexpr (!PLAIN) {
                    if (gate_fast) d = relu_mask(d, gate_v[u][it]);
                    else if (a.out_gate) d = relu_mask(d, *reinterpret_cast<const uint4*>(a.out_gate + grow * a.ldgate + n));
                }
                *reinterpret_cast<uint4*>(o) = d;
            } else if (n < a.N) {
                const uint32_t w[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if (n + e >= a.N) continue;
                    bf16_t b = (bf16_t)((w[e >> 1] >> ((e & 1) * 16)) & 0xffffu);
                    if constexpr (!PLAIN) { if (a.out_gate && !(bf16_to_f32(a.out_gate[grow * a.ldgate + n + e]) > 0.0f)) b = 0; }
                    o[e] = b;
                }
            }
        }
        } else {
#pragma unroll
        for (int it = 0; it < 2; ++it) {                       // 32 rows x 4 vectors of 16 bytes = 128 vectors, two per lane
            const int idx = it * kWave + lane;
            const int r = idx >> 2, nl = (idx & 3) * 8;
            const int n = nt0 + nl;
            const int64_t grow = m0 + r;
            const bool live = grow < a.M && n < n_store;
            if (!a.bits_out && !live) continue;                 // (with bits_out all 64 lanes stay: the quad OR below needs them)
            uint4 d = *reinterpret_cast<const uint4*>(scratch + r * kPitch + nl * 2);     // (any lane: the address is inside the tile)
            const bool pad_vec = a.pad_store && n + 8 > a.N && n + 8 <= a.ldo;    // a vector that reaches into the row padding
            if (pad_vec) d = mask_tail(d, a.N > n ? a.N - n : 0);
            if constexpr (EPI == 2) {
                const uint32_t gw = t == 0 ? gb[it].x : t == 1 ? gb[it].y : t == 2 ? gb[it].z : gb[it].w;
                d = keep_bits8(d, gw >> ((idx & 3) * 8));
            }
            if (a.bits_out) {                                   // wave-uniform
                const int valid = a.N - n;                      // elements of this vector inside the matrix
                uint32_t bits = pos_bits8(d) & (valid >= 8 ? 0xffu : valid > 0 ? (1u << valid) - 1u : 0u);
                if (!live) bits = 0;
                const uint32_t word = quad_or(bits << ((idx & 3) * 8));
                if ((lane & 3) == t) my_word[it] = word;
                if (!live) continue;
            }
            bf16_t* o = static_cast<bf16_t*>(a.out) + grow * a.ldo + n;
#if defined(DGLL_RES_ABL) && (DGLL_RES_ABL & 1)
            if (d.x != 0x12345678u) continue;                          // probe: no stores
#endif
            if ((n + 8 <= a.N || pad_vec) && vec_rows) {
                *reinterpret_cast<uint4*>(o) = d;
            } else if (n < a.N) {
                const uint32_t w[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if (n + e >= a.N) continue;
                    o[e] = (bf16_t)((w[e >> 1] >> ((e & 1) * 16)) & 0xffffu);
                }
            }
        }
        }
        __builtin_amdgcn_wave_barrier();                       // (same: the next tile's writes queue behind these reads)
    }
    }
    if (EPI != 0 && a.bits_out) {                               // 64 lanes x 4 bytes: 16 rows x the 16 contiguous bytes of this wave's tiles
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int idx = it * kWave + lane;
            const int64_t grow = m0 + (idx >> 2);
            const int wcol = (n0 >> 5) + (lane & 3);
            if (grow < a.M && wcol < a.ld_bits_out) a.bits_out[grow * a.ld_bits_out + wcol] = my_word[it];
        }
    }
}

// blocks QI .. Q-1 of a super-iteration; true = the last row block of this workgroup has been written
template <int NTW, int RG, int NC, int D, int ROWS, int Q, int EPI, int QI>
__device__ __forceinline__ bool res_blocks(const MfmaGemmArgs& a, const ResCtx& x, int64_t blk0, const ResLane& L, const char* wlane,
                                           int chunk_bytes, u32x4_t (&A)[D][4 * RG], const int (&slot_off)[4], int& t_blk, int wave) {
    if constexpr (QI < Q) {
        const int lane = L.lane;
        RES_TRACE(0);
        f32x16_t acc[RG][NTW];
#pragma unroll
        for (int g = 0; g < RG; ++g)
#pragma unroll
            for (int t = 0; t < NTW; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[g][t][r] = 0.0f;
        const int64_t blk = blk0 + QI * x.stride;
        // EPI 2: the gate words of the whole block (two 16-byte loads per row group and lane).  Every phase ends in a wait for ALL
        // loads but the youngest ones, so a load has to land within the phase that issues it -- and these miss to HBM while the
        // activation chunks mostly hit the L2 (the partner fetched them): issued at the start of the last phase they stretched it
        // by 2-3 k cycles.  They are issued at the END of the second-to-last phase instead, as the youngest loads, and that phase's
        // wait lets them fly: a whole phase to land, and the last phase's own wait covers them.
        uint4 gb[RG][2];
#pragma unroll
        for (int g = 0; g < RG; ++g) gb[g][0] = gb[g][1] = make_uint4(0u, 0u, 0u, 0u);
        auto fetch_gate_words = [&]() {
            int b_lane = L.lane, b_n0 = L.n_col0;
            asm volatile("" : "+v"(b_lane), "+s"(b_n0));        // (opaque: keeps the addresses out of the persistent loop's live set)
#pragma unroll
            for (int g = 0; g < RG; ++g)
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int64_t grow = blk * ROWS + (L.rgroup * RG + g) * 32 + ((it * kWave + b_lane) >> 2);
                    gb[g][it] = *reinterpret_cast<const uint4*>(a.gate_bits + (grow < a.M ? grow : a.M - 1) * a.ld_gate_bits + (b_n0 >> 5));
                }
        };
        if constexpr (EPI == 2 && NC >= 2) {
            res_phases<NTW, RG, NC, D, ROWS, QI * NC, QI * NC + NC - 2>(a, x, blk0, L, wlane, chunk_bytes, acc, A, slot_off, t_blk, wave);
            constexpr int P2 = QI * NC + NC - 2;
            res_phase<NTW, RG, NC, D, ROWS, P2, false>(a, x, blk0, L, wlane, chunk_bytes, acc, A, slot_off, t_blk, wave);
            fetch_gate_words();
            res_wait<4 * RG * (D - 1) + 2 * RG, 4 * RG>(A[(P2 + 1) % D]);
            RES_TRACE(2 + P2 % NC);
        } else {
            res_phases<NTW, RG, NC, D, ROWS, QI * NC, QI * NC + NC - 1>(a, x, blk0, L, wlane, chunk_bytes, acc, A, slot_off, t_blk, wave);
            if constexpr (EPI == 2) fetch_gate_words();
        }
        res_phases<NTW, RG, NC, D, ROWS, QI * NC + NC - 1, QI * NC + NC>(a, x, blk0, L, wlane, chunk_bytes, acc, A, slot_off, t_blk, wave);
        if constexpr (EPI == 2) {
            // The words have landed (the phase's own vmcnt(0)), but hipcc does not know: it would wait for each of the four loads at
            // its first use INSIDE the epilogue with the count of memory operations issued since -- and on gfx9 stores share that
            // counter, so every such wait also drains the epilogue's own stores (seen as vmcnt(2) / vmcnt(3) in front of the mask
            // arithmetic).  Touching the registers here makes it place those waits now, where they cost nothing.
#pragma unroll
            for (int g = 0; g < RG; ++g)
#pragma unroll
                for (int it = 0; it < 2; ++it)
                    asm volatile("" : "+v"(gb[g][it].x), "+v"(gb[g][it].y), "+v"(gb[g][it].z), "+v"(gb[g][it].w));
        }
        RES_TRACE(11);
#pragma unroll
        for (int g = 0; g < RG; ++g) {
            const int64_t m0 = blk * ROWS + (L.rgroup * RG + g) * 32;
            const int64_t row = m0 + L.l32;
            const int64_t row_ld = row < a.M ? row : a.M - 1;
            // opaque copies of the lane coordinates: what the epilogue derives from them (staging addresses, the store loop's
            // row / column pairs) would otherwise be hoisted out of the persistent loop and stay live through the main loop
            int e_lane = L.lane, e_half = L.half, e_l32 = L.l32, e_n0 = L.n_col0;
            asm volatile("" : "+v"(e_lane), "+v"(e_half), "+v"(e_l32), "+s"(e_n0));
            res_epilogue<NTW, EPI>(a, L.scratch, L.bias_w, acc[g], m0, row, row_ld, e_n0, e_lane, e_half, e_l32, gb[g]);
        }
        RES_TRACE(12);
        ++t_blk;
        if (blk + x.stride >= x.n_blocks) return true;
        return res_blocks<NTW, RG, NC, D, ROWS, Q, EPI, QI + 1>(a, x, blk0, L, wlane, chunk_bytes, A, slot_off, t_blk, wave);
    } else {
        return false;
    }
}

template <int RG, int NC, int D, int ROWS, int P>
__device__ __forceinline__ void res_prologue(const ResCtx& x, int64_t blk0, u32x4_t (&A)[D][4 * RG]) {
    if constexpr (P < D) {
        const ResIssue t = res_target<NC, ROWS>(x, blk0 + (P / NC) * x.stride, P % NC);
#pragma unroll
        for (int h = 0; h < 2; ++h)                      // queue order: (chunk, h0) before (chunk, h1), as the phases issue them
#pragma unroll
            for (int g = 0; g < RG; ++g)
#pragma unroll
                for (int j = 0; j < 2; ++j) res_issue1(t, 4 * g + 2 * h + j, A[P][4 * g + 2 * h + j]);
        res_prologue<RG, NC, D, ROWS, P + 1>(x, blk0, A);
    }
}

template <int NTW, int RG, int NC, int CS, int COLSPLIT, int D, int EPI, int NW, bool DUAL>
__global__ __launch_bounds__(NW * 64) void gemm_bf16_res_kernel(const MfmaGemmArgs a_in) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    static_assert(!DUAL || COLSPLIT == 2, "the dual form hands the second weight matrix to the second column share");
    MfmaGemmArgs a = a_in;
    if (DUAL) {                                         // share 1 computes the SAME rows against Wt[1] into out2: everything below
        if (((blockIdx.x >> 3) % COLSPLIT) == 1) {      // sees one ordinary single-pair product (block-uniform selects)
            a.Wt[0] = a.Wt[1]; a.ldw[0] = a.ldw[1]; a.out = a.out2; a.ldo = a.ldo2;
        }
        a.pairs = 1;
    }
    static_assert(D % NC == 0 || NC % D == 0, "ring slots must line up from one super-iteration to the next");
    constexpr int Q = D > NC ? D / NC : 1;              // row blocks per super-iteration
    constexpr int NWG_T = NTW * CS;                     // 32-column tiles this workgroup produces
    constexpr int kRows = NW * 32 * RG / CS;            // rows per block
    constexpr int kChunkBytes = NWG_T * 32 * 128;       // one 64-k chunk of this workgroup's weight rows
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l32 = lane & 31;
    const int rgroup = wave / CS, t0 = (wave % CS) * NTW;
    // partners (the COLSPLIT column shares of the same rows) sit on the same XCD: block b runs on XCD b % 8, so the partner
    // of b is b + 8.  pair = which sequence of row blocks, share = which columns.
    const int bid = blockIdx.x;
    const int share = COLSPLIT == 1 ? 0 : (bid >> 3) % COLSPLIT;
    const int pair = COLSPLIT == 1 ? bid : (bid >> 3) / COLSPLIT * 8 + (bid & 7);
    const int n_pairs = COLSPLIT == 1 ? (int)gridDim.x : (int)gridDim.x / COLSPLIT;
    const int n_wg0 = DUAL ? 0 : share * NWG_T * 32;    // first global column of this workgroup
    const int chunks0 = (a.K[0] + kChunkK - 1) / kChunkK;

    // ---- stage ALL weights of this workgroup's columns once: [chunk][row n][8 slots of 16 bytes, slot ^ ((n >> 1) & 7)]
    for (int c = 0; c < NC; ++c) {
        const bool second = c >= chunks0;
        const bf16_t* w = second ? a.Wt[1] : a.Wt[0];
        const int64_t ldw = second ? a.ldw[1] : a.ldw[0];
        const int k0 = (second ? c - chunks0 : c) * kChunkK;
        for (int v = tid; v < NWG_T * 32 * 8; v += NW * 64) {
            const int nl = v >> 3, slot = v & 7;
            const u32x4_t val = *reinterpret_cast<const u32x4_t*>(w + (int64_t)(n_wg0 + nl) * ldw + k0 + slot * 8);
            *reinterpret_cast<u32x4_t*>(smem + c * kChunkBytes + nl * 128 + ((slot ^ ((nl >> 1) & 7)) * 16)) = val;
        }
    }
    float* bias_l = reinterpret_cast<float*>(smem + NC * kChunkBytes + NW * 32 * 80);   // this workgroup's columns (zeros: no bias)
    for (int n = tid; n < NWG_T * 32; n += NW * 64) bias_l[n] = (a.bias && n_wg0 + n < a.N) ? a.bias[n_wg0 + n] : 0.0f;
    __syncthreads();                                    // the only barrier of the kernel; every staging load has landed
    const char* wlane = smem + (t0 * 32 + l32) * 128;
    const int swz = (l32 >> 1) & 7;                      // (tile * 32 + l32) >> 1 & 7 == (l32 >> 1) & 7
    int slot_off[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) slot_off[kk] = ((2 * kk + half) ^ swz) * 16;

    ResCtx x;
    x.A[0] = a.A[0]; x.A[1] = a.pairs > 1 ? a.A[1] : a.A[0];
    x.lda[0] = a.lda[0]; x.lda[1] = a.pairs > 1 ? a.lda[1] : a.lda[0];
    x.ldab[0] = (uint32_t)(x.lda[0] * 2); x.ldab[1] = (uint32_t)(x.lda[1] * 2);
    x.tail[0] = (uint32_t)((a.K[0] + 7) / 8) * 16; x.tail[1] = (uint32_t)(((a.pairs > 1 ? a.K[1] : a.K[0]) + 7) / 8) * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) {                        // i = 2 g + j
        const uint32_t rin = (uint32_t)((rgroup * RG + (i >> 1)) * 32 + 16 * (i & 1) + (lane >> 2));
        x.a_voff[0][i] = rin * x.ldab[0] + (uint32_t)(lane & 3) * 16;
        x.a_voff[1][i] = rin * x.ldab[1] + (uint32_t)(lane & 3) * 16;
    }
    x.M = a.M; x.n_blocks = (a.M + kRows - 1) / kRows; x.stride = n_pairs;
    x.chunks0 = chunks0; x.K0 = a.K[0]; x.K1 = a.K[1]; x.rotate = g_rot_enabled(a) ? 1 : 0;

    ResLane L;
    L.lane = lane; L.half = half; L.l32 = l32; L.rgroup = rgroup; L.n_col0 = n_wg0 + t0 * 32;
    L.scratch = smem + NC * kChunkBytes + wave * (32 * 80);
    L.bias_w = bias_l + t0 * 32;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = 16 * j + (lane >> 2);
        L.tw_off[j] = r * 64 + (((lane & 3) ^ ((r >> 2) & 3)) * 16);
        L.tr_off[j] = l32 * 64 + (((2 * j + half) ^ ((l32 >> 2) & 3)) * 16);
    }

    int64_t blk = pair;
    if (blk >= x.n_blocks) return;                      // more workgroups than row blocks (tiny M): nothing to do
    u32x4_t A[D][4 * RG];
    res_prologue<RG, NC, D, kRows, 0>(x, blk, A);
    res_wait<4 * RG * (D - 1), 4 * RG>(A[0]);
    int t_blk = 0;
    for (;;) {
        if (res_blocks<NTW, RG, NC, D, kRows, Q, EPI, 0>(a, x, blk, L, wlane, kChunkBytes, A, slot_off, t_blk, wave)) break;
        blk += Q * x.stride;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the loads issued past the last block (zero-byte descriptors)
}

template <int NTW, int NC, int CS, int COLSPLIT, int EPI, bool DUAL = false>
static hipError_t launch_mfma_res_p(const MfmaGemmArgs& a, hipStream_t s);

// which epilogue: 1 (plain) loads nothing; 2 = plain + the output gate as bits (fetched a phase ahead); 0 = everything else
static int res_epilogue_kind(const MfmaGemmArgs& a) {
    const bool simple = !a.out_f32 && !a.row_scale && !a.addend;
    if (simple && !a.out_gate && !a.gate_bits) return 1;
    if (simple && a.gate_bits && (a.ldo & 7) == 0 && (reinterpret_cast<uintptr_t>(a.out) & 15u) == 0) return 2;
    return 0;
}

template <int NTW, int NC, int CS, int COLSPLIT>
static hipError_t launch_mfma_res(const MfmaGemmArgs& a, hipStream_t s) {
    switch (res_epilogue_kind(a)) {
        case 1: return launch_mfma_res_p<NTW, NC, CS, COLSPLIT, 1>(a, s);
        case 2: return launch_mfma_res_p<NTW, NC, CS, COLSPLIT, 2>(a, s);
        default: return launch_mfma_res_p<NTW, NC, CS, COLSPLIT, 0>(a, s);
    }
}

template <int NTW, int NC, int CS, int COLSPLIT, int EPI, bool DUAL>
static hipError_t launch_mfma_res_p(const MfmaGemmArgs& a, hipStream_t s) {
    // ring depth: D | NC or NC | D (slot arithmetic)
#ifdef DGLL_RES_D
    constexpr int D = (NC % DGLL_RES_D == 0 || DGLL_RES_D % NC == 0) ? DGLL_RES_D : NC;    // probe builds: forced ring depth
#else
    constexpr int D = 1;      // measured (tools/probes/res_trace.hip): more chunks in flight only queue -- see the header above; round 3,
                              // narrow outputs with the whole reduction of a row block in flight (D = NC): 256 -> 47 0.317 -> 0.34 ms
#endif
#ifdef DGLL_RES_RG
    constexpr int RG = DGLL_RES_RG;
#else
    constexpr int RG = 2;
#endif
#ifdef DGLL_RES_NW
    constexpr int NW = DGLL_RES_NW;
#else
    constexpr int NW = 8;
#endif
    constexpr int NWG_T = NTW * CS;
    const size_t lds = (size_t)NC * NWG_T * 32 * 128 + NW * 32 * 80 + NWG_T * 32 * 4;
    auto kern = &gemm_bf16_res_kernel<NTW, RG, NC, CS, COLSPLIT, D, EPI, NW, DUAL>;
    // both the LDS attribute and the CU count are per device (single-process multi-device use): cached per device id
    static bool raised_on[64] = {};
    static int n_cu_of[64] = {};
    int dev = 0;
    hipError_t ge = hipGetDevice(&dev);
    if (ge != hipSuccess) return ge;
    const int slot = (dev >= 0 && dev < 64) ? dev : 0;
    if (!raised_on[slot] || slot != dev) {
        hipError_t re = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (re != hipSuccess) return re;
        raised_on[slot] = slot == dev;
    }
    if (n_cu_of[slot] == 0 || slot != dev) {
        hipDeviceProp_t prop;
        int n = 0;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
        n = n / 16 * 16;                                    // whole groups of 8 XCDs x COLSPLIT partners
        if (n <= 0) n = 16;
        n_cu_of[slot] = n;
    }
    const int n_cu = n_cu_of[slot];
    int per_cu = lds > 80 * 1024 ? 1 : 2;
    if (g_tune_res_per_cu > 0 && (size_t)g_tune_res_per_cu * lds <= 160 * 1024) per_cu = g_tune_res_per_cu;   // diagnostics
    dim3 grid((uint32_t)(n_cu * per_cu));
    hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds, s, a);
    return hipGetLastError();
}

// resident-weights kernel for this shape?  nt = 32-column tiles of N, n_chunks = 64-k chunks of K1 + K2
static bool res_applies(int nt, int n_chunks) {
    if (n_chunks < 1 || n_chunks > 8) return false;
    if (nt > 4 && n_chunks > 4) return nt <= 8;            // 256 columns x 512 k: two workgroups of 128 columns
    return (size_t)n_chunks * (nt <= 2 ? 2 : nt <= 4 ? 4 : 8) * 32 * 128 <= 128 * 1024;
}

static hipError_t launch_mfma_res_dispatch(const MfmaGemmArgs& a, int nt, int n_chunks, hipStream_t s) {
#ifdef DGLL_RES_PROBE          // probe builds instantiate the few kernels they launch themselves (3 minutes of compile time otherwise)
    (void)a; (void)nt; (void)n_chunks; (void)s;
    return hipErrorNotSupported;
#else
#define RES_NC(NTW, CS, COLSPLIT)                                                      \
    switch (n_chunks) {                                                                \
        case 1: return launch_mfma_res<NTW, 1, CS, COLSPLIT>(a, s);                    \
        case 2: return launch_mfma_res<NTW, 2, CS, COLSPLIT>(a, s);                    \
        case 3: return launch_mfma_res<NTW, 3, CS, COLSPLIT>(a, s);                    \
        case 4: return launch_mfma_res<NTW, 4, CS, COLSPLIT>(a, s);                    \
        case 5: return launch_mfma_res<NTW, 5, CS, COLSPLIT>(a, s);                    \
        case 6: return launch_mfma_res<NTW, 6, CS, COLSPLIT>(a, s);                    \
        case 7: return launch_mfma_res<NTW, 7, CS, COLSPLIT>(a, s);                    \
        default: return launch_mfma_res<NTW, 8, CS, COLSPLIT>(a, s);                   \
    }
    if (nt <= 2) { RES_NC(2, 1, 1) }                        // N <= 64: one wave per row group, all columns
    if (nt <= 4) { RES_NC(4, 1, 1) }                        // N <= 128
    if (n_chunks <= 4) { RES_NC(4, 2, 1) }                  // N <= 256, K <= 256: two waves per row group
    RES_NC(4, 1, 2)                                         // N <= 256, K <= 512: two workgroups per row block
#undef RES_NC
#endif
}

}  // namespace dgll

int g_tune_mfma_kperm = 0;   // dgll_hip_debug_tune(4, v): 0 = per-shape choice (shipped), 1 = 4-wave kernel always, 2 = no chunk rotation

// ---- weight packing: [n, k] fp32 / bf16 with arbitrary element strides (a parameter or its transposed view) -> the zero-padded
// bf16 [rows, ld] block the transform kernels stage.  ONE launch where the wrappers used to cast, zero-fill and copy (three).
namespace dgll {
// word w of row i <- the sign bits (positive = 1) of out[i, 32 w .. 32 w + 31], zeros past N and in the padding words
__global__ __launch_bounds__(kBlock) void sign_bits_kernel(const bf16_t* __restrict__ out, int64_t ldo, int64_t M, int N,
                                                           uint32_t* __restrict__ bits, int64_t ldb) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= M * ldb) return;
    const int64_t r = i / ldb;
    const int n0 = (int)(i % ldb) * 32;
    uint32_t w = 0;
    for (int e = 0; e < 32 && n0 + e < N; ++e) {
        const uint32_t v = out[r * ldo + n0 + e];
        if ((v & 0x8000u) == 0u && (v & 0x7fffu) != 0u) w |= 1u << e;
    }
    bits[i] = w;
}

template <typename T>
__global__ __launch_bounds__(kBlock) void pack_weight_kernel(const T* __restrict__ src, int64_t sr, int64_t sc, int n, int k,
                                                             bf16_t* __restrict__ dst, int64_t ld, int rows) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= (int64_t)rows * ld) return;
    const int r = (int)(i / ld), c = (int)(i % ld);
    float v = 0.0f;
    if (r < n && c < k) {
        if constexpr (sizeof(T) == 2) v = bf16_to_f32(src[r * sr + c * sc]);
        else v = src[r * sr + c * sc];
    }
    dst[i] = f32_to_bf16(v);
}
}  // namespace dgll

DGLL_API int dgll_hip_pack_weight_bf16(void* stream, const void* src, int src_dtype, int64_t stride_row, int64_t stride_col, int n,
                                       int k, void* dst, int64_t ld, int rows) {
    DGLL_REQUIRE(n >= 0 && k >= 0 && rows >= n && ld >= k, "shape");
    if (rows == 0 || ld == 0) return DGLL_OK;
    DGLL_REQUIRE(dst && (src || n == 0 || k == 0), "NULL argument");
    DGLL_REQUIRE(src_dtype == DGLL_F32 || src_dtype == DGLL_BF16, "src_dtype");
    const int64_t total = (int64_t)rows * ld;
    const dim3 grid((uint32_t)((total + kBlock - 1) / kBlock));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (src_dtype == DGLL_F32)
        hipLaunchKernelGGL((dgll::pack_weight_kernel<float>), grid, dim3(kBlock), 0, s, static_cast<const float*>(src), stride_row,
                           stride_col, n, k, static_cast<bf16_t*>(dst), ld, rows);
    else
        hipLaunchKernelGGL((dgll::pack_weight_kernel<bf16_t>), grid, dim3(kBlock), 0, s, static_cast<const bf16_t*>(src), stride_row,
                           stride_col, n, k, static_cast<bf16_t*>(dst), ld, rows);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "pack_weight_kernel launch");
    return DGLL_OK;
}

static int transform_bf16_impl(void* stream, const void* A1, int64_t lda1, int K1, const void* Wt1, int64_t ldw1,
                               const void* A2, int64_t lda2, int K2, const void* Wt2, int64_t ldw2, int wt_rows,
                               const void* relu_mask, int64_t ldm, void* out, int64_t ldo, int out_dtype, int64_t M,
                               int N, int relu, const float* bias, const void* out_gate, int64_t ldgate,
                               const float* row_scale, const void* addend, int64_t ldadd, const uint32_t* gate_bits = nullptr,
                               int64_t ld_gate_bits = 0, uint32_t* bits_out = nullptr, int64_t ld_bits_out = 0);

DGLL_API int dgll_hip_transform_bf16(void* stream, const void* A1, int64_t lda1, int K1, const void* Wt1, int64_t ldw1,
                                     const void* A2, int64_t lda2, int K2, const void* Wt2, int64_t ldw2, int wt_rows,
                                     const void* relu_mask, int64_t ldm, void* out, int64_t ldo, int out_dtype, int64_t M,
                                     int N, int relu, const float* bias) {
    return transform_bf16_impl(stream, A1, lda1, K1, Wt1, ldw1, A2, lda2, K2, Wt2, ldw2, wt_rows, relu_mask, ldm, out, ldo,
                               out_dtype, M, N, relu, bias, nullptr, 0, nullptr, nullptr, 0);
}

DGLL_API int dgll_hip_transform_bf16_gated(void* stream, const void* A1, int64_t lda1, int K1, const void* Wt1, int64_t ldw1,
                                           const void* A2, int64_t lda2, int K2, const void* Wt2, int64_t ldw2, int wt_rows,
                                           const void* relu_mask, int64_t ldm, void* out, int64_t ldo, int out_dtype,
                                           int64_t M, int N, int relu, const float* bias, const void* out_gate,
                                           int64_t ldgate, const float* row_scale) {
    return transform_bf16_impl(stream, A1, lda1, K1, Wt1, ldw1, A2, lda2, K2, Wt2, ldw2, wt_rows, relu_mask, ldm, out, ldo,
                               out_dtype, M, N, relu, bias, out_gate, ldgate, row_scale, nullptr, 0);
}

DGLL_API int dgll_hip_transform_bf16_add(void* stream, const void* A1, int64_t lda1, int K1, const void* Wt1, int64_t ldw1,
                                         const void* A2, int64_t lda2, int K2, const void* Wt2, int64_t ldw2, int wt_rows,
                                         const void* relu_mask, int64_t ldm, void* out, int64_t ldo, int out_dtype,
                                         int64_t M, int N, int relu, const float* bias, const void* out_gate,
                                         int64_t ldgate, const float* row_scale, const void* addend, int64_t ldadd) {
    return transform_bf16_impl(stream, A1, lda1, K1, Wt1, ldw1, A2, lda2, K2, Wt2, ldw2, wt_rows, relu_mask, ldm, out, ldo,
                               out_dtype, M, N, relu, bias, out_gate, ldgate, row_scale, addend, ldadd);
}

DGLL_API int dgll_hip_transform_bf16_bits(void* stream, const void* A1, int64_t lda1, int K1, const void* Wt1, int64_t ldw1,
                                          const void* A2, int64_t lda2, int K2, const void* Wt2, int64_t ldw2, int wt_rows,
                                          void* out, int64_t ldo, int64_t M, int N, int relu, const float* bias,
                                          const void* out_gate, int64_t ldgate, const uint32_t* gate_bits, int64_t ld_gate_bits,
                                          uint32_t* bits_out, int64_t ld_bits_out) {
    return transform_bf16_impl(stream, A1, lda1, K1, Wt1, ldw1, A2, lda2, K2, Wt2, ldw2, wt_rows, nullptr, 0, out, ldo, DGLL_BF16,
                               M, N, relu, bias, out_gate, ldgate, nullptr, nullptr, 0, gate_bits, ld_gate_bits, bits_out, ld_bits_out);
}

DGLL_API int dgll_hip_transform_bf16_dual(void* stream, const void* A, int64_t lda, int K, const void* Wt1, const void* Wt2,
                                          int64_t ldw, int wt_rows, void* out1, int64_t ldo1, void* out2, int64_t ldo2, int64_t M,
                                          int N) {
    DGLL_REQUIRE(M >= 0 && N >= 0 && K >= 0, "negative size");
    if (M == 0 || N == 0) return DGLL_OK;
    DGLL_REQUIRE(A && Wt1 && Wt2 && out1 && out2 && K > 0, "NULL operand");
    DGLL_REQUIRE(N <= 256 && K <= 256, "dgll_hip_transform_bf16_dual: N, K <= 256 (both weight matrices stay resident in LDS)");
    DGLL_REQUIRE(wt_rows >= 256, "Wt1 / Wt2 must be zero-padded to 256 rows");
    DGLL_REQUIRE(aligned16(A) && (lda * 2) % 16 == 0 && lda >= ((K + 7) / 8) * 8, "A: 16-byte aligned rows");
    DGLL_REQUIRE(aligned16(Wt1) && aligned16(Wt2) && (ldw * 2) % 16 == 0 && ldw >= ((K + 63) / 64) * 64,
                 "Wt must be zero-padded to a multiple of 64 columns");
    DGLL_REQUIRE(ldo1 >= N && ldo2 >= N, "output leading dimension");
    MfmaGemmArgs a{};
    a.A[0] = static_cast<const bf16_t*>(A); a.lda[0] = lda; a.K[0] = K;
    a.A[1] = a.A[0]; a.lda[1] = lda; a.K[1] = 0;
    a.Wt[0] = static_cast<const bf16_t*>(Wt1); a.Wt[1] = static_cast<const bf16_t*>(Wt2); a.ldw[0] = a.ldw[1] = ldw;
    a.pairs = 1;
    a.out = out1; a.ldo = ldo1; a.out2 = out2; a.ldo2 = ldo2; a.M = M; a.N = N;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e = hipErrorNotSupported;
#ifndef DGLL_RES_PROBE
    switch ((K + kChunkK - 1) / kChunkK) {             // 256 columns per workgroup (two waves per row group), two workgroups per row block
        case 1: e = launch_mfma_res_p<4, 1, 2, 2, 1, true>(a, s); break;
        case 2: e = launch_mfma_res_p<4, 2, 2, 2, 1, true>(a, s); break;
        case 3: e = launch_mfma_res_p<4, 3, 2, 2, 1, true>(a, s); break;
        default: e = launch_mfma_res_p<4, 4, 2, 2, 1, true>(a, s); break;
    }
#endif
    if (e != hipSuccess) return hip_fail(e, "gemm_bf16_res_kernel (dual) launch");
    return DGLL_OK;
}

static int transform_bf16_impl(void* stream, const void* A1, int64_t lda1, int K1, const void* Wt1, int64_t ldw1,
                               const void* A2, int64_t lda2, int K2, const void* Wt2, int64_t ldw2, int wt_rows,
                               const void* relu_mask, int64_t ldm, void* out, int64_t ldo, int out_dtype, int64_t M,
                               int N, int relu, const float* bias, const void* out_gate, int64_t ldgate,
                               const float* row_scale, const void* addend, int64_t ldadd, const uint32_t* gate_bits,
                               int64_t ld_gate_bits, uint32_t* bits_out, int64_t ld_bits_out) {
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
    a.out = out; a.ldo = ldo; a.M = M; a.N = N; a.relu = relu & 1; a.out_f32 = out_dtype == DGLL_F32; a.bias = bias;
    // relu bit 1: the caller owns the row padding [N, ldo) of a bf16 output and lets the kernel write zeros there
    a.pad_store = (relu & 2) && out_dtype == DGLL_BF16 && (ldo % 8) == 0 && aligned16(out) ? 1 : 0;
    DGLL_REQUIRE(!out_gate || (ldgate >= N && aligned16(out_gate) && (ldgate * 2) % 16 == 0),
                 "out_gate: bf16 [M, ldgate >= N], 16-byte aligned rows");
    a.out_gate = static_cast<const bf16_t*>(out_gate); a.ldgate = ldgate;
    a.row_scale = row_scale;
    DGLL_REQUIRE(!addend || ldadd >= N, "addend: bf16 [M, ldadd >= N]");
    a.addend = static_cast<const bf16_t*>(addend); a.ldadd = ldadd;
    const int n_words = -(-((N + 31) / 32) / 4) * 4;      // gate bits: 32 columns per word, rows padded to whole 16-byte vectors
    DGLL_REQUIRE(!gate_bits || (ld_gate_bits >= n_words && ld_gate_bits % 4 == 0 && aligned16(gate_bits)),
                 "gate_bits: uint32 [M, ld >= 4 * ceil(N / 128)], ld a multiple of 4, 16-byte aligned");
    DGLL_REQUIRE(!bits_out || (ld_bits_out >= n_words && ld_bits_out % 4 == 0 && aligned16(bits_out) && out_dtype == DGLL_BF16),
                 "bits_out: uint32 [M, ld >= 4 * ceil(N / 128)], ld a multiple of 4, 16-byte aligned; bf16 output only");
    a.gate_bits = gate_bits; a.ld_gate_bits = ld_gate_bits; a.bits_out = bits_out; a.ld_bits_out = ld_bits_out;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nt = (N + 31) / 32;
    hipError_t e;
    const int n_chunks = (K1 + kChunkK - 1) / kChunkK + (A2 ? (K2 + kChunkK - 1) / kChunkK : 0);
    a.no_rotate = g_tune_mfma_kperm == 2;
    // Which kernel (tools/transform_probe.py, M = 2.45 M): the persistent resident-weights kernel wherever the weights of the
    // whole reduction fit LDS (K1 + K2 <= 512) -- fused 256+256 -> 256: 0.90 ms against 1.2-1.4 for the 4-wave kernel, single
    // 256 -> 256: 0.59 against 0.82, 256 -> 47: 0.37 against 0.40.  The 4-wave kernel keeps the input-mask form and longer
    // reductions (weights staged per chunk).
    const bool res = g_tune_mfma_kperm != 1 && !a.mask && res_applies(nt, n_chunks) && a.ldw[0] == (a.pairs > 1 ? a.ldw[1] : a.ldw[0]);
    // the bits are an ALTERNATIVE reading of the gate: only the resident-weights kernel's bit epilogue takes them; every other path
    // reads out_gate, which must then be there too
    DGLL_REQUIRE(!gate_bits || out_gate || (res && res_epilogue_kind(a) == 2),
                 "gate_bits alone: this shape / operand set runs a kernel that reads the gate as bf16 -- pass out_gate as well");
    const bool epilogue_bits = res && res_epilogue_kind(a) != 0;      // the plain / bit-gated epilogues write the sign bits themselves
    if (!epilogue_bits) a.bits_out = nullptr;
    if (res) {
        e = launch_mfma_res_dispatch(a, nt, n_chunks, s);
        if (e != hipSuccess) return hip_fail(e, "gemm_bf16_res_kernel launch");
    } else {
        if (nt <= 2) e = launch_mfma<2>(a, s);
        else if (nt <= 4) e = launch_mfma<4>(a, s);
        else e = launch_mfma<8>(a, s);
        if (e != hipSuccess) return hip_fail(e, "gemm_bf16_nt_kernel launch");
    }
    if (bits_out && !epilogue_bits) {                    // the general epilogue and the 4-wave kernel: one pass over what they wrote
        const int64_t n_items = M * ld_bits_out;
        hipLaunchKernelGGL(sign_bits_kernel, dim3((uint32_t)((n_items + kBlock - 1) / kBlock)), dim3(kBlock), 0, s,
                           static_cast<const bf16_t*>(out), ldo, M, N, bits_out, ld_bits_out);
        e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e, "sign_bits_kernel launch");
    }
    return DGLL_OK;
}
