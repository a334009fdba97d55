// gemm_f32.hip -- the dense products of the layers for fp32 tensors (the 1e-4 parity path), hand-written for gfx950.
//
// Reference call sites: F.mm(x, weight) gcnconv.py:30, F.matmul sageconv.py:41,72, F.mm gatconv.py:31,117 and, through
// autograd, their gradients g.W^T and x^T.g.  Two shapes, both tall-skinny (M ~ 1e4 .. 1e6 rows, N, K <= a few hundred), both on
// v_mfma_f32_32x32x2_f32 (fp32 in, fp32 accumulate: exact fp32 FMA arithmetic, 64 flops per cycle and SIMD = 157 TFLOP/s on 256 CUs;
// at N = K = 256 a product is matrix-core bound, about 2.5 x its HBM time):
//   * dgll_hip_mm2_f32 / dgll_hip_mm_f32   C[M, N] = gate(act(A1[M, K1] . W1t[N, K1]^T (+ A2[M, K2] . W2t[N, K2]^T) + addend + bias))
//     forward products and input gradients; the two-operand form is sageConv's self + neighbour term (sageconv.py:72-75) or the
//     layer's two input-gradient products in ONE accumulation (no stored intermediate), `gate` the ReLU mask of the layer below.
//     A workgroup of 8 wavefronts owns 256 rows x all N <= 256 columns; the reduction runs in chunks of 32: the chunk of the weights
//     ([N, 32]) and of the activations ([256, 32]) is fetched with coalesced 16-byte loads one chunk ahead (registers), parked in
//     LDS (rows padded to 36 floats: conflict-free ds_read_b128) and read from there as MFMA operands -- round 5's kernel fetched
//     every operand straight from L1 with one cache line per lane (9 loads x 64 lines per 32 MFMAs: it ran at 0.3 of the MFMA
//     rate).  The weights are the MFMA "A" operand and the activations "B", so a lane ends up with 4 consecutive output columns
//     of ONE row (float4 stores).  The reduction index inside a group of 8 is permuted -- lane half h takes k = 8 j + 4 h .. + 3
//     -- so that every lane's four MFMA inputs are one 16-byte LDS read, for the activations and the (transposed) weights alike.
//   * dgll_hip_grad_weight_f32   dW[K, N] = X[M, K]^T . G[M, N]          a long reduction into a small output
//     split over row slabs (grid z); inside a slab a wavefront owns up to 2 x 8 output tiles of 32 x 32 and walks the rows two at a
//     time (one MFMA step): lanes read 32 consecutive columns of a row of X (its k-tile) and of G (each n-tile) -- coalesced 128-byte
//     rows, fetched eight rows ahead; rows are summed in row order inside a slab, the slab partials in slab order by a second
//     kernel: deterministic, no atomics.  (Round 5: 64 x 64 tiles through LDS with fmaf on the vector ALUs, 3 x slower.)
#include <algorithm>

#include "common.hpp"

namespace dgll {

typedef __attribute__((ext_vector_type(16))) float f32x16_t;

struct MmF32Args {
    const float* A1;        // [M, lda1]
    int64_t lda1;
    const float* W1;        // [N, ldw1]: the weight TRANSPOSED
    int64_t ldw1;
    int K1;
    const float* A2;        // optional second product (nullptr: none)
    int64_t lda2;
    const float* W2;
    int64_t ldw2;
    int K2;
    float* C;
    int64_t ldc;
    int64_t M;
    int N, relu;
    const float* bias;
    const float* addend;    // optional [M, ldadd]: added before the activation
    int64_t ldadd;
    const float* gate;      // optional [M, ldgate]: outputs are zeroed where gate <= 0 (after the activation)
    int64_t ldgate;
    int vec;                // 1: every operand's rows (activations AND weights) are 16-byte aligned, pitches multiples of 4 (float4 loads)
};

constexpr int kMmThreads = 512;          // 8 wavefronts: two per SIMD
constexpr int kMmRows = 256;             // rows of C per workgroup pass (32 per wavefront)
constexpr int kMmChunk = 32;             // reduction indices per LDS stage
constexpr int kMmLd = kMmChunk + 4;      // LDS row pitch in floats: lanes of a ds_read_b128 land on distinct banks

// Four reduction indices k .. k + 3 of one row WITHOUT a branch: the address is clamped into the row (a `cond ? load : 0` becomes
// an exec-masked branch with its own s_waitcnt -- 1 500 basic blocks in the first version of this kernel, every prefetch
// serialized); what lies past K is cleared later, when the registers are parked in LDS (clear_past: a bit mask -- applied right
// after the load it would make the wavefront wait for the data before the multiply it is meant to overlap).
template <bool VEC>
__device__ __forceinline__ float4 load4(const float* __restrict__ p, int k, int K, int ld) {
    float4 v;
    if constexpr (VEC) {                       // rows 16-byte aligned, ld a multiple of 4 (>= K): the whole vector lies inside the pitch
        const int kk = k < ld - 4 ? k : ld - 4;
        v = *reinterpret_cast<const float4*>(p + kk);
    } else {
        const int last = K - 1;
        v.x = p[k + 0 < last ? k + 0 : last];
        v.y = p[k + 1 < last ? k + 1 : last];
        v.z = p[k + 2 < last ? k + 2 : last];
        v.w = p[k + 3 < last ? k + 3 : last];
    }
    return v;
}
__device__ __forceinline__ float4 clear_past(float4 v, int k, int K) {
    v.x = __uint_as_float(__float_as_uint(v.x) & (k + 0 < K ? 0xffffffffu : 0u));
    v.y = __uint_as_float(__float_as_uint(v.y) & (k + 1 < K ? 0xffffffffu : 0u));
    v.z = __uint_as_float(__float_as_uint(v.z) & (k + 2 < K ? 0xffffffffu : 0u));
    v.w = __uint_as_float(__float_as_uint(v.w) & (k + 3 < K ? 0xffffffffu : 0u));
    return v;
}

template <int NT, bool VEC>
__global__ __launch_bounds__(kMmThreads) void mm_f32_mfma_kernel(const MmF32Args a) {
    extern __shared__ __attribute__((aligned(16))) char mm_smem[];
    float* smem = reinterpret_cast<float*>(mm_smem);
    constexpr int WROWS = NT * 32;
    constexpr int BUF = (WROWS + kMmRows) * kMmLd;      // floats per stage: weight chunk, then the activation chunk
    constexpr int WIT = (WROWS + 63) / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, l32 = lane & 31;
    const int c4 = (tid & 7) * 4, r0 = tid >> 3;        // staging: 8 lanes cover the 32 floats of a row's chunk, 64 rows per sweep
    const int nch1 = (a.K1 + kMmChunk - 1) / kMmChunk;
    const int nch = nch1 + (a.A2 ? (a.K2 + kMmChunk - 1) / kMmChunk : 0);
    const bool c_vec = (a.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(a.C) & 15u) == 0;
    {   // one pass of 256 rows per workgroup (a persistent loop made the compiler hoist the epilogue's 128 lane masks: SGPR spills)
        const int64_t row_base = (int64_t)blockIdx.x * kMmRows;
        f32x16_t acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
        float4 wreg[WIT], areg[4];
        // rows past M / weight rows past N are clamped to the last valid one: what they produce is never stored
        auto fetch = [&](int ch) {
            const bool second = ch >= nch1;
            const float* A = second ? a.A2 : a.A1;
            const float* W = second ? a.W2 : a.W1;
            const int64_t lda = second ? a.lda2 : a.lda1, ldw = second ? a.ldw2 : a.ldw1;
            const int K = second ? a.K2 : a.K1;
            const int k = (second ? ch - nch1 : ch) * kMmChunk + c4;
#pragma unroll
            for (int it = 0; it < WIT; ++it) {
                const int n = r0 + 64 * it;
                wreg[it] = load4<VEC>(W + (int64_t)(n < a.N ? n : a.N - 1) * ldw, k, K, (int)ldw);
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int64_t r = row_base + r0 + 64 * it;
                areg[it] = load4<VEC>(A + (r < a.M ? r : a.M - 1) * lda, k, K, (int)lda);
            }
        };
        auto stash = [&](int ch) {
            float* b = smem + (ch & 1) * BUF;
            const bool second = ch >= nch1;
            const int K = second ? a.K2 : a.K1;
            const int k = (second ? ch - nch1 : ch) * kMmChunk + c4;
#pragma unroll
            for (int it = 0; it < WIT; ++it) {
                const int n = r0 + 64 * it;
                if (n < WROWS) *reinterpret_cast<float4*>(b + n * kMmLd + c4) = clear_past(wreg[it], k, K);
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) *reinterpret_cast<float4*>(b + (WROWS + r0 + 64 * it) * kMmLd + c4) = clear_past(areg[it], k, K);
        };
        fetch(0);
        stash(0);
        __syncthreads();
        for (int ch = 0; ch < nch; ++ch) {
            if (ch + 1 < nch) fetch(ch + 1);            // in flight while this chunk is multiplied
            const float* b = smem + (ch & 1) * BUF;
            const float* wp = b + l32 * kMmLd + 4 * h;
            const float* ap = b + (WROWS + wave * 32 + l32) * kMmLd + 4 * h;
            const bool second = ch >= nch1;
            const int left = (second ? a.K2 - (ch - nch1) * kMmChunk : a.K1 - ch * kMmChunk);      // valid reduction indices of this chunk
            const int steps = left >= kMmChunk ? kMmChunk / 8 : (left + 7) / 8;                     // (the rest of the chunk is zeros)
            for (int s = 0; s < steps; ++s) {
                const float4 av = *reinterpret_cast<const float4*>(ap + 8 * s);
                float4 wv[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) wv[t] = *reinterpret_cast<const float4*>(wp + t * 32 * kMmLd + 8 * s);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[t].x, av.x, acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[t].y, av.y, acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[t].z, av.z, acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[t].w, av.w, acc[t], 0, 0, 0);
            }
            if (ch + 1 < nch) stash(ch + 1);
            __syncthreads();      // the next chunk is complete; nobody still reads the stage that is overwritten after the NEXT multiply
        }
        const int64_t row = row_base + wave * 32 + l32;
        if (row >= a.M) return;
        // D[i][j]: j = lane % 32 = this lane's row, i = (r & 3) + 8 (r >> 2) + 4 h = the output column inside the tile
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = t * 32 + g * 8 + h * 4;
                if (n >= a.N) continue;
                float v[4], ad[4] = {0.0f, 0.0f, 0.0f, 0.0f}, gt[4] = {1.0f, 1.0f, 1.0f, 1.0f};
                const bool whole = n + 4 <= a.N;
                if (a.addend) {
                    const float* q = a.addend + row * a.ldadd + n;
                    if (whole && (a.ldadd & 3) == 0 && (reinterpret_cast<uintptr_t>(a.addend) & 15u) == 0) {
                        const float4 t4 = *reinterpret_cast<const float4*>(q);
                        ad[0] = t4.x; ad[1] = t4.y; ad[2] = t4.z; ad[3] = t4.w;
                    } else {
                        for (int i = 0; i < 4; ++i) if (n + i < a.N) ad[i] = q[i];
                    }
                }
                if (a.gate) {
                    const float* q = a.gate + row * a.ldgate + n;
                    if (whole && (a.ldgate & 3) == 0 && (reinterpret_cast<uintptr_t>(a.gate) & 15u) == 0) {
                        const float4 t4 = *reinterpret_cast<const float4*>(q);
                        gt[0] = t4.x; gt[1] = t4.y; gt[2] = t4.z; gt[3] = t4.w;
                    } else {
                        for (int i = 0; i < 4; ++i) if (n + i < a.N) gt[i] = q[i];
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float x = acc[t][g * 4 + i] + ad[i];
                    if (a.bias && n + i < a.N) x += a.bias[n + i];
                    if (a.relu) x = fmaxf(x, 0.0f);
                    if (!(gt[i] > 0.0f)) x = 0.0f;
                    v[i] = x;
                }
                float* o = a.C + row * a.ldc + n;
                if (n + 4 <= a.N && c_vec)
                    *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
                else
                    for (int i = 0; i < 4; ++i) if (n + i < a.N) o[i] = v[i];
            }
        }
    }
}

// ---- dW = X^T . G: slab partials on the matrix cores -------------------------------------------------------------------
// One MFMA step multiplies TWO rows of the reduction: operand "A" of lane (l32, h) is X[m + h][k-tile + l32], operand "B" is
// G[m + h][n-tile + l32]; D[i][j] accumulates dW[k-tile + i][n-tile + j].  Wavefront w of the 8 owns k-tile w (KTL = 8: 256 columns of
// X per workgroup; KTL = 4: 128, wavefronts 4-7 only help with the staging) and all NT <= 8 n-tiles (256 columns of G).  Chunks of 32
// rows of X and G are fetched one chunk ahead with coalesced 16-byte loads (registers), parked in LDS and read from there as MFMA
// operands (consecutive lanes read consecutive floats: conflict-free); rows are summed in row order.  (The first MFMA version
// fetched the operands straight from global memory, one dword per lane and row: 36 loads in flight per wavefront, 244 registers,
// and slower than the fmaf kernel it replaced.)
constexpr int kGwThreads = 512;
constexpr int kGwRows = 32;      // rows per LDS stage (16 MFMA steps)

template <int KTL, int NT, bool VEC>
__global__ __launch_bounds__(kGwThreads) void gradw_f32_mfma_kernel(const float* __restrict__ X, int64_t ldx, const float* __restrict__ G,
                                                                    int64_t ldg, float* __restrict__ partial, int64_t M, int K, int N,
                                                                    int64_t rows_per_slab) {
    extern __shared__ __attribute__((aligned(16))) char gw_smem[];
    float* smem = reinterpret_cast<float*>(gw_smem);
    constexpr int XW = KTL * 32, GW = NT * 32;
    constexpr int BUF = kGwRows * (XW + GW);                      // floats per stage: the X chunk, then the G chunk
    constexpr int XV = kGwRows * XW / 4, GV = kGwRows * GW / 4;   // 16-byte vectors per chunk
    constexpr int XIT = (XV + kGwThreads - 1) / kGwThreads, GIT = (GV + kGwThreads - 1) / kGwThreads;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, l32 = lane & 31;
    const int kb = blockIdx.x * XW, nb = blockIdx.y * 256;
    const int64_t m_begin = (int64_t)blockIdx.z * rows_per_slab;
    const int64_t m_end = m_begin + rows_per_slab < M ? m_begin + rows_per_slab : M;
    const bool works = wave < KTL && kb + wave * 32 < K;          // this wavefront owns a k-tile
    f32x16_t acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    float4 xreg[XIT], greg[GIT];
    // columns past K / N are clamped into the row (what they accumulate is never stored); rows past the matrix are clamped to its
    // last row and cleared when they are parked (rows past the slab's end must contribute nothing)
    auto fetch = [&](int64_t m) {
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            const int v = tid + it * kGwThreads;
            const int r = v / (XW / 4), c = (v % (XW / 4)) * 4;
            int64_t row = m + r;
            row = row < M ? row : M - 1;
            xreg[it] = load4<VEC>(X + row * ldx, kb + c, K, (int)ldx);
        }
#pragma unroll
        for (int it = 0; it < GIT; ++it) {
            const int v = tid + it * kGwThreads;
            const int r = v / (GW / 4), c = (v % (GW / 4)) * 4;
            int64_t row = m + (r < kGwRows ? r : kGwRows - 1);
            row = row < M ? row : M - 1;
            greg[it] = load4<VEC>(G + row * ldg, nb + c, N, (int)ldg);
        }
    };
    auto stash = [&](int64_t m, int buf) {
        float* b = smem + buf * BUF;
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            const int v = tid + it * kGwThreads;
            const int r = v / (XW / 4), c = (v % (XW / 4)) * 4;
            if (XV % kGwThreads == 0 || v < XV) *reinterpret_cast<float4*>(b + r * XW + c) = clear_past(xreg[it], 0, m + r < m_end ? 4 : 0);
        }
#pragma unroll
        for (int it = 0; it < GIT; ++it) {
            const int v = tid + it * kGwThreads;
            const int r = v / (GW / 4), c = (v % (GW / 4)) * 4;
            if (GV % kGwThreads == 0 || v < GV) *reinterpret_cast<float4*>(b + kGwRows * XW + r * GW + c) = greg[it];
        }
    };
    if (m_begin < m_end) {          // (uniform per workgroup: every wavefront takes the same barriers)
        fetch(m_begin);
        stash(m_begin, 0);
        __syncthreads();
        int buf = 0;
        for (int64_t m = m_begin; m < m_end; m += kGwRows, buf ^= 1) {
            const bool more = m + kGwRows < m_end;
            if (more) fetch(m + kGwRows);
            if (works) {
                const float* xs = smem + buf * BUF + wave * 32 + l32;
                const float* gs = smem + buf * BUF + kGwRows * XW + l32;
#pragma unroll 4
                for (int u = 0; u < kGwRows / 2; ++u) {
                    const float xa = xs[(2 * u + h) * XW];
                    float gb[NT];
#pragma unroll
                    for (int t = 0; t < NT; ++t) gb[t] = gs[(2 * u + h) * GW + t * 32];
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa, gb[t], acc[t], 0, 0, 0);
                }
            }
            if (more) stash(m + kGwRows, buf ^ 1);
            __syncthreads();
        }
    }
    if (!works) return;
    // D[i][j]: register r of lane (l32, h) holds i = (r & 3) + 8 (r >> 2) + 4 h, j = l32: 32 lanes write 32 consecutive n
    float* p = partial + (int64_t)blockIdx.z * K * N;
    const int k0 = kb + wave * 32;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int n = nb + t * 32 + l32;
        if (n >= N) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = k0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (k < K) p[(int64_t)k * N + n] = acc[t][r];
        }
    }
}

__global__ __launch_bounds__(kBlock) void gradw_f32_reduce_kernel(const float* __restrict__ partial, int slabs, int64_t count,
                                                                  float* __restrict__ dW, int64_t lddw, int N) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= count) return;
    float s = 0.0f;
    for (int z = 0; z < slabs; ++z) s += partial[(int64_t)z * count + i];      // slab order: deterministic
    dW[(i / N) * lddw + (i % N)] = s;
}

}  // namespace dgll

using namespace dgll;

static int launch_mm_f32(void* stream, MmF32Args a) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nt = (a.N + 31) / 32;
    const int64_t nblocks = (a.M + kMmRows - 1) / kMmRows;
    DGLL_REQUIRE(nblocks <= 0x7fffffff, "too many rows for one launch");
    dim3 grid((uint32_t)nblocks);
    const size_t lds = (size_t)2 * (nt * 32 + kMmRows) * kMmLd * sizeof(float);
#define DGLL_MM_V(T, V)                                                                                                         \
    {                                                                                                                           \
        static hipError_t raised = hipFuncSetAttribute(reinterpret_cast<const void*>(&mm_f32_mfma_kernel<T, V>),                \
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (T * 32 + kMmRows) * kMmLd * 4); \
        DGLL_HIP_TRY(raised);                                                                                                   \
        hipLaunchKernelGGL((mm_f32_mfma_kernel<T, V>), grid, dim3(kMmThreads), lds, s, a);                                      \
    }
#define DGLL_MM(T) if (a.vec) DGLL_MM_V(T, true) else DGLL_MM_V(T, false)
    switch (nt) {
        case 1: DGLL_MM(1) break;
        case 2: DGLL_MM(2) break;
        case 3: DGLL_MM(3) break;
        case 4: DGLL_MM(4) break;
        case 5: DGLL_MM(5) break;
        case 6: DGLL_MM(6) break;
        case 7: DGLL_MM(7) break;
        default: DGLL_MM(8) break;
    }
#undef DGLL_MM
#undef DGLL_MM_V
    DGLL_HIP_TRY(hipGetLastError());
    return DGLL_OK;
}

static inline int rows_vec(const float* p, int64_t ld) { return aligned16(p) && (ld & 3) == 0; }

DGLL_API int dgll_hip_mm_f32(void* stream, const float* A, int64_t lda, const float* Wt, int64_t ldw, float* C, int64_t ldc,
                             int64_t M, int N, int K, const float* bias, int relu, const float* addend, int64_t ldadd) {
    DGLL_REQUIRE(M >= 0 && N >= 0 && K >= 0, "negative size");
    if (M == 0 || N == 0) return DGLL_OK;
    DGLL_REQUIRE(A && Wt && C && K > 0, "NULL operand");
    DGLL_REQUIRE(N <= 256, "dgll_hip_mm_f32 keeps all N <= 256 output columns of a row block in accumulators (split columns on the host)");
    DGLL_REQUIRE(lda >= K && ldw >= K && ldc >= N && (!addend || ldadd >= N), "leading dimension too small");
    MmF32Args a{};
    a.A1 = A; a.lda1 = lda; a.W1 = Wt; a.ldw1 = ldw; a.K1 = K; a.C = C; a.ldc = ldc; a.M = M; a.N = N; a.relu = relu; a.bias = bias;
    a.addend = addend; a.ldadd = ldadd;
    a.vec = rows_vec(A, lda) && rows_vec(Wt, ldw);
    return launch_mm_f32(stream, a);
}

DGLL_API int dgll_hip_mm2_f32(void* stream, const float* A1, int64_t lda1, const float* W1t, int64_t ldw1, int K1, const float* A2,
                              int64_t lda2, const float* W2t, int64_t ldw2, int K2, float* C, int64_t ldc, int64_t M, int N,
                              const float* bias, int relu, const float* addend, int64_t ldadd, const float* gate, int64_t ldgate) {
    DGLL_REQUIRE(M >= 0 && N >= 0 && K1 >= 0 && K2 >= 0, "negative size");
    if (M == 0 || N == 0) return DGLL_OK;
    DGLL_REQUIRE(A1 && W1t && C && K1 > 0, "NULL operand");
    DGLL_REQUIRE(!A2 || (W2t && K2 > 0), "the second product needs its weight and a reduction length");
    DGLL_REQUIRE(N <= 256, "dgll_hip_mm2_f32 keeps all N <= 256 output columns of a row block in accumulators (split columns on the host)");
    DGLL_REQUIRE(lda1 >= K1 && ldw1 >= K1 && ldc >= N && (!addend || ldadd >= N) && (!gate || ldgate >= N) && (!A2 || (lda2 >= K2 && ldw2 >= K2)),
                 "leading dimension too small");
    MmF32Args a{};
    a.A1 = A1; a.lda1 = lda1; a.W1 = W1t; a.ldw1 = ldw1; a.K1 = K1;
    a.A2 = A2; a.lda2 = lda2; a.W2 = W2t; a.ldw2 = ldw2; a.K2 = A2 ? K2 : 0;
    a.C = C; a.ldc = ldc; a.M = M; a.N = N; a.relu = relu; a.bias = bias; a.addend = addend; a.ldadd = ldadd; a.gate = gate; a.ldgate = ldgate;
    a.vec = rows_vec(A1, lda1) && rows_vec(W1t, ldw1) && (!A2 || (rows_vec(A2, lda2) && rows_vec(W2t, ldw2)));
    return launch_mm_f32(stream, a);
}

DGLL_API int64_t dgll_hip_grad_weight_f32_workspace(int K, int N, int slabs) {
    return (int64_t)std::max(slabs, 1) * K * N * (int64_t)sizeof(float);
}

DGLL_API int dgll_hip_grad_weight_f32(void* stream, const float* X, int64_t ldx, const float* G, int64_t ldg, float* dW,
                                      int64_t lddw, int64_t M, int K, int N, void* workspace, int64_t workspace_bytes, int slabs) {
    DGLL_REQUIRE(M >= 0 && K >= 0 && N >= 0, "negative size");
    if (K == 0 || N == 0) return DGLL_OK;
    DGLL_REQUIRE(X && G && dW && ldx >= K && ldg >= N && lddw >= N, "bad operand");
    hipStream_t s = static_cast<hipStream_t>(stream);
    slabs = (int)std::max<int64_t>(1, std::min<int64_t>(slabs, (M + kGwRows - 1) / kGwRows));
    slabs = std::min(slabs, 65535);
    DGLL_REQUIRE(workspace && workspace_bytes >= dgll_hip_grad_weight_f32_workspace(K, N, slabs), "workspace too small for the slab partials");
    const int64_t per = ((M + slabs - 1) / slabs + kGwRows - 1) / kGwRows * kGwRows;
    const int used = M > 0 ? (int)((M + per - 1) / per) : 1;
    // 8 wavefronts; 256 (K > 128) or 128 columns of X and up to 256 columns of G per workgroup
    const int ktl = K > 128 ? 8 : 4;
    const int nt = (std::min(N, 256) + 31) / 32;
    const bool vec = rows_vec(X, ldx) && rows_vec(G, ldg);
    dim3 grid((uint32_t)((K + ktl * 32 - 1) / (ktl * 32)), (uint32_t)((N + 255) / 256), (uint32_t)used);
    const size_t lds = (size_t)2 * kGwRows * (ktl * 32 + nt * 32) * sizeof(float);
#define DGLL_GW_V(KT, T, V)                                                                                                       \
    {                                                                                                                             \
        static hipError_t raised = hipFuncSetAttribute(reinterpret_cast<const void*>(&gradw_f32_mfma_kernel<KT, T, V>),           \
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kGwRows * (KT * 32 + T * 32) * 4); \
        DGLL_HIP_TRY(raised);                                                                                                     \
        hipLaunchKernelGGL((gradw_f32_mfma_kernel<KT, T, V>), grid, dim3(kGwThreads), lds, s, X, ldx, G, ldg,                     \
                           static_cast<float*>(workspace), M, K, N, per);                                                         \
    }
#define DGLL_GW(T)                                                                                                                \
    if (ktl == 8) { if (vec) DGLL_GW_V(8, T, true) else DGLL_GW_V(8, T, false) }                                                  \
    else { if (vec) DGLL_GW_V(4, T, true) else DGLL_GW_V(4, T, false) }
    switch (nt) {
        case 1: DGLL_GW(1) break;
        case 2: DGLL_GW(2) break;
        case 3: DGLL_GW(3) break;
        case 4: DGLL_GW(4) break;
        case 5: DGLL_GW(5) break;
        case 6: DGLL_GW(6) break;
        case 7: DGLL_GW(7) break;
        default: DGLL_GW(8) break;
    }
#undef DGLL_GW
#undef DGLL_GW_V
    DGLL_HIP_TRY(hipGetLastError());
    const int64_t count = (int64_t)K * N;
    hipLaunchKernelGGL(gradw_f32_reduce_kernel, dim3((uint32_t)((count + kBlock - 1) / kBlock)), dim3(kBlock), 0, s,
                       static_cast<const float*>(workspace), used, count, dW, lddw, N);
    DGLL_HIP_TRY(hipGetLastError());
    return DGLL_OK;
}
