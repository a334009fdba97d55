// gemm_f32.hip -- the dense products of the layers for fp32 tensors (the 1e-4 parity path), hand-written for gfx950.
//
// Reference call sites: F.mm(x, weight) gcnconv.py:30, F.matmul sageconv.py:41,72, F.mm gatconv.py:31,117 and, through
// autograd, their gradients g.W^T and x^T.g.  Two shapes, both tall-skinny (M ~ 1e4 .. 1e6 rows, N, K <= a few hundred):
//   * dgll_hip_mm_f32       C[M, N] = act(A[M, K] . Wt[N, K]^T + bias)      forward products and input gradients
//     v_mfma_f32_32x32x2_f32 (fp32 in, fp32 accumulate: exact fp32 FMA arithmetic, 1/16 of the bf16 rate -- at N = K = 256 the
//     product is matrix-core bound, about 2.5 x the HBM time).  One wavefront owns 32 rows x all N <= 256 columns; the weights
//     are the MFMA "A" operand and the activations "B", so a lane ends up with 4 consecutive output columns of ONE row
//     (float4 stores).  The reduction index inside a group of 8 is permuted -- lane half h takes k = 8 j + 4 h .. + 3 -- so
//     that every lane's four MFMA inputs are one 16-byte load, for the activations and the (transposed) weights alike.
//   * dgll_hip_grad_weight_f32   dW[K, N] = X[M, K]^T . G[M, N]          a long reduction into a small output
//     split over row slabs (grid z), 64 x 64 output tiles per workgroup through LDS, fmaf in row order inside a slab, the slab
//     partials summed in slab order by a second kernel: deterministic, no atomics.
#include <algorithm>

#include "common.hpp"

namespace dgll {

typedef __attribute__((ext_vector_type(16))) float f32x16_t;

struct MmF32Args {
    const float* A;
    int64_t lda;
    const float* Wt;
    int64_t ldw;
    float* C;
    int64_t ldc;
    int64_t M;
    int N, K, relu;
    const float* bias;
    const float* addend;    // optional [M, ldadd]: added before the activation
    int64_t ldadd;
    int vec;                // 1: A / Wt rows are 16-byte aligned (float4 loads)
};

__device__ __forceinline__ void load4(const float* __restrict__ p, int k, int K, bool vec, float (&v)[4]) {
    if (vec && k + 4 <= K) {
        const float4 t = *reinterpret_cast<const float4*>(p + k);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = k + i < K ? p[k + i] : 0.0f;
    }
}

template <int NT>
__global__ __launch_bounds__(kBlock) void mm_f32_mfma_kernel(const MmF32Args a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, l32 = lane & 31;
    const int64_t row = ((int64_t)blockIdx.x * kWavesPerBlock + wave) * 32 + l32;
    const int64_t row_ld = row < a.M ? row : a.M - 1;
    const float* ap = a.A + row_ld * a.lda;
    f32x16_t acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    for (int k0 = 0; k0 < a.K; k0 += 8) {
        const int k = k0 + 4 * h;
        float av[4], wv[NT][4];
        load4(ap, k, a.K, a.vec != 0, av);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int n = t * 32 + l32;
            if (n < a.N) load4(a.Wt + (int64_t)n * a.ldw, k, a.K, a.vec != 0, wv[t]);
            else { wv[t][0] = wv[t][1] = wv[t][2] = wv[t][3] = 0.0f; }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[t][i], av[i], acc[t], 0, 0, 0);
    }
    if (row >= a.M) return;
    // D[i][j]: j = lane % 32 = this lane's row, i = (r & 3) + 8 (r >> 2) + 4 h = the output column inside the tile
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = t * 32 + g * 8 + h * 4;
            if (n >= a.N) continue;
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float x = acc[t][g * 4 + i];
                if (a.addend && n + i < a.N) x += a.addend[row * a.ldadd + n + i];
                if (a.bias && n + i < a.N) x += a.bias[n + i];
                if (a.relu) x = fmaxf(x, 0.0f);
                v[i] = x;
            }
            float* o = a.C + row * a.ldc + n;
            if (n + 4 <= a.N && (a.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(a.C) & 15u) == 0)
                *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
            else
                for (int i = 0; i < 4; ++i) if (n + i < a.N) o[i] = v[i];
        }
    }
}

// ---- dW = X^T . G: slab partials ------------------------------------------------------------------------------------
constexpr int GT = 64, GK = 16;

__global__ __launch_bounds__(kBlock) void gradw_f32_slab_kernel(const float* __restrict__ X, int64_t ldx, const float* __restrict__ G,
                                                                int64_t ldg, float* __restrict__ partial, int64_t M, int K, int N,
                                                                int64_t rows_per_slab) {
    __shared__ float sX[GK][GT + 4];   // [reduction row][k column of X]
    __shared__ float sG[GK][GT + 4];   // [reduction row][n column of G]
    const int tx = threadIdx.x % 16, ty = threadIdx.x / 16;
    const int k0 = blockIdx.x * GT, n0 = blockIdx.y * GT;
    const int64_t m_begin = (int64_t)blockIdx.z * rows_per_slab;
    const int64_t m_end = m_begin + rows_per_slab < M ? m_begin + rows_per_slab : M;
    float acc[4][4] = {};
    for (int64_t m0 = m_begin; m0 < m_end; m0 += GK) {
        for (int i = threadIdx.x; i < GK * GT; i += kBlock) {
            const int r = i / GT, c = i % GT;
            const int64_t m = m0 + r;
            sX[r][c] = (m < m_end && k0 + c < K) ? X[m * ldx + k0 + c] : 0.0f;
            sG[r][c] = (m < m_end && n0 + c < N) ? G[m * ldg + n0 + c] : 0.0f;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < GK; ++r) {
            float xv[4], gv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { xv[i] = sX[r][ty * 4 + i]; gv[i] = sG[r][tx * 4 + i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(xv[i], gv[j], acc[i][j]);
        }
        __syncthreads();
    }
    float* p = partial + (int64_t)blockIdx.z * K * N;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = k0 + ty * 4 + i;
        if (k >= K) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + tx * 4 + j;
            if (n < N) p[(int64_t)k * N + n] = acc[i][j];
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

DGLL_API int dgll_hip_mm_f32(void* stream, const float* A, int64_t lda, const float* Wt, int64_t ldw, float* C, int64_t ldc,
                             int64_t M, int N, int K, const float* bias, int relu, const float* addend, int64_t ldadd) {
    DGLL_REQUIRE(M >= 0 && N >= 0 && K >= 0, "negative size");
    if (M == 0 || N == 0) return DGLL_OK;
    DGLL_REQUIRE(A && Wt && C && K > 0, "NULL operand");
    DGLL_REQUIRE(N <= 256, "dgll_hip_mm_f32 keeps all N <= 256 output columns of a row block in accumulators (split columns on the host)");
    DGLL_REQUIRE(lda >= K && ldw >= K && ldc >= N && (!addend || ldadd >= N), "leading dimension too small");
    MmF32Args a{};
    a.A = A; a.lda = lda; a.Wt = Wt; a.ldw = ldw; a.C = C; a.ldc = ldc; a.M = M; a.N = N; a.K = K; a.relu = relu; a.bias = bias;
    a.addend = addend; a.ldadd = ldadd;
    a.vec = aligned16(A) && aligned16(Wt) && (lda & 3) == 0 && (ldw & 3) == 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    dim3 grid((uint32_t)((M + 127) / 128));
    const int nt = (N + 31) / 32;
#define DGLL_MM(T) hipLaunchKernelGGL((mm_f32_mfma_kernel<T>), grid, dim3(kBlock), 0, s, a)
    switch (nt) {
        case 1: DGLL_MM(1); break;
        case 2: DGLL_MM(2); break;
        case 3: DGLL_MM(3); break;
        case 4: DGLL_MM(4); break;
        case 5: DGLL_MM(5); break;
        case 6: DGLL_MM(6); break;
        case 7: DGLL_MM(7); break;
        default: DGLL_MM(8); break;
    }
#undef DGLL_MM
    DGLL_HIP_TRY(hipGetLastError());
    return DGLL_OK;
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
    slabs = (int)std::max<int64_t>(1, std::min<int64_t>(slabs, (M + GK - 1) / GK));
    slabs = std::min(slabs, 65535);
    DGLL_REQUIRE(workspace && workspace_bytes >= dgll_hip_grad_weight_f32_workspace(K, N, slabs), "workspace too small for the slab partials");
    const int64_t per = ((M + slabs - 1) / slabs + GK - 1) / GK * GK;
    const int used = M > 0 ? (int)((M + per - 1) / per) : 1;
    dim3 grid((uint32_t)((K + GT - 1) / GT), (uint32_t)((N + GT - 1) / GT), (uint32_t)used);
    hipLaunchKernelGGL(gradw_f32_slab_kernel, grid, dim3(kBlock), 0, s, X, ldx, G, ldg, static_cast<float*>(workspace), M, K, N, per);
    DGLL_HIP_TRY(hipGetLastError());
    const int64_t count = (int64_t)K * N;
    hipLaunchKernelGGL(gradw_f32_reduce_kernel, dim3((uint32_t)((count + kBlock - 1) / kBlock)), dim3(kBlock), 0, s,
                       static_cast<const float*>(workspace), used, count, dW, lddw, N);
    DGLL_HIP_TRY(hipGetLastError());
    return DGLL_OK;
}
