// dense.hip -- dense transforms next to the aggregation: C[M,N] = act(A[M,K] . B[K,N] + bias).
//
// Reference call sites: F.mm(x, weight) gcnconv.py:30, F.matmul sageconv.py:41,72, F.mm gatconv.py:31,117, and the
// X.W inside FusedKernel/gcn_fused_kernel.cu:46-54.  This file holds the exact-fp32 kernel used by the C-ABI-only
// entry points (the fused GCN launcher); the Python layers use library GEMMs for forward/dX.
#include "common.hpp"

namespace dgll {

constexpr int TM = 64, TN = 64, TK = 16;   // 64x64 output tile per 256-thread block, 4x4 outputs per thread

// fp32 LDS-tiled GEMM, fmaf accumulation in k order (bit-stable), optional bias and ReLU epilogue.
__global__ __launch_bounds__(kBlock) void gemm_f32_kernel(const float* __restrict__ A, int64_t lda,
                                                          const float* __restrict__ B, int64_t ldb,
                                                          float* __restrict__ C, int64_t ldc, int64_t M, int N, int K,
                                                          const float* __restrict__ bias, int relu) {
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
            sA[c][r] = (gm < M && k0 + c < K) ? A[gm * lda + k0 + c] : 0.0f;
        }
        for (int i = threadIdx.x; i < TK * TN; i += kBlock) {   // B tile: TK rows x TN cols
            const int r = i / TN, c = i % TN;
            sB[r][c] = (k0 + r < K && n0 + c < N) ? B[(int64_t)(k0 + r) * ldb + n0 + c] : 0.0f;
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
                    int64_t M, int N, int K, const float* bias, int relu) {
    if (M <= 0 || N <= 0) return DGLL_OK;
    dim3 grid((uint32_t)((M + TM - 1) / TM), (uint32_t)((N + TN - 1) / TN));
    hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(kBlock), 0, s, A, lda, B, ldb, C, ldc, M, N, K, bias, relu);
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
    return launch_gemm_f32(static_cast<hipStream_t>(stream), A, lda, B, ldb, C, ldc, M, N, K, bias, relu);
}
