// fused_gcn.hip -- the reference's native entry point, same symbol and signature, on the new kernels.
//
//   extern "C" void launch_gcn_fused_kernel(row_ptr, col_idx, values, X, W, H, num_neighbors,
//                                           N, F_padded, actual_F, H_dim, total_nnz)
//   /root/reference/dgll/FusedKernel/gcn_fused_kernel.cu:190-195 (bound by gcn_extension.cpp:5-10,46-55)
//   semantics: H = relu( A_csr . ( X[:, :actual_F] . W[:actual_F, :] ) ), int32 CSR, fp32       (:39-69)
//
// The reference kernel recomputes X.W for every (edge, output column) -- O(nnz.F.H) flops with scalar uncoalesced
// loads; that design is rejected.  Here the product is factored the cheap way round:
// aggregate-then-transform when actual_F <= H_dim (the PPI layers: 50 -> 64 -> 121), transform-then-aggregate
// otherwise, i.e. one CSR-SpMM launch of the engine plus one dense kernel with the ReLU fused into whichever runs
// last.  Differences kept from the reference contract: int32 CSR is accepted as is (widened on the device), the
// call is synchronous on the default stream (gcn_fused_kernel.cu:229) and `num_neighbors` must equal
// diff(row_ptr) (train_gcn.py:77).  Differences NOT kept: exit(1) on error (:224-227) -- the int-returning twin
// dgll_hip_gcn_fused_forward reports errors instead, and the void symbol prints the error and returns.
#include <cstdio>
#include <vector>

#include "common.hpp"

namespace dgll {
int launch_gemm_f32(hipStream_t s, const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc,
                    int64_t M, int N, int K, const float* bias, int relu, int trans);

// G = grad_output where S > 0 else 0 (the ReLU of the forward), in place over S
__global__ void relu_mask_kernel(float* __restrict__ s, const float* __restrict__ g, int64_t n) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) s[i] = s[i] > 0.0f ? g[i] : 0.0f;
}

__global__ void widen_rowptr_kernel(const int32_t* __restrict__ in, int64_t* __restrict__ out, int64_t n) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i];
}
}  // namespace dgll

using namespace dgll;

DGLL_API int dgll_hip_gcn_fused_forward(void* stream, const int32_t* row_ptr, const int32_t* col_idx, const float* values,
                                        const float* X, const float* W, float* H, int N, int F_padded, int actual_F,
                                        int H_dim, int total_nnz, void* workspace, size_t workspace_bytes) {
    DGLL_REQUIRE(N >= 0 && F_padded >= actual_F && actual_F >= 0 && H_dim >= 0 && total_nnz >= 0, "bad sizes");
    if (N == 0 || H_dim == 0) return DGLL_OK;
    DGLL_REQUIRE(row_ptr && col_idx && values && X && W && H, "NULL argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool aggregate_first = actual_F <= H_dim;
    const int mid = aggregate_first ? actual_F : H_dim;           // width of the intermediate matrix
    const int64_t mid_ld = (mid + 3) & ~3;
    const size_t rp_bytes = (((size_t)N + 1) * sizeof(int64_t) + 15) & ~(size_t)15;
    const size_t need = rp_bytes + (size_t)N * mid_ld * sizeof(float);
    if (!workspace || workspace_bytes < need) {
        set_error("dgll_hip_gcn_fused_forward: workspace too small (need " + std::to_string(need) + " bytes)");
        return DGLL_ERR_WORKSPACE;
    }
    int64_t* rowptr64 = static_cast<int64_t*>(workspace);
    float* mid_buf = reinterpret_cast<float*>(static_cast<char*>(workspace) + rp_bytes);
    hipLaunchKernelGGL(widen_rowptr_kernel, dim3((N + 1 + 255) / 256), dim3(256), 0, s, row_ptr, rowptr64, (int64_t)N + 1);
    DGLL_HIP_TRY(hipGetLastError());
    int rc;
    if (aggregate_first) {
        // AX = A . X[:, :actual_F]   then   H = relu(AX . W[:actual_F, :])
        rc = dgll_hip_spmm_csr(stream, nullptr, rowptr64, col_idx, values, X, F_padded, DGLL_F32, mid_buf, mid_ld, DGLL_F32,
                               N, N, actual_F, DGLL_REDUCE_SUM, DGLL_EPI_NONE, nullptr, nullptr, 0);
        if (rc != DGLL_OK) return rc;
        return launch_gemm_f32(s, mid_buf, mid_ld, W, H_dim, H, H_dim, N, H_dim, actual_F, nullptr, 1, 0);
    }
    // S = X[:, :actual_F] . W   then   H = relu(A . S)
    rc = launch_gemm_f32(s, X, F_padded, W, H_dim, mid_buf, mid_ld, N, H_dim, actual_F, nullptr, 0, 0);
    if (rc != DGLL_OK) return rc;
    return dgll_hip_spmm_csr(stream, nullptr, rowptr64, col_idx, values, mid_buf, mid_ld, DGLL_F32, H, H_dim, DGLL_F32, N, N,
                             H_dim, DGLL_REDUCE_SUM, DGLL_EPI_RELU, nullptr, nullptr, 0);
}

DGLL_API size_t dgll_hip_gcn_fused_workspace_bytes(int N, int actual_F, int H_dim) {
    const int mid = actual_F <= H_dim ? actual_F : H_dim;
    const size_t mid_ld = ((size_t)mid + 3) & ~(size_t)3;
    return ((((size_t)N + 1) * sizeof(int64_t) + 15) & ~(size_t)15) + (size_t)N * mid_ld * sizeof(float);
}

// The reference symbol, verbatim signature (gcn_fused_kernel.cu:190-195).  Scratch is a grow-only per-process buffer
// because this signature has no workspace argument; default stream; synchronous like the original (:229).
DGLL_API void launch_gcn_fused_kernel(const int* row_ptr, const int* col_idx, const float* values, const float* X,
                                      const float* W, float* H, const int* num_neighbors, int N, int F_padded,
                                      int actual_F, int H_dim, int total_nnz) {
    (void)num_neighbors;  // == diff(row_ptr) in the reference's own caller (train_gcn.py:77)
    static void* scratch = nullptr;
    static size_t scratch_bytes = 0;
    const size_t need = dgll_hip_gcn_fused_workspace_bytes(N, actual_F, H_dim);
    if (need > scratch_bytes) {
        if (scratch) (void)hipFree(scratch);
        scratch = nullptr;
        scratch_bytes = 0;
        if (hipMalloc(&scratch, need) != hipSuccess) {
            std::fprintf(stderr, "launch_gcn_fused_kernel: cannot allocate %zu bytes of scratch\n", need);
            return;
        }
        scratch_bytes = need;
    }
    int rc = dgll_hip_gcn_fused_forward(nullptr, row_ptr, col_idx, values, X, W, H, N, F_padded, actual_F, H_dim, total_nnz,
                                        scratch, scratch_bytes);
    if (rc == DGLL_OK && hipDeviceSynchronize() != hipSuccess) rc = DGLL_ERR_HIP;
    if (rc != DGLL_OK) std::fprintf(stderr, "launch_gcn_fused_kernel failed (%d): %s\n", rc, dgll_hip_last_error());
}

// The reference's backward symbol, verbatim signature (gcn_fused_kernel.cu:238-244).  The reference kernel is known-wrong
// (no ReLU mask, grad_X written to the wrong row, a shared-memory race -- SURVEY.md section 2.1); this computes the
// gradient of H = relu(A.(X[:, :F].W[:F])):   G = grad_output * (A.X.W > 0),
//     grad_W[:F] = (A.X)^T . G,   grad_X[:, :F] = A^T . (G . W[:F]^T),   padded rows / columns zero.
// Synchronous on the default stream like the original (:277).  A^T is built on the host from the int32 CSR (this call is
// synchronous by contract; the training path proper keeps its transposed CSR resident, dgll_amd/graph.py).
DGLL_API void launch_gcn_fused_kernel_backward_optimized(const int* row_ptr, const int* col_idx, const float* values,
                                                         const float* X, const float* W, const float* grad_output,
                                                         float* grad_W, float* grad_X, const int* num_neighbors, int N,
                                                         int F_padded, int actual_F, int H_dim, int total_nnz) {
    (void)num_neighbors;
    if (N <= 0 || H_dim <= 0 || actual_F <= 0) return;
    const int F = actual_F;
    const int64_t f_ld = (F + 3) & ~3;
    std::vector<int> h_rp(N + 1), h_ci(total_nnz);
    std::vector<float> h_v(total_nnz);
    bool ok = hipMemcpy(h_rp.data(), row_ptr, sizeof(int) * (N + 1), hipMemcpyDeviceToHost) == hipSuccess &&
              hipMemcpy(h_ci.data(), col_idx, sizeof(int) * total_nnz, hipMemcpyDeviceToHost) == hipSuccess &&
              hipMemcpy(h_v.data(), values, sizeof(float) * total_nnz, hipMemcpyDeviceToHost) == hipSuccess;
    // transpose (stable: entries of a column keep ascending row order -> fixed reduction order)
    std::vector<int64_t> t_rp(N + 1, 0), rp64(N + 1);
    std::vector<int> t_ci(total_nnz);
    std::vector<float> t_v(total_nnz);
    if (ok) {
        for (int k = 0; k < total_nnz; ++k) t_rp[h_ci[k] + 1]++;
        for (int i = 0; i < N; ++i) t_rp[i + 1] += t_rp[i];
        std::vector<int64_t> cur(t_rp.begin(), t_rp.end() - 1);
        for (int r = 0; r < N; ++r)
            for (int k = h_rp[r]; k < h_rp[r + 1]; ++k) {
                const int64_t at = cur[h_ci[k]]++;
                t_ci[at] = r;
                t_v[at] = h_v[k];
            }
        for (int i = 0; i <= N; ++i) rp64[i] = h_rp[i];
    }
    int64_t *d_rp = nullptr, *d_trp = nullptr;
    int* d_tci = nullptr;
    float *d_tv = nullptr, *d_ax = nullptr, *d_s = nullptr, *d_z = nullptr;
    auto alloc = [&](void** p, size_t bytes) { return ok && (ok = hipMalloc(p, bytes ? bytes : 16) == hipSuccess); };
    alloc((void**)&d_rp, sizeof(int64_t) * (N + 1));
    alloc((void**)&d_trp, sizeof(int64_t) * (N + 1));
    alloc((void**)&d_tci, sizeof(int) * total_nnz);
    alloc((void**)&d_tv, sizeof(float) * total_nnz);
    alloc((void**)&d_ax, sizeof(float) * (size_t)N * f_ld);
    alloc((void**)&d_s, sizeof(float) * (size_t)N * H_dim);
    alloc((void**)&d_z, sizeof(float) * (size_t)N * f_ld);
    int rc = ok ? DGLL_OK : DGLL_ERR_HIP;
    if (ok) {
        ok = hipMemcpy(d_rp, rp64.data(), sizeof(int64_t) * (N + 1), hipMemcpyHostToDevice) == hipSuccess &&
             hipMemcpy(d_trp, t_rp.data(), sizeof(int64_t) * (N + 1), hipMemcpyHostToDevice) == hipSuccess &&
             hipMemcpy(d_tci, t_ci.data(), sizeof(int) * total_nnz, hipMemcpyHostToDevice) == hipSuccess &&
             hipMemcpy(d_tv, t_v.data(), sizeof(float) * total_nnz, hipMemcpyHostToDevice) == hipSuccess &&
             hipMemset(grad_W, 0, sizeof(float) * (size_t)F_padded * H_dim) == hipSuccess &&
             hipMemset(grad_X, 0, sizeof(float) * (size_t)N * F_padded) == hipSuccess;
        rc = ok ? DGLL_OK : DGLL_ERR_HIP;
    }
    hipStream_t s = nullptr;
    if (rc == DGLL_OK)   // AX = A . X[:, :F]
        rc = dgll_hip_spmm_csr(nullptr, nullptr, d_rp, col_idx, values, X, F_padded, DGLL_F32, d_ax, f_ld, DGLL_F32, N, N, F,
                               DGLL_REDUCE_SUM, DGLL_EPI_NONE, nullptr, nullptr, 0);
    if (rc == DGLL_OK) rc = launch_gemm_f32(s, d_ax, f_ld, W, H_dim, d_s, H_dim, N, H_dim, F, nullptr, 0, 0);   // S = AX . W
    if (rc == DGLL_OK) {
        const int64_t n = (int64_t)N * H_dim;
        hipLaunchKernelGGL(relu_mask_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, d_s, grad_output, n);
        if (hipGetLastError() != hipSuccess) rc = DGLL_ERR_HIP;
    }
    if (rc == DGLL_OK) rc = launch_gemm_f32(s, d_ax, f_ld, d_s, H_dim, grad_W, H_dim, F, H_dim, N, nullptr, 0, 1);  // AX^T . G
    if (rc == DGLL_OK) rc = launch_gemm_f32(s, d_s, H_dim, W, H_dim, d_z, f_ld, N, F, H_dim, nullptr, 0, 2);        // G . W^T
    if (rc == DGLL_OK)   // grad_X[:, :F] = A^T . Z
        rc = dgll_hip_spmm_csr(nullptr, nullptr, d_trp, d_tci, d_tv, d_z, f_ld, DGLL_F32, grad_X, F_padded, DGLL_F32, N, N, F,
                               DGLL_REDUCE_SUM, DGLL_EPI_NONE, nullptr, nullptr, 0);
    if (rc == DGLL_OK && hipDeviceSynchronize() != hipSuccess) rc = DGLL_ERR_HIP;
    if (rc != DGLL_OK) std::fprintf(stderr, "launch_gcn_fused_kernel_backward_optimized failed (%d): %s\n", rc, dgll_hip_last_error());
    for (void* p : {(void*)d_rp, (void*)d_trp, (void*)d_tci, (void*)d_tv, (void*)d_ax, (void*)d_s, (void*)d_z})
        if (p) (void)hipFree(p);
}
