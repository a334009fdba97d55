/* Plain-C client of the C ABI (no Python, no torch): the call a cgo / JNI / pybind binding would make.
 * Builds a 4 x 4 CSR on the device, runs dgll_hip_spmm_csr (fp32, weighted, bias + ReLU epilogue) and the reference-named
 * fused GCN launcher, and checks both against host loops.  Exit code 0 = pass. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "dgll_hip.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

int main(void) {
    const int64_t rowptr[5] = {0, 2, 2, 5, 6};                 /* row 1 is empty */
    const int32_t col[6] = {1, 3, 0, 1, 2, 3};
    const float val[6] = {0.5f, -1.0f, 2.0f, 1.0f, 0.25f, 3.0f};
    enum { N = 4, F = 8 };
    float X[N * F], bias[F], want[N * F], got[N * F];
    for (int i = 0; i < N * F; ++i) X[i] = (float)((i * 7) % 11) - 5.0f;
    for (int f = 0; f < F; ++f) bias[f] = 0.1f * (float)f - 0.3f;
    for (int r = 0; r < N; ++r)
        for (int f = 0; f < F; ++f) {
            float s = 0.0f;
            for (int64_t k = rowptr[r]; k < rowptr[r + 1]; ++k) s += val[k] * X[col[k] * F + f];
            s += bias[f];
            want[r * F + f] = s > 0.0f ? s : 0.0f;
        }
    int64_t* d_rp; int32_t* d_col; float *d_val, *d_x, *d_y, *d_bias;
    CHECK(hipMalloc((void**)&d_rp, sizeof rowptr)); CHECK(hipMalloc((void**)&d_col, sizeof col));
    CHECK(hipMalloc((void**)&d_val, sizeof val)); CHECK(hipMalloc((void**)&d_x, sizeof X));
    CHECK(hipMalloc((void**)&d_y, sizeof got)); CHECK(hipMalloc((void**)&d_bias, sizeof bias));
    CHECK(hipMemcpy(d_rp, rowptr, sizeof rowptr, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_col, col, sizeof col, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_val, val, sizeof val, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_x, X, sizeof X, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_bias, bias, sizeof bias, hipMemcpyHostToDevice));

    if (dgll_hip_abi_version() != DGLL_HIP_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 3; }
    int rc = dgll_hip_spmm_csr(NULL, NULL, d_rp, d_col, d_val, d_x, F, DGLL_F32, d_y, F, DGLL_F32, N, N, F, DGLL_REDUCE_SUM,
                               DGLL_EPI_BIAS | DGLL_EPI_RELU, d_bias, NULL, 0);
    if (rc != DGLL_OK) { fprintf(stderr, "spmm failed: %s\n", dgll_hip_last_error()); return 4; }
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(got, d_y, sizeof got, hipMemcpyDeviceToHost));
    for (int i = 0; i < N * F; ++i)
        if (fabsf(got[i] - want[i]) > 1e-5f) { fprintf(stderr, "spmm mismatch at %d: %g vs %g\n", i, got[i], want[i]); return 5; }

    /* error path: bad dtype is reported, not fatal */
    rc = dgll_hip_spmm_csr(NULL, NULL, d_rp, d_col, d_val, d_x, F, 7, d_y, F, DGLL_F32, N, N, F, 0, 0, NULL, NULL, 0);
    if (rc != DGLL_ERR_INVALID) { fprintf(stderr, "expected DGLL_ERR_INVALID, got %d\n", rc); return 6; }

    /* the reference's own entry point (gcn_fused_kernel.cu:190-195): int32 CSR, H = relu(A.(X.W)) */
    const int rp32[5] = {0, 2, 2, 5, 6};
    enum { H = 3 };
    float W[F * H], wantH[N * H], gotH[N * H];
    for (int i = 0; i < F * H; ++i) W[i] = 0.05f * (float)((i * 5) % 13) - 0.2f;
    for (int r = 0; r < N; ++r)
        for (int h = 0; h < H; ++h) {
            float s = 0.0f;
            for (int k = rp32[r]; k < rp32[r + 1]; ++k) {
                float z = 0.0f;
                for (int f = 0; f < F; ++f) z += X[col[k] * F + f] * W[f * H + h];
                s += val[k] * z;
            }
            wantH[r * H + h] = s > 0.0f ? s : 0.0f;
        }
    int *d_rp32, *d_nn; float *d_w, *d_h;
    const int nn[4] = {2, 0, 3, 1};
    CHECK(hipMalloc((void**)&d_rp32, sizeof rp32)); CHECK(hipMalloc((void**)&d_nn, sizeof nn));
    CHECK(hipMalloc((void**)&d_w, sizeof W)); CHECK(hipMalloc((void**)&d_h, sizeof gotH));
    CHECK(hipMemcpy(d_rp32, rp32, sizeof rp32, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_nn, nn, sizeof nn, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_w, W, sizeof W, hipMemcpyHostToDevice));
    launch_gcn_fused_kernel(d_rp32, d_col, d_val, d_x, d_w, d_h, d_nn, N, F, F, H, 6);
    CHECK(hipMemcpy(gotH, d_h, sizeof gotH, hipMemcpyDeviceToHost));
    for (int i = 0; i < N * H; ++i)
        if (fabsf(gotH[i] - wantH[i]) > 1e-4f) { fprintf(stderr, "fused gcn mismatch at %d: %g vs %g\n", i, gotH[i], wantH[i]); return 7; }
    printf("c abi smoke ok\n");
    return 0;
}
