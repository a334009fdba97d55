// optim.hip -- the optimizer step of the training loops around the aggregation path as ONE launch.
//
// The reference's loops call torch.optim.Adam(model.parameters()) (GPU Accelerator/MQGCN.py:141-144, Evaluation/PPI/train_gcn.py:26)
// after an all-reduce that walks the parameters one by one (MQGCN.py:55-79).  torch's multi-tensor Adam is seven launches per
// step; at 8 ranks of the products-sized graph a rank's whole step is ~4 ms, and those launches plus the copies into and out of
// RaCoM's flat bucket plus one pack launch per weight and direction were a tenth of it.  Here every parameter lives in one flat
// fp32 buffer, every gradient in a second one (dgll_amd/optim.py: the weight-gradient kernels write their slots directly, the
// gradient all-reduce runs on that buffer in place), and this kernel
//   * applies Adam (torch.optim.Adam's arithmetic, in its order of operations) to the whole buffer, and
//   * for the parameters that are weight matrices of the MFMA transforms, writes the updated value straight into the two packed
//     bf16 forms those kernels take ([N rows, K padded] of W^T for the forward product, [K rows, N padded] of W for the input
//     gradient) -- what dgll_hip_pack_weight_bf16 otherwise does once per product and step.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "common.hpp"

namespace dgll {

constexpr int kAdamMaxSegments = 24;

struct AdamSegment {
    int64_t begin, end;      // element range of the flat buffer
    int cols;                // columns of the [rows, cols] matrix (0: not a packed weight)
    int64_t ld, ldt;         // pitches (elements) of the two packed forms
    bf16_t* packed;          // [rows_padded, ld]   : W itself           (wt = W,   the input-gradient product)
    bf16_t* packed_t;        // [cols_padded, ldt]  : W transposed       (wt = W^T, the forward product)
};

struct AdamArgs {
    float* param;
    const float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    int64_t n;
    float grad_scale;        // applied to the gradient first (1 / world_size of the RaCoM average, MQGCN.py:64)
    float weight_decay, beta1, beta2, eps;
    float step_size;         // lr / (1 - beta1^t)
    float bc2_sqrt;          // sqrt(1 - beta2^t)
    int n_segments;
    AdamSegment seg[kAdamMaxSegments];
};

__global__ __launch_bounds__(256) void adam_flat_kernel(const AdamArgs a) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    float p = a.param[i];
    float g = a.grad[i] * a.grad_scale;
    if (a.weight_decay != 0.0f) g = g + a.weight_decay * p;                     // grad.add(param, alpha = weight_decay)
    float m = a.exp_avg[i];
    m = m + (g - m) * (1.0f - a.beta1);                                        // exp_avg.lerp_(grad, 1 - beta1)
    float v = a.exp_avg_sq[i] * a.beta2;                                       // exp_avg_sq.mul_(beta2)
    v = v + (1.0f - a.beta2) * (g * g);                                        //           .addcmul_(grad, grad, value = 1 - beta2)
    const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;                         // (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps)
    p = p - a.step_size * (m / denom);                                         // param.addcdiv_(exp_avg, denom, value = -step_size)
    a.param[i] = p;
    a.exp_avg[i] = m;
    a.exp_avg_sq[i] = v;
    for (int s = 0; s < a.n_segments; ++s) {                                   // a handful of segments: uniform over most of a wave
        const AdamSegment& sg = a.seg[s];
        if (i >= sg.begin && i < sg.end && sg.cols > 0) {
            const int64_t e = i - sg.begin;
            const int64_t r = e / sg.cols, c = e - r * sg.cols;
            const bf16_t b = f32_to_bf16(p);
            if (sg.packed) sg.packed[r * sg.ld + c] = b;
            if (sg.packed_t) sg.packed_t[c * sg.ldt + r] = b;
            break;
        }
    }
}

}  // namespace dgll

using namespace dgll;

DGLL_API int dgll_hip_adam_flat(void* stream, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                                float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step, float grad_scale,
                                int n_segments, const int64_t* seg_begin, const int64_t* seg_end, const int* seg_cols,
                                void* const* seg_packed, const int64_t* seg_ld, void* const* seg_packed_t, const int64_t* seg_ldt) {
    DGLL_REQUIRE(n >= 0 && step >= 1, "bad size / step");
    DGLL_REQUIRE(n == 0 || (param && grad && exp_avg && exp_avg_sq), "NULL operand");
    DGLL_REQUIRE(n_segments >= 0 && n_segments <= kAdamMaxSegments, "too many packed segments (at most 24)");
    DGLL_REQUIRE(n_segments == 0 || (seg_begin && seg_end && seg_cols && seg_packed && seg_ld && seg_packed_t && seg_ldt), "NULL segment table");
    if (n == 0) return DGLL_OK;
    AdamArgs a{};
    a.param = param; a.grad = grad; a.exp_avg = exp_avg; a.exp_avg_sq = exp_avg_sq; a.n = n;
    a.grad_scale = grad_scale; a.weight_decay = weight_decay; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps;
    // the bias corrections in double, as the Python floats of torch.optim.adam._single_tensor_adam are
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    a.step_size = (float)((double)lr / bc1);
    a.bc2_sqrt = (float)sqrt(bc2);
    a.n_segments = n_segments;
    for (int s = 0; s < n_segments; ++s) {
        DGLL_REQUIRE(seg_begin[s] >= 0 && seg_end[s] <= n && seg_begin[s] <= seg_end[s] && seg_cols[s] >= 0, "bad segment");
        a.seg[s] = AdamSegment{seg_begin[s], seg_end[s], seg_cols[s], seg_ld[s], seg_ldt[s], static_cast<bf16_t*>(seg_packed[s]),
                               static_cast<bf16_t*>(seg_packed_t[s])};
    }
    hipLaunchKernelGGL(adam_flat_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "adam_flat_kernel launch");
    return DGLL_OK;
}
