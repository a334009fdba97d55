// loss.hip -- softmax cross-entropy over the last layer's rows, loss and gradient each in one pass.
//
// The reference's training loops end in nn.CrossEntropyLoss on the model output (Evaluation/PPI/train_gcn.py:27,45; the
// GraphSAGE loop in examples).  Done with library ops that is six passes over [N, C] (cast, log_softmax, gather, and their
// backward twins, ~1.6 ms per step on the products-shaped graph); here each direction reads the logits once:
//   row_loss[i] = logsumexp(z_i) - z_i[label_i]                       (0 where label_i < 0: ignored, as ignore_index)
//   grad[i, c]  = scale * (softmax(z_i)[c] - [c == label_i])          (scale read from device memory: no host sync)
// A group of G lanes owns one row (64/G rows per wavefront) and each lane up to PER = 8 classes (G = pow2 >= C/8: 8 lanes
// for 47 classes, so a wavefront covers 8 rows = one contiguous 752-byte stretch of bf16 logits); group reductions with
// wave shuffles, fp32 math.  HBM-bound: C * (s_z [+ s_g]) bytes per row.  (bf16 logits on 16-byte aligned rows with class-index labels
// take softmax_xent_vec_kernel: eight CONSECUTIVE classes per lane, one 16-byte access each.)
#include "common.hpp"

namespace dgll {

struct XentArgs {
    const void* z;
    int64_t ldz;
    const int64_t* labels;  // class-index targets, or NULL when `soft` is given
    const float* soft;      // probability / multi-hot targets fp32 [n_rows, lds] (nn.CrossEntropyLoss with float targets)
    int64_t lds;
    float* row_loss;        // optional
    void* grad;             // optional
    int64_t ldg;
    const float* scale;     // device scalar (may be NULL: 1)
    int64_t n_rows;
    int n_classes;
    int pad_store;          // the gradient rows' padding [n_classes, ldg) belongs to the output: whole 16-byte vectors are stored (zeros)
    int mask_nonpositive;   // gradient pass: zero the gradient where the logit is <= 0 (the logits are ReLU outputs and the loss
                            // takes over that ReLU's backward: d loss / d pre-activation)
};

template <typename T> __device__ __forceinline__ float ld(const T* p);
template <> __device__ __forceinline__ float ld<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld<bf16_t>(const bf16_t* p) { return bf16_to_f32(*p); }

template <typename T, int G, int PER, bool SOFT>
__global__ __launch_bounds__(kBlock) void softmax_xent_kernel(const XentArgs a) {
    constexpr int ROWS = kWave / G;
    const int lane = lane_id();
    const int sub = lane % G;
    const int64_t row = ((int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6)) * ROWS + lane / G;
    const bool live = row < a.n_rows;
    const int64_t r = live ? row : a.n_rows - 1;          // every lane takes part in the shuffles
    const T* z = static_cast<const T*>(a.z) + r * a.ldz;
    float v[PER], t[SOFT ? PER : 1];
    float m = -INFINITY, tsum = 0.0f, tz = 0.0f;
    uint32_t keep = 0xffffffffu;          // bit j: class sub + j G keeps its gradient
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int c = sub + j * G;
        v[j] = c < a.n_classes ? ld<T>(z + c) : -INFINITY;
        if (a.mask_nonpositive && !(v[j] > 0.0f)) keep &= ~(1u << j);
        m = fmaxf(m, v[j]);
        if constexpr (SOFT) {
            t[j] = c < a.n_classes ? a.soft[r * a.lds + c] : 0.0f;
            tsum += t[j];
            if (c < a.n_classes) tz += t[j] * v[j];
        }
    }
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) {
        m = fmaxf(m, __shfl_xor(m, off));
        if constexpr (SOFT) { tsum += __shfl_xor(tsum, off); tz += __shfl_xor(tz, off); }
    }
    float sum = 0.0f;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        v[j] = __expf(v[j] - m);                           // exp(-inf) = 0 for the padding classes
        sum += v[j];
    }
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    if (!live) return;
    const float inv = 1.0f / sum;
    if constexpr (SOFT) {      // loss_i = -sum_c t_c log p_c = (logsumexp) * sum_c t_c - sum_c t_c z_c ;  d/dz_c = p_c sum_c t_c - t_c
        if (a.row_loss && sub == 0) a.row_loss[r] = (m + __logf(sum)) * tsum - tz;
        if (a.grad) {
            const float scale = a.scale ? *a.scale : 1.0f;
            T* g = static_cast<T*>(a.grad) + r * a.ldg;
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const int c = sub + j * G;
                if (c < a.n_classes) store_one<T>(g + c, ((keep >> j) & 1u) ? scale * (v[j] * inv * tsum - t[j]) : 0.0f);
            }
        }
    } else {
        const int64_t label = a.labels[r];
        const bool counted = label >= 0 && label < a.n_classes;
        if (a.row_loss && sub == 0) a.row_loss[r] = counted ? (m + __logf(sum)) - ld<T>(z + label) : 0.0f;
        if (a.grad) {
            const float scale = counted ? (a.scale ? *a.scale : 1.0f) : 0.0f;
            T* g = static_cast<T*>(a.grad) + r * a.ldg;
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const int c = sub + j * G;
                if (c < a.n_classes) store_one<T>(g + c, ((keep >> j) & 1u) ? scale * (v[j] * inv - (c == label ? 1.0f : 0.0f)) : 0.0f);
            }
        }
    }
}

// bf16 logits, class-index labels, 16-byte aligned rows: lane `sub` of a row's group holds EIGHT CONSECUTIVE classes -- one 16-byte
// load (and one 16-byte gradient store) per lane, a wave instruction covers 64 / G whole rows -- instead of eight 2-byte accesses
// strided by G (each of which touched every row's line again: 2.1 / 3.3 TB/s for the two directions at 47 classes).
template <int G>
__global__ __launch_bounds__(kBlock) void softmax_xent_vec_kernel(const XentArgs a) {
    constexpr int ROWS = kWave / G;
    const int lane = lane_id();
    const int sub = lane % G;
    const int64_t row = ((int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6)) * ROWS + lane / G;
    const bool live = row < a.n_rows;
    const int64_t r = live ? row : a.n_rows - 1;          // every lane takes part in the shuffles
    const bf16_t* z = static_cast<const bf16_t*>(a.z) + r * a.ldz;
    const int c0 = sub * 8;
    float v[8];
    {
        uint4 raw = make_uint4(0u, 0u, 0u, 0u);
        if (c0 < a.n_classes) raw = *reinterpret_cast<const uint4*>(z + c0);      // c0 + 8 <= ldz: the host checked ldz >= 8 ceil(C / 8)
        const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
        for (int d = 0; d < 4; ++d) { v[2 * d] = bf16_lo(w[d]); v[2 * d + 1] = bf16_hi(w[d]); }
    }
    float m = -INFINITY;
    uint32_t keep = 0xffu;
    float zl = 0.0f;                                       // the label's logit (one lane of the group holds it)
    const int64_t label = a.labels[r];
    const bool counted = label >= 0 && label < a.n_classes;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (c0 + j >= a.n_classes) v[j] = -INFINITY;
        if (a.mask_nonpositive && !(v[j] > 0.0f)) keep &= ~(1u << j);
        if (c0 + j == label) zl = v[j];
        m = fmaxf(m, v[j]);
    }
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) { m = fmaxf(m, __shfl_xor(m, off)); zl += __shfl_xor(zl, off); }
    float sum = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { v[j] = __expf(v[j] - m); sum += v[j]; }
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    if (!live) return;
    const float inv = 1.0f / sum;
    if (a.row_loss && sub == 0) a.row_loss[r] = counted ? (m + __logf(sum)) - zl : 0.0f;
    if (a.grad && c0 < a.n_classes) {
        const float scale = counted ? (a.scale ? *a.scale : 1.0f) : 0.0f;
        bf16_t* g = static_cast<bf16_t*>(a.grad) + r * a.ldg + c0;
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = ((keep >> j) & 1u) ? scale * (v[j] * inv - (c0 + j == label ? 1.0f : 0.0f)) : 0.0f;
        if (c0 + 8 <= a.n_classes || (a.pad_store && c0 + 8 <= a.ldg)) {   // (classes past C: exp(-inf) = 0 -> a zero gradient)
            *reinterpret_cast<uint4*>(g) = make_uint4(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7]));
        } else {                                           // the row's ragged last vector: only the classes that exist are written
#pragma unroll
            for (int j = 0; j < 8; ++j) if (c0 + j < a.n_classes) g[j] = f32_to_bf16(o[j]);
        }
    }
}

template <int G>
static hipError_t launch_xent_vec(const XentArgs& a, hipStream_t s) {
    const int64_t rows_per_block = (int64_t)kWavesPerBlock * (kWave / G);
    dim3 grid((uint32_t)((a.n_rows + rows_per_block - 1) / rows_per_block));
    hipLaunchKernelGGL((softmax_xent_vec_kernel<G>), grid, dim3(kBlock), 0, s, a);
    return hipGetLastError();
}

template <typename T, int G, int PER>
static hipError_t launch_xent(const XentArgs& a, hipStream_t s) {
    const int64_t rows_per_block = (int64_t)kWavesPerBlock * (kWave / G);
    dim3 grid((uint32_t)((a.n_rows + rows_per_block - 1) / rows_per_block));
    if (a.soft) hipLaunchKernelGGL((softmax_xent_kernel<T, G, PER, true>), grid, dim3(kBlock), 0, s, a);
    else hipLaunchKernelGGL((softmax_xent_kernel<T, G, PER, false>), grid, dim3(kBlock), 0, s, a);
    return hipGetLastError();
}

template <typename T>
static hipError_t dispatch_xent(const XentArgs& a, hipStream_t s) {
    const int c = a.n_classes;
    if (c <= 8) return launch_xent<T, 1, 8>(a, s);
    if (c <= 16) return launch_xent<T, 2, 8>(a, s);
    if (c <= 32) return launch_xent<T, 4, 8>(a, s);
    if (c <= 64) return launch_xent<T, 8, 8>(a, s);
    if (c <= 128) return launch_xent<T, 16, 8>(a, s);
    if (c <= 256) return launch_xent<T, 32, 8>(a, s);
    if (c <= 512) return launch_xent<T, 64, 8>(a, s);
    return launch_xent<T, 64, 16>(a, s);
}

}  // namespace dgll

using namespace dgll;

static int xent_impl(void* stream, const void* logits, int64_t ldz, int dtype, const int64_t* labels, const float* soft,
                     int64_t lds, float* row_loss, void* grad, int64_t ldg, const float* grad_scale, int64_t n_rows,
                     int n_classes, int flags = 0) {
    DGLL_REQUIRE(n_rows >= 0 && n_classes >= 0, "negative size");
    if (n_rows == 0 || n_classes == 0) return DGLL_OK;
    DGLL_REQUIRE(logits && (labels || soft), "NULL logits/targets");
    DGLL_REQUIRE(row_loss || grad, "nothing to compute: pass row_loss and/or grad");
    DGLL_REQUIRE(ldz >= n_classes && (!grad || ldg >= n_classes) && (!soft || lds >= n_classes),
                 "leading dimension smaller than n_classes");
    DGLL_REQUIRE(dtype == DGLL_F32 || dtype == DGLL_BF16, "dtype");
    if (n_classes > 1024) {
        set_error("dgll_hip_softmax_xent keeps a row's classes in registers: n_classes <= 1024");
        return DGLL_ERR_UNSUPPORTED;
    }
    XentArgs a{};
    a.z = logits; a.ldz = ldz; a.labels = labels; a.soft = soft; a.lds = lds; a.row_loss = row_loss; a.grad = grad; a.ldg = ldg;
    a.scale = grad_scale; a.n_rows = n_rows; a.n_classes = n_classes; a.mask_nonpositive = flags & 1; a.pad_store = (flags >> 1) & 1;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // bf16 logits with class-index labels and 16-byte aligned rows (what the layers produce): eight consecutive classes per lane
    const int c8 = (n_classes + 7) / 8;
    const bool vec = dtype == DGLL_BF16 && !soft && n_classes <= 512 && aligned16(logits) && (ldz % 8) == 0 && ldz >= 8 * c8 &&
                     (!grad || (aligned16(grad) && (ldg % 8) == 0 && ldg >= n_classes));
    hipError_t e;
    if (vec) {
        if (c8 <= 1) e = launch_xent_vec<1>(a, s);
        else if (c8 <= 2) e = launch_xent_vec<2>(a, s);
        else if (c8 <= 4) e = launch_xent_vec<4>(a, s);
        else if (c8 <= 8) e = launch_xent_vec<8>(a, s);
        else if (c8 <= 16) e = launch_xent_vec<16>(a, s);
        else if (c8 <= 32) e = launch_xent_vec<32>(a, s);
        else e = launch_xent_vec<64>(a, s);
    } else {
        e = dtype == DGLL_F32 ? dispatch_xent<float>(a, s) : dispatch_xent<bf16_t>(a, s);
    }
    if (e != hipSuccess) return hip_fail(e, "softmax_xent_kernel launch");
    return DGLL_OK;
}

DGLL_API int dgll_hip_softmax_xent(void* stream, const void* logits, int64_t ldz, int dtype, const int64_t* labels,
                                   float* row_loss, void* grad, int64_t ldg, const float* grad_scale, int64_t n_rows,
                                   int n_classes) {
    DGLL_REQUIRE(labels || n_rows == 0, "NULL labels");
    return xent_impl(stream, logits, ldz, dtype, labels, nullptr, 0, row_loss, grad, ldg, grad_scale, n_rows, n_classes);
}

DGLL_API int dgll_hip_softmax_xent_soft(void* stream, const void* logits, int64_t ldz, int dtype, const float* targets,
                                        int64_t ldt, float* row_loss, void* grad, int64_t ldg, const float* grad_scale,
                                        int64_t n_rows, int n_classes) {
    DGLL_REQUIRE(targets || n_rows == 0, "NULL targets");
    return xent_impl(stream, logits, ldz, dtype, nullptr, targets, ldt, row_loss, grad, ldg, grad_scale, n_rows, n_classes);
}

// flags bit 1: the gradient rows' padding [n_classes, ldg) belongs to the output and may be written (zeros): the ragged last vector of
// a row is then one 16-byte store instead of up to seven 2-byte ones.
// flags bit 0: the logits are the output of a ReLU and the gradient returned is d loss / d PRE-activation -- zero wherever the
// logit is <= 0 (aten::threshold_backward folded into this pass; the reference's GraphSage ends in a ReLU, sageconv.py:83).
DGLL_API int dgll_hip_softmax_xent_ex(void* stream, const void* logits, int64_t ldz, int dtype, const int64_t* labels,
                                      const float* targets, int64_t ldt, float* row_loss, void* grad, int64_t ldg,
                                      const float* grad_scale, int64_t n_rows, int n_classes, int flags) {
    DGLL_REQUIRE((labels != nullptr) != (targets != nullptr) || n_rows == 0, "pass class-index labels OR probability targets");
    return xent_impl(stream, logits, ldz, dtype, labels, targets, ldt, row_loss, grad, ldg, grad_scale, n_rows, n_classes, flags);
}

// ---- the loss of a mini-batch out of its per-row losses, ONE launch -------------------------------------------------------------------
// nn.CrossEntropyLoss(reduction='mean') over class-index targets divides by the number of targets that are not ignored.  As tensor ops
// that is a zeroed buffer, two compares, an and, a copy, two sums and a division -- nine launches of ~4 us inside a 1.1 ms mini-batch
// step.  One workgroup walks the rows once (fixed order: bit-reproducible) and writes {total, count, total / count, 1 / count}.
namespace dgll {
__global__ __launch_bounds__(kBlock) void xent_reduce_kernel(const float* __restrict__ row_loss, const int64_t* __restrict__ labels,
                                                             int64_t n, int n_classes, float* __restrict__ out) {
    __shared__ float part[2][kBlock];
    float sum = 0.0f, cnt = 0.0f;
    for (int64_t i = threadIdx.x; i < n; i += kBlock) {
        sum += row_loss[i];
        cnt += (!labels || (labels[i] >= 0 && labels[i] < n_classes)) ? 1.0f : 0.0f;
    }
    part[0][threadIdx.x] = sum; part[1][threadIdx.x] = cnt;
    __syncthreads();
    for (int w = kBlock / 2; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) { part[0][threadIdx.x] += part[0][threadIdx.x + w]; part[1][threadIdx.x] += part[1][threadIdx.x + w]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float total = part[0][0], count = part[1][0];
        out[0] = total; out[1] = count; out[2] = total / count; out[3] = 1.0f / count;       // count == 0: NaN / inf, as torch
    }
}
}  // namespace dgll

DGLL_API int dgll_hip_xent_reduce(void* stream, const float* row_loss, const int64_t* labels, int64_t n_rows, int n_classes, float* out4) {
    DGLL_REQUIRE(row_loss && out4 && n_rows >= 0 && n_rows <= (1 << 20), "row losses, a 4-float output, at most 2^20 rows (one workgroup)");
    hipLaunchKernelGGL(dgll::xent_reduce_kernel, dim3(1), dim3(kBlock), 0, static_cast<hipStream_t>(stream), row_loss, labels, n_rows, n_classes,
                       out4);
    DGLL_HIP_TRY(hipGetLastError());
    return DGLL_OK;
}

