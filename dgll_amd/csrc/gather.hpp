// gather.hpp -- the inner gather loop shared by the SpMM kernels (spmm.hip) and the fused aggregate -> transform kernel
// (fused_sage.hip): see spmm.hip's header for the design.
#pragma once
#include "common.hpp"

// spmm.hip: the SpMM behind every dgll_hip_spmm_csr* entry point (only_long: the long rows of the plan only)
int dgll_spmm_csr_impl(void* stream, const dgll_csr_plan* plan, const int64_t* rowptr, const int32_t* col,
                       const float* val, const void* X, int64_t ldx, int x_dtype, void* Y, int64_t ldy,
                       int y_dtype, int64_t n_rows, int64_t n_cols, int feat, int reduce, int epilogue,
                       const float* bias, void* workspace, size_t workspace_bytes, const float* row_scale, int accumulate,
                       const void* gate, int64_t ldg, int only_long);

namespace dgll {

// Accumulate edges [b, e) of one row into acc (this lane's EPV columns starting at xcol).
template <typename XT, int EPV, int LPR, bool HAS_VAL, int U>
__device__ __forceinline__ void gather_edges(const int32_t* __restrict__ col, const float* __restrict__ val,
                                             const XT* __restrict__ xcol, int64_t ldx, int64_t b, int64_t e,
                                             int lane, float (&acc)[EPV], bool preloaded = false, int first_col = 0,
                                             float first_val = 0.0f) {
    typedef VecIO<XT, EPV> IO;
    constexpr int SLOTS = kWave / LPR;
    const int slot = lane / LPR;

    int my_col = first_col;               // preloaded: the caller requested the row's first index batch a row ago
    float my_val = first_val;
    if (!preloaded) {
        my_col = 0;
        my_val = 0.0f;
        if (b + lane < e) {
            my_col = __builtin_nontemporal_load(col + b + lane);   // indices and weights are streamed once: keep them
            if (HAS_VAL) my_val = __builtin_nontemporal_load(val + b + lane);   // from displacing feature rows in L2
        }
    }
    for (int64_t k0 = b; k0 < e; k0 += kWave) {
        const int64_t left = e - k0;
        const int nb = left < kWave ? (int)left : kWave;
        const int cur_col = my_col;
        const float cur_val = my_val;
        // prefetch the next batch of indices while this one is consumed
        const int64_t kn = k0 + kWave + lane;
        if (kn < e) {
            my_col = __builtin_nontemporal_load(col + kn);
            if (HAS_VAL) my_val = __builtin_nontemporal_load(val + kn);
        }
        // Row offsets are formed with ONE 32x32->64 multiply (v_mad_u64_u32): column ids and the leading dimension both
        // fit 32 bits.  Full rounds need no masking; only the last, partial round clamps and zeroes its idle slots.
        const uint32_t ld32 = (uint32_t)ldx;
        int j = 0;
        for (; j + SLOTS * U <= nb; j += SLOTS * U) {
            int c[U];
            float w[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int src = j + u * SLOTS + slot;
                c[u] = __shfl(cur_col, src);
                w[u] = HAS_VAL ? __shfl(cur_val, src) : 1.0f;
            }
            typename IO::raw_t v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = IO::load(xcol + (uint64_t)(uint32_t)c[u] * ld32);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float f[EPV];
                IO::unpack(v[u], f);
#pragma unroll
                for (int i = 0; i < EPV; ++i) acc[i] = HAS_VAL ? fmaf(w[u], f[i], acc[i]) : acc[i] + f[i];
            }
        }
        if (j < nb) {
            // the U gathers are still issued back to back with no branch in between: an out-of-range slot re-reads the
            // batch's last valid edge (same cache lines as a live request) and is zeroed after the load
            int c[U];
            float w[U];
            bool ok[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = j + u * SLOTS + slot;
                ok[u] = idx < nb;
                const int src = ok[u] ? idx : nb - 1;
                c[u] = __shfl(cur_col, src);
                w[u] = HAS_VAL ? __shfl(cur_val, src) : 1.0f;
            }
            typename IO::raw_t v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = IO::load(xcol + (uint64_t)(uint32_t)c[u] * ld32);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float f[EPV];
                IO::unpack(ok[u] ? v[u] : IO::zero(), f);
#pragma unroll
                for (int i = 0; i < EPV; ++i) acc[i] = HAS_VAL ? fmaf(w[u], f[i], acc[i]) : acc[i] + f[i];
            }
        }
    }
    // combine the slots: lanes that differ only in the slot bits hold the same columns
#pragma unroll
    for (int off = LPR; off < kWave; off <<= 1) {
#pragma unroll
        for (int i = 0; i < EPV; ++i) acc[i] += __shfl_xor(acc[i], off);
    }
}

}  // namespace dgll
