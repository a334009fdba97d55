// edge_args.hpp -- launch arguments and wave-level helpers shared by the per-edge kernels (edge.hip, gat.hip).
#pragma once
#include "common.hpp"

namespace dgll {

struct EdgeArgs {
    const int64_t* rowptr;
    const int32_t* col;
    const int64_t* perm;        // A^T edge slot -> A edge slot (only to index edge_scale from the transposed pass)
    const void* H;              // gathered matrix [n_cols, ld]
    int64_t ldh;
    const void* G;              // row-side matrix [n_rows, ld] (sddmm: grad_out; gat_bwd_rows: grad_out)
    int64_t ldg;
    const void* O;              // forward output [n_rows, ld] (gat_bwd_rows)
    int64_t ldo;
    void* Y;                    // main output matrix
    int64_t ldy;
    const float* S;             // [*, heads] row-side scores
    const float* T;             // [*, heads] gathered-side scores
    const float* M;             // [*, heads] row maxima (mode 1) or NULL
    const float* DEN;           // [*, heads] denominators
    const float* DD;            // [*, heads] d(denominator)
    const float* edge_scale;    // [nnz, heads] dropout multipliers or NULL
    float* out_a;               // fp32 [*, heads] output (rowsum / ds / dt)
    float* out_b;               // fp32 [*, heads] output (rowmax / dden)
    float* edge_out;            // fp32 [nnz] (sddmm)
    int64_t n_rows;
    int heads, fo, feat;        // feat = heads * fo
    float alpha, sign;          // leaky-relu slope; -1: exp(-lrelu) (sparseGatConv), +1: softmax(+lrelu) (gatConv)
    int apply_elu, use_max;
    int raw, accumulate;        // partitioned use: raw = leave the row un-normalised (num, den); accumulate = add what is already there
    int exact_dd;               // second-generation rows pass: dd_i from the pass's own dot products (gat_kernel.hpp).  1 = the only
                                // launch over these rows; split over column halves of A: 2 = first launch (partial sums -> part3),
                                // 3 = a middle one (part3 +=), 4 = the last (adds part3, then finalises ds_i, dd_i)
    float* part3;               // fp32 [n_rows, 3 * heads]: (sa, sb, sw) per (row, head) between the launches of a split exact rows pass
    int32_t* arg_out;           // segment_max: int32 [n_rows, ld] source row of the maximum
    // long-row schedule (threshold == 0: none): chunk work items come first in the grid, partials go to `ws`
    int threshold;
    int64_t n_chunks;
    const int64_t* chunk_begin;
    const int64_t* chunk_end;
    const int64_t* chunk_row;
    float* ws;                  // [n_chunks, ws_ld]: [0, feat) vector partial | [ws_vec, +heads) scalar | [+heads, +2 heads) max / 2nd | [+2 heads, +3 heads) 3rd
    int ws_ld, ws_vec;
    uint32_t chunk_blocks;
    int rows_per_wave;          // consecutive rows one wavefront handles before retiring (row items only)
    // second-generation GAT passes (gat_kernel.hpp) only:
    int vph;                    // 16-byte vectors a head really uses (fo / vector width): <= the lanes a head owns
    int tstride;                // node stride, in floats, of the gathered-side score arrays T / DD (compact arrays: heads)
    float* sd_out;              // rows pass, optional: {s_i, dd_i} written side by side, s at [row * sd_stride + head],
    int sd_stride;              //   dd at [row * sd_stride + heads + head]
    // transposed pass, optional: the scores' own contribution to grad_H is added in the epilogue,
    //   grad_H[j, f] += grad_S[j, head(f)] * attn1[f] + grad_T[j, head(f)] * attn2[f]        (S = H.a1, T = H.a2 per head)
    const float* attn1;         // fp32 [heads * fo]: a1 of every head, laid out like a row of H
    const float* attn2;         // fp32 [heads * fo]
    const float* gs_rows;       // fp32 [n_rows of this pass, heads]: grad_S of the pass's rows (from the rows pass)
};

// Row epilogue of the exact rows pass for one (row, head): the launch's sums (sa, sb, sw) = (sum c.dot, sum c, sum w.dot) -- plus the
// partial sums an earlier launch over another column half of A left in part3 -- either go (back) to part3 (more launches follow) or
// are finalised: dd_i = -sw / den_i, ds_i = sa + dd_i sb (bilinear in the sums, so no launch can finalise its own share).
__device__ __forceinline__ void gat_finish_scores(const EdgeArgs& a, int64_t row, int head, float sa, float sb, float sw) {
    float* p3 = a.part3 ? a.part3 + (row * a.heads + head) * 3 : nullptr;
    if (a.exact_dd >= 3) { sa += p3[0]; sb += p3[1]; sw += p3[2]; }
    if (a.exact_dd == 2 || a.exact_dd == 3) { p3[0] = sa; p3[1] = sb; p3[2] = sw; return; }
    const float dd = -sw / a.DEN[row * a.heads + head];
    a.out_a[row * a.heads + head] = sa + dd * sb;
    if (a.out_b) a.out_b[row * a.heads + head] = dd;
    if (a.sd_out) {
        a.sd_out[row * a.sd_stride + head] = a.S[row * a.heads + head];
        a.sd_out[row * a.sd_stride + a.heads + head] = dd;
    }
}

// Work item of this wavefront: a whole (short) row, or one chunk of a long row.
struct WorkItem {
    int64_t row, b, e, chunk;   // chunk < 0: whole row
    bool valid, first, done;    // first: the item starts at the row's first edge (writes the per-row outputs);
};                              // done: nothing more for this wavefront; !valid && !done: skip to the next row

// r-th item of this wavefront (r < rows_per_wave for row items; chunk items are a single item).
__device__ __forceinline__ WorkItem resolve_item(const EdgeArgs& a, int wave, int r) {
    WorkItem w;
    w.chunk = -1;
    w.first = true;
    w.done = false;
    const uint32_t bid = blockIdx.x;
    if (bid < a.chunk_blocks) {
        const int64_t c = __builtin_amdgcn_readfirstlane((int)(bid * kWavesPerBlock + wave));
        w.valid = r == 0 && c < a.n_chunks;
        if (!w.valid) { w.row = w.b = w.e = 0; w.done = true; return w; }
        w.chunk = c;
        w.row = uniform64(a.chunk_row[c]);
        w.b = uniform64(a.chunk_begin[c]);
        w.e = uniform64(a.chunk_end[c]);
        w.first = w.b == uniform64(a.rowptr[w.row]);
        return w;
    }
    w.row = ((int64_t)(bid - a.chunk_blocks) * kWavesPerBlock + wave) * a.rows_per_wave + r;
    w.valid = w.row < a.n_rows;
    if (!w.valid) { w.b = w.e = 0; w.done = true; return w; }
    w.b = uniform64(a.rowptr[w.row]);
    w.e = uniform64(a.rowptr[w.row + 1]);
    if (a.threshold > 0 && w.e - w.b > a.threshold) w.valid = false;   // handled as chunks
    return w;
}

__device__ __forceinline__ float lrelu(float z, float alpha) { return z > 0.0f ? z : alpha * z; }

template <typename T> __device__ __forceinline__ float load_scalar(const T* p);
template <> __device__ __forceinline__ float load_scalar<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float load_scalar<bf16_t>(const bf16_t* p) { return bf16_to_f32(*p); }

// sum over the `lph` adjacent lanes that hold one head's columns (lph is a power of two <= 64)
__device__ __forceinline__ float head_sum(float v, int lph) {
    for (int off = 1; off < lph; off <<= 1) v += __shfl_xor(v, off);
    return v;
}

template <int LPR>
__device__ __forceinline__ float slot_sum(float v) {
#pragma unroll
    for (int off = LPR; off < kWave; off <<= 1) v += __shfl_xor(v, off);
    return v;
}
template <int LPR>
__device__ __forceinline__ float slot_max(float v) {
#pragma unroll
    for (int off = LPR; off < kWave; off <<= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}

// Iterates the edges [b, e) of one row in coalesced batches of 64; `body(j0, nb, cur_col, k0)` is called once
// per batch with the lane-distributed column ids (lane l holds edge k0 + l).
template <typename Body>
__device__ __forceinline__ void for_each_batch(const int32_t* __restrict__ col, int64_t b, int64_t e, int lane, Body body) {
    int my_col = 0;
    if (b + lane < e) my_col = col[b + lane];
    for (int64_t k0 = b; k0 < e; k0 += kWave) {
        const int64_t left = e - k0;
        const int nb = left < kWave ? (int)left : kWave;
        const int cur_col = my_col;
        const int64_t kn = k0 + kWave + lane;
        if (kn < e) my_col = col[kn];
        body(nb, cur_col, k0);
    }
}

// gat_fwd.hip / gat_bwd_rows.hip / gat_bwd_cols.hip: second-generation GAT passes (0 forward, 1 backward over the rows of A,
// 2 backward over the rows of A^T) for `nh` heads per wavefront on `lpr` lanes per row; inrow: the gathered-side scores sit
// in the padding of the gathered rows and arrive with the gather (one head).  False: no such instantiation.
bool gat2_launch_0(int dtype, int lpr, int nh, dim3 grid, hipStream_t s, const EdgeArgs& a, bool inrow);
bool gat2_launch_0r(int dtype, int lpr, int nh, dim3 grid, hipStream_t s, const EdgeArgs& a);             // pass 0, scores from the gathered rows
bool gat2_launch_1(int dtype, int lpr, int nh, dim3 grid, hipStream_t s, const EdgeArgs& a, bool inrow);
bool gat2_launch_3(int dtype, int lpr, int nh, dim3 grid, hipStream_t s, const EdgeArgs& a, bool inrow);   // pass 1, exact-dd form alone
bool gat2_launch_3r(int dtype, int lpr, int nh, dim3 grid, hipStream_t s, const EdgeArgs& a);            // pass 1 exact, scores from the gathered rows
bool gat2_launch_2(int dtype, int lpr, int nh, dim3 grid, hipStream_t s, const EdgeArgs& a, bool inrow);

}  // namespace dgll
