// spmm.hip -- CSR SpMM neighbour aggregation for gfx950 (MI355X), the kernel the whole engine is judged on.
//
//   Y[i, :] = epilogue( reduce_{k in row i} val[k] * X[col[k], :] )
//
// Serves F.spmm / torch.sparse.mm of the reference (dgll/nn/Convolution/gcnconv.py:31,
// Evaluation/PPI/gcn_model.py:76), the K-axis mean of NeighborAggregator (sageconv.py:33-36) and, on the
// transposed structure, every grad_X = A^T.g.
//
// Design: the op is an HBM-bound gather -- no MFMA.
//   * one wavefront (64 lanes) owns one output row at a time; a feature row of the gathered matrix is read
//     by LPR adjacent lanes with one 16-byte load each (global_load_dwordx4), so a 256-wide bf16 row is one
//     512-byte fully coalesced request; the 64/LPR lane groups ("slots") of the wave gather different
//     neighbours concurrently and U loads are kept in flight per lane;
//   * column indices / edge weights of up to 64 edges are fetched with ONE coalesced load per batch and handed
//     to the slots with ds_bpermute (__shfl), so the dependent chain is rowptr -> col batch -> gathers, with
//     the next batch prefetched while the current one is consumed;
//   * fp32 accumulation in registers, cross-slot tree reduction with wave shuffles, fused epilogue
//     (mean scale, bias, ReLU, bf16 down-convert) and one vector store per lane;
//   * rows longer than the plan's threshold are cut into chunks that run as ordinary work items (scheduled
//     FIRST, longest-processing-time style) and write fp32 partials that a tiny second kernel reduces in a
//     fixed order -- no atomics anywhere, results are bit-reproducible;
//   * optional XCD-contiguous row mapping (xcd_remap) for graphs whose neighbours are close in id space; off by
//     default: on RMAT-like inputs (degree correlated with id) it unbalances the XCDs (measured -4 % .. -45 %).
#include <algorithm>
#include <vector>

#include "common.hpp"
#include "gather.hpp"
#include <type_traits>

namespace dgll {

struct LongRow { int64_t row, begin, end; };

__global__ void find_long_rows_kernel(const int64_t* __restrict__ rowptr, int64_t n_rows, int threshold,
                                      LongRow* __restrict__ out, unsigned long long* __restrict__ count,
                                      unsigned long long capacity) {
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = rowptr[r], e = rowptr[r + 1];
        if (e - b > threshold) {
            const unsigned long long at = atomicAdd(count, 1ull);
            if (at < capacity) out[at] = LongRow{r, b, e};
        }
    }
}

// flattened schedule: wave w owns the rows whose KEY rowptr[row] + kFlatRowCost * row lies in [w C, (w + 1) C): balanced on edges
// plus a per-row charge (a row end costs a cross-slot reduction, an epilogue and a store whatever its length -- and a graph's
// rows WITHOUT edges, which a locality order puts side by side by the hundred thousand, would otherwise all fall to one wave).
constexpr int64_t kFlatRowCost = 4;
__global__ void flat_row0_kernel(const int64_t* __restrict__ rowptr, int64_t n_rows, int64_t cost_per_wave, int64_t n_flat,
                                 int64_t* __restrict__ row0) {
    for (int64_t w = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; w <= n_flat; w += (int64_t)gridDim.x * blockDim.x) {
        if (w == n_flat) { row0[w] = n_rows; continue; }
        const int64_t target = w * cost_per_wave;
        int64_t lo = 0, hi = n_rows;               // lower bound of the (strictly increasing) key over rows [0, n_rows)
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (rowptr[mid] + kFlatRowCost * mid < target) lo = mid + 1; else hi = mid;
        }
        row0[w] = lo;
    }
}

struct SpmmArgs {
    const int64_t* rowptr;
    const int32_t* col;
    const float* val;
    const void* X;
    void* Y;
    int64_t ldx, ldy, n_rows;
    int feat, reduce, epilogue;
    const float* bias;
    // long-row schedule (threshold == 0: none)
    int threshold;
    int64_t n_chunks;
    const int64_t* chunk_begin;
    const int64_t* chunk_end;
    float* ws;
    int ws_ld;
    uint32_t chunk_blocks, row_blocks;
    int rows_per_wave;
    int flags;  // bit 0: XCD-contiguous row mapping
    const float* row_scale;   // optional fp32[n_rows]: replaces the reduce's own scale (split adjacencies share one degree)
    int accumulate;           // 1: Y = epi(scale * (A.X + Y));  2: Y += gate(scale * A.X), rows without edges untouched
    const void* gate;         // optional [n_rows, ldg] of Y's type: outputs are zeroed where gate <= 0 (fused ReLU backward)
    int64_t ldg;
    const int64_t* flat_row0; // flattened kernel: first row of every wave's share (plan->d_flat_row0), n_flat + 1 entries
    int64_t n_flat;
};

template <typename T> __device__ __forceinline__ float load_one(const T* p);
template <> __device__ __forceinline__ float load_one<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float load_one<bf16_t>(const bf16_t* p) { return bf16_to_f32(*p); }

// EPV consecutive elements of the OUTPUT type, kept PACKED (one or two 16-byte registers quads, or one scalar) while the
// gather runs and unpacked to floats only in the epilogue.
template <typename YT, int EPV> struct RowVec {
    static constexpr bool kVec = sizeof(YT) * EPV >= 16;
    static constexpr int kN = kVec ? (int)(sizeof(YT) * EPV / 16) : 1;
    uint4 q[kN];
    __device__ __forceinline__ void load(const YT* __restrict__ p) {
        if constexpr (kVec) {
#pragma unroll
            for (int i = 0; i < kN; ++i) q[i] = reinterpret_cast<const uint4*>(p)[i];
        } else {
            q[0].x = (uint32_t)*reinterpret_cast<const typename std::conditional<sizeof(YT) == 2, uint16_t, uint32_t>::type*>(p);
        }
    }
    __device__ __forceinline__ void unpack(float (&f)[EPV]) const {
        if constexpr (!kVec) {
            if constexpr (sizeof(YT) == 2) f[0] = bf16_to_f32((bf16_t)q[0].x);
            else f[0] = __uint_as_float(q[0].x);
        } else if constexpr (sizeof(YT) == 2) {
#pragma unroll
            for (int i = 0; i < kN; ++i) {
                f[8 * i + 0] = bf16_lo(q[i].x); f[8 * i + 1] = bf16_hi(q[i].x); f[8 * i + 2] = bf16_lo(q[i].y); f[8 * i + 3] = bf16_hi(q[i].y);
                f[8 * i + 4] = bf16_lo(q[i].z); f[8 * i + 5] = bf16_hi(q[i].z); f[8 * i + 6] = bf16_lo(q[i].w); f[8 * i + 7] = bf16_hi(q[i].w);
            }
        } else {
#pragma unroll
            for (int i = 0; i < kN; ++i) {
                f[4 * i + 0] = __uint_as_float(q[i].x); f[4 * i + 1] = __uint_as_float(q[i].y);
                f[4 * i + 2] = __uint_as_float(q[i].z); f[4 * i + 3] = __uint_as_float(q[i].w);
            }
        }
    }
};

// Epilogue of one row.  EXTRA = the launch accumulates into Y and/or gates the output (a separate instantiation, so the
// plain aggregation keeps its register budget: 64 VGPRs = 8 waves per SIMD).  `prev` / `gatev` arrive as packed 16-byte
// loads when the lane's EPV columns are all inside the row (`full`); the ragged last vector of a row reads them here,
// element by element.  (Issuing those loads ahead of the gather costs 8 more live registers -- 82 VGPRs, 5 waves -- and
// measured no faster than loading after it at 78 / 6 waves; forcing 72 with waves_per_eu spills and is 11 % slower.)
template <typename YT, int EPV, bool EXTRA>
__device__ __forceinline__ void finish_row(YT* __restrict__ y, int c0, int feat, float scale, int epilogue,
                                           const float* __restrict__ bias, float (&acc)[EPV], int accumulate,
                                           const YT* __restrict__ gate, bool full, const RowVec<YT, EPV>& prev,
                                           const RowVec<YT, EPV>& gatev) {
    if constexpr (EXTRA) {
        if (accumulate == 1) {
            float p[EPV];
            if (full) prev.unpack(p);
#pragma unroll
            for (int i = 0; i < EPV; ++i)
                if (full) acc[i] += p[i];
                else if (c0 + i < feat) acc[i] += load_one<YT>(y + c0 + i);
        }
    }
    float bv[EPV];
    const bool vec_bias = (epilogue & DGLL_EPI_BIAS) && full && EPV >= 4 && (reinterpret_cast<uintptr_t>(bias) & 15u) == 0;
    if (vec_bias) {   // one or two 16-byte loads instead of EPV scalar ones (a per-row cost: 0.5 ms per launch at F = 256)
#pragma unroll
        for (int q = 0; q < EPV / 4; ++q) {
            const float4 b4 = reinterpret_cast<const float4*>(bias + c0)[q];
            bv[4 * q + 0] = b4.x; bv[4 * q + 1] = b4.y; bv[4 * q + 2] = b4.z; bv[4 * q + 3] = b4.w;
        }
    }
#pragma unroll
    for (int i = 0; i < EPV; ++i) {
        float v = acc[i] * scale;
        if (vec_bias) v += bv[i];
        else if ((epilogue & DGLL_EPI_BIAS) && c0 + i < feat) v += bias[c0 + i];
        if (epilogue & DGLL_EPI_RELU) v = fmaxf(v, 0.0f);
        acc[i] = v;
    }
    if constexpr (EXTRA) {
        if (gate) {
            float gv[EPV];
            if (full) gatev.unpack(gv);
#pragma unroll
            for (int i = 0; i < EPV; ++i) {
                if (full) { if (!(gv[i] > 0.0f)) acc[i] = 0.0f; }
                else if (c0 + i < feat && !(load_one<YT>(gate + c0 + i) > 0.0f)) acc[i] = 0.0f;
            }
        }
        if (accumulate == 2) {       // increment form: the (scaled, gated) sum of this launch is added to what Y holds
            float p[EPV];
            if (full) prev.unpack(p);
#pragma unroll
            for (int i = 0; i < EPV; ++i)
                if (full) acc[i] += p[i];
                else if (c0 + i < feat) acc[i] += load_one<YT>(y + c0 + i);
        }
    }
    if (full) {
        if constexpr (std::is_same<YT, bf16_t>::value && EPV == 8) {
            VecIO<YT, EPV>::store_nt(y + c0, acc);      // streaming store (A/B: F = 256 forward 4.42 -> 4.38 ms, never slower)
        } else {
            VecIO<YT, EPV>::store(y + c0, acc);
        }
    } else {
#pragma unroll
        for (int i = 0; i < EPV; ++i)
            if (c0 + i < feat) store_one<YT>(y + c0 + i, acc[i]);
    }
}

template <typename XT, typename YT, int EPV, int LPR, bool HAS_VAL, int U, bool EXTRA, bool PF = false>
__global__ __launch_bounds__(kBlock) void spmm_csr_kernel(const SpmmArgs a) {
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int sub = lane % LPR;
    const int c0 = ((int)blockIdx.y * LPR + sub) * EPV;
    const bool col_ok = c0 < a.feat;
    const XT* xcol = static_cast<const XT*>(a.X) + (col_ok ? c0 : 0);  // idle lanes re-read column 0, never store
    uint32_t bid = blockIdx.x;

    if (bid < a.chunk_blocks) {  // ---- a chunk of a long row: fp32 partial into the workspace
        const int64_t chunk = __builtin_amdgcn_readfirstlane((int)(bid * kWavesPerBlock + wave));
        if (chunk >= a.n_chunks) return;
        float acc[EPV];
#pragma unroll
        for (int i = 0; i < EPV; ++i) acc[i] = 0.0f;
        gather_edges<XT, EPV, LPR, HAS_VAL, U>(a.col, a.val, xcol, a.ldx, uniform64(a.chunk_begin[chunk]),
                                               uniform64(a.chunk_end[chunk]), lane, acc);
        if (lane < LPR && col_ok) {
            float* w = a.ws + chunk * a.ws_ld + c0;
#pragma unroll
            for (int i = 0; i < EPV; ++i) w[i] = acc[i];
        }
        return;
    }

    bid -= a.chunk_blocks;
    if (a.flags & 1) bid = xcd_remap(bid, a.row_blocks);
    const int64_t row0 = ((int64_t)bid * kWavesPerBlock + wave) * a.rows_per_wave;
    // flags bit 2: the NEXT row's first index batch is requested before this row's gathers (one dependent load less per row)
    constexpr bool pf = PF;        // (a template parameter: the two extra live registers cost the default variant a wavefront per SIMD)
    int64_t nb_ = 0, ne_ = 0;
    int ncol = 0;
    float nval = 0.0f;
    if (pf && row0 < a.n_rows) {
        nb_ = uniform64(a.rowptr[row0]); ne_ = uniform64(a.rowptr[row0 + 1]);
        if (nb_ + lane < ne_) {
            ncol = __builtin_nontemporal_load(a.col + nb_ + lane);
            if (HAS_VAL) nval = __builtin_nontemporal_load(a.val + nb_ + lane);
        }
    }
    for (int r = 0; r < a.rows_per_wave; ++r) {
        const int64_t row = row0 + r;
        if (row >= a.n_rows) return;
        int64_t b, e;
        int fcol = 0;
        float fval = 0.0f;
        if (pf) {
            b = nb_; e = ne_; fcol = ncol; fval = nval;
            if (r + 1 < a.rows_per_wave && row + 1 < a.n_rows) {
                nb_ = e; ne_ = uniform64(a.rowptr[row + 2]);
                ncol = 0; nval = 0.0f;
                if (nb_ + lane < ne_) {
                    ncol = __builtin_nontemporal_load(a.col + nb_ + lane);
                    if (HAS_VAL) nval = __builtin_nontemporal_load(a.val + nb_ + lane);
                }
            }
        } else {
            b = uniform64(a.rowptr[row]); e = uniform64(a.rowptr[row + 1]);
        }
        if (a.threshold > 0 && e - b > a.threshold) continue;  // handled as chunks
        if constexpr (EXTRA) {
            if (a.accumulate == 2 && e == b) continue;         // increment form: nothing to add to this row
        }
        float acc[EPV];
#pragma unroll
        for (int i = 0; i < EPV; ++i) acc[i] = 0.0f;
        YT* yrow = static_cast<YT*>(a.Y) + row * a.ldy;
        const bool writer = lane < LPR && col_ok;
        const bool full = c0 + EPV <= a.feat;
        RowVec<YT, EPV> prev, gatev;
        const YT* grow = nullptr;
        if constexpr (EXTRA) {
            grow = a.gate ? static_cast<const YT*>(a.gate) + row * a.ldg : nullptr;
        }
        gather_edges<XT, EPV, LPR, HAS_VAL, U>(a.col, a.val, xcol, a.ldx, b, e, lane, acc, pf, fcol, fval);
        if constexpr (EXTRA) {
            if (writer && full) {
                if (a.accumulate) prev.load(yrow + c0);
                if (grow) gatev.load(grow + c0);
            }
        }
        if (writer) {
            const float scale = a.row_scale ? a.row_scale[row]
                                            : ((a.reduce == DGLL_REDUCE_MEAN && e > b) ? 1.0f / (float)(e - b) : 1.0f);
            finish_row<YT, EPV, EXTRA>(yrow, c0, a.feat, scale, a.epilogue, a.bias, acc, a.accumulate, grow, full, prev, gatev);
        }
    }
}

// ---- flattened variant: a wavefront walks the EDGE STREAM of its rows, not one row after the other ---------------------------
// The wave-per-row kernel drains its gather pipeline at every row end (reduce, epilogue, next row pointers, next index batch):
// rows of 64+ edges run at 0.032-0.036 ns per edge, rows of 16-64 edges at 0.054-0.059, rows under 16 at 0.06-0.09
// (tools/rowlen_probe.py, products-sized bench graph, F = 256 bf16) -- and a fifth of the edges sit in rows under 64.  The same
// edge stream cut into rows of exactly 128 edges takes 3.61 ms where the real rows take 4.37.
// Here wave w owns a run of whole rows of about E edges (plan->d_flat_row0: balanced on edges + 4 per row).  It reads the column
// ids of its share in coalesced batches of 64 REGARDLESS of row boundaries (the next batch requested while the current one is
// consumed), so a row end costs no row-pointer -> index-batch -> gather chain: the row bounds live in scalar registers and a
// group of up to SLOTS x U gathers is cut at the current row's end; after the group that completes a row the slots are reduced,
// the epilogue runs (same code as above) and the accumulators restart.  All of that control flow is wave-uniform.  Rows above the
// plan's threshold are skipped (their chunk items run in the same launch); rows without edges get the epilogue of an empty sum.
// The order in which a row's terms are added is fixed (bit-reproducible run to run) and equals the wave-per-row kernel's for rows
// that lie inside one index batch; other rows differ from it in the last bits.
template <typename XT, typename YT, int EPV, int LPR, bool HAS_VAL, int U, bool EXTRA>
__global__ __launch_bounds__(kBlock) void spmm_csr_flat_kernel(const SpmmArgs a) {
    typedef VecIO<XT, EPV> IO;
    constexpr int SLOTS = kWave / LPR, G = SLOTS * U;
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int sub = lane % LPR, slot = lane / LPR;
    const int c0 = ((int)blockIdx.y * LPR + sub) * EPV;
    const bool col_ok = c0 < a.feat;
    const XT* xcol = static_cast<const XT*>(a.X) + (col_ok ? c0 : 0);
    uint32_t bid = blockIdx.x;

    if (bid < a.chunk_blocks) {  // ---- a chunk of a long row: identical to spmm_csr_kernel's chunk items
        const int64_t chunk = __builtin_amdgcn_readfirstlane((int)(bid * kWavesPerBlock + wave));
        if (chunk >= a.n_chunks) return;
        float acc[EPV];
#pragma unroll
        for (int i = 0; i < EPV; ++i) acc[i] = 0.0f;
        gather_edges<XT, EPV, LPR, HAS_VAL, U>(a.col, a.val, xcol, a.ldx, uniform64(a.chunk_begin[chunk]),
                                               uniform64(a.chunk_end[chunk]), lane, acc);
        if (lane < LPR && col_ok) {
            float* w = a.ws + chunk * a.ws_ld + c0;
#pragma unroll
            for (int i = 0; i < EPV; ++i) w[i] = acc[i];
        }
        return;
    }
    bid -= a.chunk_blocks;
    const int64_t w = (int64_t)bid * kWavesPerBlock + wave;
    if (w >= a.n_flat) return;
    int64_t r = uniform64(a.flat_row0[w]);
    const int64_t r_end = uniform64(a.flat_row0[w + 1]);
    if (r >= r_end) return;
    int64_t rb = uniform64(a.rowptr[r]), re = uniform64(a.rowptr[r + 1]);       // the current row's edges [rb, re)
    const int64_t p_end = uniform64(a.rowptr[r_end]);
    int64_t p = rb;                       // next edge to consume; invariant: acc = sum over the current row's edges [rb, p)
    const bool writer = lane < LPR && col_ok;
    const bool full = c0 + EPV <= a.feat;
    const uint32_t ld32 = (uint32_t)a.ldx;
    const int thr = a.threshold;
    float acc[EPV];
#pragma unroll
    for (int i = 0; i < EPV; ++i) acc[i] = 0.0f;
    bool stop = false;                    // the current row must not be gathered here (long row) or the share is done

    // the current row is complete in acc (per slot): reduce across the slots, epilogue, store; move on to the next row
    auto flush = [&]() {
#pragma unroll
        for (int off = LPR; off < kWave; off <<= 1) {
#pragma unroll
            for (int i = 0; i < EPV; ++i) acc[i] += __shfl_xor(acc[i], off);
        }
        bool emit = writer;
        if constexpr (EXTRA) {
            if (a.accumulate == 2 && re == rb) emit = false;       // increment form: nothing to add to a row without edges
        }
        if (emit) {
            YT* yrow = static_cast<YT*>(a.Y) + r * a.ldy;
            RowVec<YT, EPV> prev, gatev;
            const YT* grow = nullptr;
            if constexpr (EXTRA) {
                grow = a.gate ? static_cast<const YT*>(a.gate) + r * a.ldg : nullptr;
                if (full) {
                    if (a.accumulate) prev.load(yrow + c0);
                    if (grow) gatev.load(grow + c0);
                }
            }
            const float scale = a.row_scale ? a.row_scale[r]
                                            : ((a.reduce == DGLL_REDUCE_MEAN && re > rb) ? 1.0f / (float)(re - rb) : 1.0f);
            finish_row<YT, EPV, EXTRA>(yrow, c0, a.feat, scale, a.epilogue, a.bias, acc, a.accumulate, grow, full, prev, gatev);
        }
#pragma unroll
        for (int i = 0; i < EPV; ++i) acc[i] = 0.0f;
        ++r;
        rb = re;
        if (r < r_end) {
            re = uniform64(a.rowptr[r + 1]);
            stop = thr > 0 && re - rb > thr;
        } else {
            stop = true;
        }
    };

    stop = thr > 0 && re - rb > thr;
    // index batches: 64 consecutive column ids of the share, the next batch requested while the current one is consumed
    int my_col = 0, nx_col = 0;
    float my_val = 0.0f, nx_val = 0.0f;
    int64_t nx_p = -1;                    // position the prefetched batch starts at (-1: none)
    while (r < r_end) {
        if (stop) {                       // a long row: its chunk items do the work; nothing of it was consumed here
            p = re;
            ++r;
            rb = re;
            stop = false;
            if (r < r_end) {
                re = uniform64(a.rowptr[r + 1]);
                stop = thr > 0 && re - rb > thr;
            }
            continue;
        }
        if (re == rb) {                   // a row without edges: its epilogue of an empty sum
            flush();
            continue;
        }
        const int64_t left = p_end - p;
        const int nb = left < kWave ? (int)left : kWave;
        if (nx_p == p) {
            my_col = nx_col;
            my_val = nx_val;
        } else {
            my_col = 0;
            my_val = 0.0f;
            if (lane < nb) {
                my_col = __builtin_nontemporal_load(a.col + p + lane);
                if (HAS_VAL) my_val = __builtin_nontemporal_load(a.val + p + lane);
            }
        }
        nx_p = p + kWave;
        nx_col = 0;
        nx_val = 0.0f;
        if (nx_p + lane < p_end) {
            nx_col = __builtin_nontemporal_load(a.col + nx_p + lane);
            if (HAS_VAL) nx_val = __builtin_nontemporal_load(a.val + nx_p + lane);
        }
        int j = 0;
        while (j < nb && !stop) {
            // one group = up to SLOTS x U edges of the CURRENT row: a group never crosses a row end, so one flush site serves all
            const int64_t q0 = p + j;
            const int64_t in_row = re - q0;                        // >= 1: the current row has edges left (invariant)
            int cnt = nb - j < G ? nb - j : G;
            if (in_row < cnt) cnt = (int)in_row;
            if (cnt == G) {
                int c[U];
                float wv[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int src = j + u * SLOTS + slot;
                    c[u] = __shfl(my_col, src);
                    wv[u] = HAS_VAL ? __shfl(my_val, src) : 1.0f;
                }
                typename IO::raw_t v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) v[u] = IO::load(xcol + (uint64_t)(uint32_t)c[u] * ld32);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    float f[EPV];
                    IO::unpack(v[u], f);
#pragma unroll
                    for (int i = 0; i < EPV; ++i) acc[i] = HAS_VAL ? fmaf(wv[u], f[i], acc[i]) : acc[i] + f[i];
                }
            } else {
                // the U gathers are still issued back to back: an idle slot re-reads the group's last live edge and is zeroed after the load
                int c[U];
                float wv[U];
                bool ok[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int idx = u * SLOTS + slot;
                    ok[u] = idx < cnt;
                    const int src = j + (ok[u] ? idx : cnt - 1);
                    c[u] = __shfl(my_col, src);
                    wv[u] = HAS_VAL ? __shfl(my_val, src) : 1.0f;
                }
                typename IO::raw_t v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) v[u] = IO::load(xcol + (uint64_t)(uint32_t)c[u] * ld32);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    float f[EPV];
                    IO::unpack(ok[u] ? v[u] : IO::zero(), f);
#pragma unroll
                    for (int i = 0; i < EPV; ++i) acc[i] = HAS_VAL ? fmaf(wv[u], f[i], acc[i]) : acc[i] + f[i];
                }
            }
            j += cnt;
            if (q0 + cnt == re) {                                   // the row is complete
                flush();
                while (!stop && re == rb) flush();                  // rows without edges that follow it
            }
        }
        // what the batch consumed: j edges, unless a long row (or the end of the share) stopped it -- then the current row starts at rb
        p = stop ? rb : p + j;
    }
}

// ---- short-row / narrow-row variant: one output row per SLOT, 64 / LPR rows per wavefront at a time ---------------------
// The wave-per-row kernel above pays a fixed price per row -- row pointers, one index batch, a cross-slot shuffle tree
// (log2(SLOTS) x EPV shuffles), the epilogue -- and runs those prices one row after the other.  That is noise for a 256-wide
// row with 50 neighbours; it is most of the time for narrow rows (F <= 64: 8 slots, 24 shuffles per row) and for short rows
// (the halo halves of a partitioned graph average 3 - 9 edges per row).  Here every lane group of LPR lanes owns a row of its
// own: the SLOTS rows of a wavefront advance together, each slot walks ITS row U edges at a time (one 16-byte read of U
// column ids, U feature-row gathers), nothing is reduced across lanes, and the epilogues of SLOTS rows are one pass.  Rows
// above the plan's threshold are skipped here and run as chunks exactly as above (same launch, same workspace).
template <typename XT, typename YT, int EPV, int LPR, bool HAS_VAL, bool EXTRA>
__global__ __launch_bounds__(kBlock) void spmm_rowslot_kernel(const SpmmArgs a) {
    typedef VecIO<XT, EPV> IO;
    constexpr int SLOTS = kWave / LPR;
    constexpr int U = 4;
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int sub = lane % LPR, slot = lane / LPR;
    const int c0 = ((int)blockIdx.y * LPR + sub) * EPV;
    const bool col_ok = c0 < a.feat;
    const XT* xcol = static_cast<const XT*>(a.X) + (col_ok ? c0 : 0);
    uint32_t bid = blockIdx.x;

    if (bid < a.chunk_blocks) {  // ---- a chunk of a long row: identical to spmm_csr_kernel's chunk items
        const int64_t chunk = __builtin_amdgcn_readfirstlane((int)(bid * kWavesPerBlock + wave));
        if (chunk >= a.n_chunks) return;
        float acc[EPV];
#pragma unroll
        for (int i = 0; i < EPV; ++i) acc[i] = 0.0f;
        gather_edges<XT, EPV, LPR, HAS_VAL, U>(a.col, a.val, xcol, a.ldx, uniform64(a.chunk_begin[chunk]),
                                               uniform64(a.chunk_end[chunk]), lane, acc);
        if (lane < LPR && col_ok) {
            float* w = a.ws + chunk * a.ws_ld + c0;
#pragma unroll
            for (int i = 0; i < EPV; ++i) w[i] = acc[i];
        }
        return;
    }
    bid -= a.chunk_blocks;
    const uint32_t ld32 = (uint32_t)a.ldx;
    const int64_t row0 = ((int64_t)bid * kWavesPerBlock + wave) * a.rows_per_wave;     // rows_per_wave is a multiple of SLOTS
    for (int r = 0; r < a.rows_per_wave; r += SLOTS) {
        if (row0 + r >= a.n_rows) return;
        const int64_t row = row0 + r + slot;
        int64_t b = 0, e = 0;
        if (row < a.n_rows) { b = a.rowptr[row]; e = a.rowptr[row + 1]; }
        const int64_t full_len = e - b;
        bool mine = row < a.n_rows && !(a.threshold > 0 && full_len > a.threshold);    // long rows: handled as chunks
        if constexpr (EXTRA) {
            if (a.accumulate == 2 && full_len == 0) mine = false;                        // increment form: nothing to add
        }
        const int n = mine ? (int)full_len : 0;
        float acc[EPV];
#pragma unroll
        for (int i = 0; i < EPV; ++i) acc[i] = 0.0f;
        // this slot's column ids, U at a time; the next group is fetched while the current one is gathered
        int cnext[U];
        float wnext[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            cnext[u] = u < n ? __builtin_nontemporal_load(a.col + b + u) : 0;
            wnext[u] = (HAS_VAL && u < n) ? __builtin_nontemporal_load(a.val + b + u) : 0.0f;
        }
        for (int k = 0; __any(k < n); k += U) {
            int c[U];
            float w[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { c[u] = cnext[u]; w[u] = wnext[u]; }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int kn = k + U + u;
                cnext[u] = kn < n ? __builtin_nontemporal_load(a.col + b + kn) : 0;
                wnext[u] = (HAS_VAL && kn < n) ? __builtin_nontemporal_load(a.val + b + kn) : 0.0f;
            }
            typename IO::raw_t v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = IO::load(xcol + (uint64_t)(uint32_t)c[u] * ld32);   // idle slots re-read row 0
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float f[EPV];
                IO::unpack(k + u < n ? v[u] : IO::zero(), f);
#pragma unroll
                for (int i = 0; i < EPV; ++i) acc[i] = HAS_VAL ? fmaf(w[u], f[i], acc[i]) : acc[i] + f[i];
            }
        }
        if (!mine || !col_ok) continue;
        YT* yrow = static_cast<YT*>(a.Y) + row * a.ldy;
        const bool full = c0 + EPV <= a.feat;
        RowVec<YT, EPV> prev, gatev;
        const YT* grow = nullptr;
        if constexpr (EXTRA) {
            grow = a.gate ? static_cast<const YT*>(a.gate) + row * a.ldg : nullptr;
            if (full) {
                if (a.accumulate) prev.load(yrow + c0);
                if (grow) gatev.load(grow + c0);
            }
        }
        const float scale = a.row_scale ? a.row_scale[row]
                                        : ((a.reduce == DGLL_REDUCE_MEAN && n > 0) ? 1.0f / (float)n : 1.0f);
        finish_row<YT, EPV, EXTRA>(yrow, c0, a.feat, scale, a.epilogue, a.bias, acc, a.accumulate, grow, full, prev, gatev);
    }
}

// Second pass for long rows: sum the chunk partials in chunk order, then the same epilogue.  One wavefront per long row,
// four columns per lane (float4 reads of the partials); most long rows have only two or three chunks.
template <typename YT>
__global__ __launch_bounds__(kBlock) void spmm_long_finalize_kernel(const SpmmArgs a, const int64_t* __restrict__ long_row,
                                                                    const int32_t* __restrict__ long_chunk0, int64_t n_long) {
    const int lane = lane_id();
    const int64_t li = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (li >= n_long) return;
    const int64_t row = long_row[li];
    const int cb = long_chunk0[li], ce = long_chunk0[li + 1];
    float scale = 1.0f;
    if (a.row_scale) scale = a.row_scale[row];
    else if (a.reduce == DGLL_REDUCE_MEAN) scale = 1.0f / (float)(a.rowptr[row + 1] - a.rowptr[row]);
    YT* y = static_cast<YT*>(a.Y) + row * a.ldy;
    const YT* gate = a.gate ? static_cast<const YT*>(a.gate) + row * a.ldg : nullptr;
    for (int f = lane * 4; f < a.feat; f += kWave * 4) {   // ws_ld is a multiple of 8 floats: the float4 stays in the row
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        // eight partial rows in flight, added in chunk order (a hub of some ten thousand edges has dozens of chunks: one dependent
        // load per chunk made this kernel's run time the latency of the longest row -- 45-50 us per launch on an 8-way partition)
        for (int c = cb; c < ce; c += 8) {
            float4 p[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int cc = c + u < ce ? c + u : ce - 1;
                p[u] = *reinterpret_cast<const float4*>(a.ws + (int64_t)cc * a.ws_ld + f);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (c + u < ce) { s.x += p[u].x; s.y += p[u].y; s.z += p[u].z; s.w += p[u].w; }
        }
        float v[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (f + i >= a.feat) continue;
            float t = v[i];
            if (a.accumulate == 1) t += load_one<YT>(y + f + i);
            t *= scale;
            if (a.epilogue & DGLL_EPI_BIAS) t += a.bias[f + i];
            if (a.epilogue & DGLL_EPI_RELU) t = fmaxf(t, 0.0f);
            if (gate && !(load_one<YT>(gate + f + i) > 0.0f)) t = 0.0f;
            if (a.accumulate == 2) t += load_one<YT>(y + f + i);
            v[i] = t;
        }
        if (f + 4 <= a.feat && (a.flags & 2)) VecIO<YT, 4>::store(y + f, v);     // flags bit 1: rows are 16-byte aligned
        else
            for (int i = 0; i < 4; ++i) if (f + i < a.feat) store_one<YT>(y + f + i, v[i]);
    }
}

static int ws_ld_for(int feat) { return (feat + 7) & ~7; }

// Tuning knobs (diagnostics; defaults are the shipped configuration).  Set through dgll_hip_debug_tune().
static int g_tune_unroll = 4;        // gathers in flight per lane (2, 4 or 8; 8 only for the widest variants)
static int g_tune_rows_per_wave = 0; // 0 = automatic
static int g_tune_flags = 0;         // bit 0: XCD-contiguous row mapping (off: measured slower when degree correlates with row id)
static int g_tune_threshold = 0;     // 0 = plan default (256)
static int g_tune_rowslot = 0;       // 0 = automatic choice of the row-per-slot kernel, 1 = never, 2 = whenever it applies
static int g_tune_flat = 0;          // flattened kernel (spmm_csr_flat_kernel): 0 = automatic, 1 = never, 2 = whenever the plan has its schedule
static int g_tune_flat_edges = 256;  // edges per wave of the flattened schedule (read when a plan is created)

template <typename XT, typename YT, int EPV, int LPR, int U>
static hipError_t launch_u(const SpmmArgs& a, dim3 grid, hipStream_t s) {
    const bool extra = a.accumulate || a.gate;
    if constexpr (LPR == 32 && U == 4 && sizeof(XT) == 2 && sizeof(YT) == 2) {
        if (a.flags & 4) {       // diagnostics (dgll_hip_debug_tune(2, 4)): next-row index prefetch
            if (a.val) {
                if (extra) hipLaunchKernelGGL((spmm_csr_kernel<XT, YT, EPV, LPR, true, U, true, true>), grid, dim3(kBlock), 0, s, a);
                else hipLaunchKernelGGL((spmm_csr_kernel<XT, YT, EPV, LPR, true, U, false, true>), grid, dim3(kBlock), 0, s, a);
            } else {
                if (extra) hipLaunchKernelGGL((spmm_csr_kernel<XT, YT, EPV, LPR, false, U, true, true>), grid, dim3(kBlock), 0, s, a);
                else hipLaunchKernelGGL((spmm_csr_kernel<XT, YT, EPV, LPR, false, U, false, true>), grid, dim3(kBlock), 0, s, a);
            }
            return hipGetLastError();
        }
    }
    if (a.val) {
        if (extra) hipLaunchKernelGGL((spmm_csr_kernel<XT, YT, EPV, LPR, true, U, true>), grid, dim3(kBlock), 0, s, a);
        else hipLaunchKernelGGL((spmm_csr_kernel<XT, YT, EPV, LPR, true, U, false>), grid, dim3(kBlock), 0, s, a);
    } else {
        if (extra) hipLaunchKernelGGL((spmm_csr_kernel<XT, YT, EPV, LPR, false, U, true>), grid, dim3(kBlock), 0, s, a);
        else hipLaunchKernelGGL((spmm_csr_kernel<XT, YT, EPV, LPR, false, U, false>), grid, dim3(kBlock), 0, s, a);
    }
    return hipGetLastError();
}

template <typename XT, typename YT, int EPV, int LPR>
static hipError_t launch_variant(const SpmmArgs& a, dim3 grid, hipStream_t s) {
    if constexpr (LPR >= 32 && EPV > 1) {  // the wide-row variants also exist with other unroll depths
        if (g_tune_unroll == 8) return launch_u<XT, YT, EPV, LPR, 8>(a, grid, s);
        if (g_tune_unroll == 2) return launch_u<XT, YT, EPV, LPR, 2>(a, grid, s);
    }
    return launch_u<XT, YT, EPV, LPR, 4>(a, grid, s);
}

template <typename XT, typename YT, int EPV, int LPR>
static hipError_t launch_rowslot(const SpmmArgs& a, dim3 grid, hipStream_t s) {
    const bool extra = a.accumulate || a.gate;
    if (a.val) {
        if (extra) hipLaunchKernelGGL((spmm_rowslot_kernel<XT, YT, EPV, LPR, true, true>), grid, dim3(kBlock), 0, s, a);
        else hipLaunchKernelGGL((spmm_rowslot_kernel<XT, YT, EPV, LPR, true, false>), grid, dim3(kBlock), 0, s, a);
    } else {
        if (extra) hipLaunchKernelGGL((spmm_rowslot_kernel<XT, YT, EPV, LPR, false, true>), grid, dim3(kBlock), 0, s, a);
        else hipLaunchKernelGGL((spmm_rowslot_kernel<XT, YT, EPV, LPR, false, false>), grid, dim3(kBlock), 0, s, a);
    }
    return hipGetLastError();
}

template <typename XT, typename YT, int EPV>
static hipError_t launch_rowslot_lpr(const SpmmArgs& a, int lpr, dim3 grid, hipStream_t s) {
    switch (lpr) {
        case 4: return launch_rowslot<XT, YT, EPV, 4>(a, grid, s);
        case 8: return launch_rowslot<XT, YT, EPV, 8>(a, grid, s);
        case 16: return launch_rowslot<XT, YT, EPV, 16>(a, grid, s);
        default: return launch_rowslot<XT, YT, EPV, 32>(a, grid, s);
    }
}

template <typename XT, typename YT, int EPV, int LPR>
static hipError_t launch_flat(const SpmmArgs& a, dim3 grid, hipStream_t s) {
    const bool extra = a.accumulate || a.gate;
    if (a.val) {
        if (extra) hipLaunchKernelGGL((spmm_csr_flat_kernel<XT, YT, EPV, LPR, true, 4, true>), grid, dim3(kBlock), 0, s, a);
        else hipLaunchKernelGGL((spmm_csr_flat_kernel<XT, YT, EPV, LPR, true, 4, false>), grid, dim3(kBlock), 0, s, a);
    } else {
        if (extra) hipLaunchKernelGGL((spmm_csr_flat_kernel<XT, YT, EPV, LPR, false, 4, true>), grid, dim3(kBlock), 0, s, a);
        else hipLaunchKernelGGL((spmm_csr_flat_kernel<XT, YT, EPV, LPR, false, 4, false>), grid, dim3(kBlock), 0, s, a);
    }
    return hipGetLastError();
}

template <typename XT, typename YT, int EPV>
static hipError_t launch_flat_lpr(const SpmmArgs& a, int lpr, dim3 grid, hipStream_t s) {
    if (lpr == 16) return launch_flat<XT, YT, EPV, 16>(a, grid, s);
    return launch_flat<XT, YT, EPV, 32>(a, grid, s);
}

template <typename XT, typename YT, int EPV>
static hipError_t launch_lpr(const SpmmArgs& a, int lpr, dim3 grid, hipStream_t s) {
    switch (lpr) {
        case 4: return launch_variant<XT, YT, EPV, 4>(a, grid, s);
        case 8: return launch_variant<XT, YT, EPV, 8>(a, grid, s);
        case 16: return launch_variant<XT, YT, EPV, 16>(a, grid, s);
        case 32: return launch_variant<XT, YT, EPV, 32>(a, grid, s);
        default: return launch_variant<XT, YT, EPV, 64>(a, grid, s);
    }
}

}  // namespace dgll

using namespace dgll;

DGLL_API int dgll_hip_csr_plan_create(void* stream, const int64_t* rowptr, int64_t n_rows, int64_t nnz,
                                      int long_row_threshold, dgll_csr_plan** out_plan) {
    DGLL_REQUIRE(out_plan != nullptr, "out_plan is NULL");
    DGLL_REQUIRE(rowptr != nullptr && n_rows >= 0 && nnz >= 0, "bad CSR arguments");
    hipStream_t s = static_cast<hipStream_t>(stream);
    dgll_csr_plan* p = new dgll_csr_plan();
    p->n_rows = n_rows;
    p->nnz = nnz;
    p->threshold = long_row_threshold > 0 ? long_row_threshold : (g_tune_threshold > 0 ? g_tune_threshold : 256);
    hipError_t e = hipGetDevice(&p->device);
    if (e != hipSuccess) { delete p; return hip_fail(e, "hipGetDevice"); }
    if (long_row_threshold < 0) {   // caller's guarantee: no row is longer than the default threshold (e.g. a sampled block with
        p->threshold = 0;           // fan-out <= 128) -- no device scan, no allocation, no synchronisation.  threshold 0 =
        *out_plan = p;              // "no chunk schedule": every row is gathered inline whatever its length, so a wrong
        return DGLL_OK;             // bound (or the debug knob) can cost speed but never leaves a row unwritten
    }

    const unsigned long long capacity = (unsigned long long)(nnz / p->threshold) + 1;
    LongRow* d_list = nullptr;
    unsigned long long* d_count = nullptr;
    std::vector<LongRow> h_list;
    unsigned long long h_count = 0;
    auto cleanup = [&]() { if (d_list) (void)hipFree(d_list); if (d_count) (void)hipFree(d_count); };
#define PLAN_TRY(expr)                                                                   \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) { cleanup(); dgll_hip_csr_plan_destroy(p); return hip_fail(_e, #expr); } \
    } while (0)
    PLAN_TRY(hipMalloc(&d_list, capacity * sizeof(LongRow)));
    PLAN_TRY(hipMalloc(&d_count, sizeof(unsigned long long)));
    PLAN_TRY(hipMemsetAsync(d_count, 0, sizeof(unsigned long long), s));
    if (n_rows > 0) {
        const int blocks = (int)std::min<int64_t>((n_rows + kBlock - 1) / kBlock, 2048);
        hipLaunchKernelGGL(find_long_rows_kernel, dim3(blocks), dim3(kBlock), 0, s, rowptr, n_rows, p->threshold,
                           d_list, d_count, capacity);
        PLAN_TRY(hipGetLastError());
    }
    PLAN_TRY(hipMemcpyAsync(&h_count, d_count, sizeof(h_count), hipMemcpyDeviceToHost, s));
    PLAN_TRY(hipStreamSynchronize(s));
    if (h_count > capacity) {
        cleanup();
        dgll_hip_csr_plan_destroy(p);
        set_error("rowptr is inconsistent with nnz (more long rows than nnz allows)");
        return DGLL_ERR_INVALID;
    }
    h_list.resize(h_count);
    if (h_count) {
        PLAN_TRY(hipMemcpyAsync(h_list.data(), d_list, h_count * sizeof(LongRow), hipMemcpyDeviceToHost, s));
        PLAN_TRY(hipStreamSynchronize(s));
    }
    std::sort(h_list.begin(), h_list.end(), [](const LongRow& x, const LongRow& y) { return x.row < y.row; });

    std::vector<int64_t> long_row(h_count), chunk_begin, chunk_end, chunk_row;
    std::vector<int32_t> chunk0(h_count + 1, 0);
    for (size_t i = 0; i < h_count; ++i) {
        long_row[i] = h_list[i].row;
        chunk0[i] = (int32_t)chunk_begin.size();
        for (int64_t b = h_list[i].begin; b < h_list[i].end; b += p->threshold) {
            chunk_begin.push_back(b);
            chunk_end.push_back(std::min<int64_t>(b + p->threshold, h_list[i].end));
            chunk_row.push_back(h_list[i].row);
        }
    }
    chunk0[h_count] = (int32_t)chunk_begin.size();
    p->n_long = (int64_t)h_count;
    p->n_chunks = (int64_t)chunk_begin.size();
    if (p->n_long > 0) {
        PLAN_TRY(hipMalloc(&p->d_long_row, sizeof(int64_t) * h_count));
        PLAN_TRY(hipMalloc(&p->d_long_chunk0, sizeof(int32_t) * (h_count + 1)));
        PLAN_TRY(hipMalloc(&p->d_chunk_begin, sizeof(int64_t) * chunk_begin.size()));
        PLAN_TRY(hipMalloc(&p->d_chunk_end, sizeof(int64_t) * chunk_end.size()));
        PLAN_TRY(hipMalloc(&p->d_chunk_row, sizeof(int64_t) * chunk_row.size()));
        PLAN_TRY(hipMemcpyAsync(p->d_chunk_row, chunk_row.data(), sizeof(int64_t) * chunk_row.size(), hipMemcpyHostToDevice, s));
        PLAN_TRY(hipMemcpyAsync(p->d_long_row, long_row.data(), sizeof(int64_t) * h_count, hipMemcpyHostToDevice, s));
        PLAN_TRY(hipMemcpyAsync(p->d_long_chunk0, chunk0.data(), sizeof(int32_t) * (h_count + 1), hipMemcpyHostToDevice, s));
        PLAN_TRY(hipMemcpyAsync(p->d_chunk_begin, chunk_begin.data(), sizeof(int64_t) * chunk_begin.size(), hipMemcpyHostToDevice, s));
        PLAN_TRY(hipMemcpyAsync(p->d_chunk_end, chunk_end.data(), sizeof(int64_t) * chunk_end.size(), hipMemcpyHostToDevice, s));
        PLAN_TRY(hipStreamSynchronize(s));
    }
    // the flattened kernel's wave schedule (one binary search per wave, once per graph)
    if (n_rows > 0 && g_tune_flat_edges > 0) {
        p->flat_edges = g_tune_flat_edges;
        p->n_flat = (nnz + kFlatRowCost * n_rows + p->flat_edges - 1) / p->flat_edges;
        PLAN_TRY(hipMalloc(&p->d_flat_row0, sizeof(int64_t) * (size_t)(p->n_flat + 1)));
        const int blocks = (int)std::min<int64_t>((p->n_flat + 1 + kBlock - 1) / kBlock, 4096);
        hipLaunchKernelGGL(flat_row0_kernel, dim3(blocks), dim3(kBlock), 0, s, rowptr, n_rows, (int64_t)p->flat_edges, p->n_flat,
                           p->d_flat_row0);
        PLAN_TRY(hipGetLastError());
        PLAN_TRY(hipStreamSynchronize(s));
    }
#undef PLAN_TRY
    cleanup();
    *out_plan = p;
    return DGLL_OK;
}

extern int g_tune_mfma_kperm;   // dense.hip
extern int g_tune_gat_gen;      // edge.hip
extern int g_tune_res_per_cu;
extern int g_tune_loader_blocks_per_cu;   // gather.hip

DGLL_API int dgll_hip_debug_tune(int key, int value) {
    switch (key) {
        case 0: g_tune_unroll = value; break;
        case 1: g_tune_rows_per_wave = value; break;
        case 2: g_tune_flags = value; break;
        case 3: g_tune_threshold = value; break;
        case 4: g_tune_mfma_kperm = value; break;
        case 5: g_tune_rowslot = value; break;
        case 13: g_tune_flat = value; break;
        case 14: g_tune_flat_edges = value; break;
        case 7: break;                              // (retired: unroll depth of the first-generation GAT backward passes)
        case 9: g_tune_gat_gen = value; break;
        case 11: g_tune_res_per_cu = value; break;
        case 12: g_tune_loader_blocks_per_cu = value; break;
        default: set_error("unknown tuning key"); return DGLL_ERR_INVALID;
    }
    return DGLL_OK;
}

DGLL_API void dgll_hip_csr_plan_destroy(dgll_csr_plan* p) {
    if (!p) return;
    if (p->d_long_row) (void)hipFree(p->d_long_row);
    if (p->d_long_chunk0) (void)hipFree(p->d_long_chunk0);
    if (p->d_chunk_begin) (void)hipFree(p->d_chunk_begin);
    if (p->d_chunk_end) (void)hipFree(p->d_chunk_end);
    if (p->d_chunk_row) (void)hipFree(p->d_chunk_row);
    if (p->d_flat_row0) (void)hipFree(p->d_flat_row0);
    delete p;
}

DGLL_API size_t dgll_hip_csr_plan_workspace_bytes(const dgll_csr_plan* p, int feat) {
    if (!p || p->n_chunks == 0 || feat <= 0) return 0;
    return (size_t)p->n_chunks * (size_t)ws_ld_for(feat) * sizeof(float);
}

DGLL_API int64_t dgll_hip_csr_plan_num_long_rows(const dgll_csr_plan* p) { return p ? p->n_long : 0; }
DGLL_API int64_t dgll_hip_csr_plan_num_chunks(const dgll_csr_plan* p) { return p ? p->n_chunks : 0; }


DGLL_API int dgll_hip_spmm_csr(void* stream, const dgll_csr_plan* plan, const int64_t* rowptr, const int32_t* col,
                               const float* val, const void* X, int64_t ldx, int x_dtype, void* Y, int64_t ldy,
                               int y_dtype, int64_t n_rows, int64_t n_cols, int feat, int reduce, int epilogue,
                               const float* bias, void* workspace, size_t workspace_bytes) {
    return dgll_spmm_csr_impl(stream, plan, rowptr, col, val, X, ldx, x_dtype, Y, ldy, y_dtype, n_rows, n_cols, feat, reduce,
                              epilogue, bias, workspace, workspace_bytes, nullptr, 0, nullptr, 0, 0);
}

DGLL_API int dgll_hip_spmm_csr_ex(void* stream, const dgll_csr_plan* plan, const int64_t* rowptr, const int32_t* col,
                                  const float* val, const void* X, int64_t ldx, int x_dtype, void* Y, int64_t ldy,
                                  int y_dtype, int64_t n_rows, int64_t n_cols, int feat, int reduce, int epilogue,
                                  const float* bias, void* workspace, size_t workspace_bytes, const float* row_scale,
                                  int accumulate) {
    return dgll_spmm_csr_impl(stream, plan, rowptr, col, val, X, ldx, x_dtype, Y, ldy, y_dtype, n_rows, n_cols, feat, reduce,
                              epilogue, bias, workspace, workspace_bytes, row_scale, accumulate, nullptr, 0, 0);
}

DGLL_API int dgll_hip_spmm_csr_gated(void* stream, const dgll_csr_plan* plan, const int64_t* rowptr, const int32_t* col,
                                     const float* val, const void* X, int64_t ldx, int x_dtype, void* Y, int64_t ldy,
                                     int y_dtype, int64_t n_rows, int64_t n_cols, int feat, int reduce, int epilogue,
                                     const float* bias, void* workspace, size_t workspace_bytes, const float* row_scale,
                                     int accumulate, const void* gate, int64_t ldg) {
    return dgll_spmm_csr_impl(stream, plan, rowptr, col, val, X, ldx, x_dtype, Y, ldy, y_dtype, n_rows, n_cols, feat, reduce,
                              epilogue, bias, workspace, workspace_bytes, row_scale, accumulate, gate, ldg, 0);
}

// only_long != 0: just the rows longer than the plan's threshold (their chunk items + the finalize pass) -- the fused
// aggregate -> transform kernel (fused_sage.hip) gathers every other row itself.
int dgll_spmm_csr_impl(void* stream, const dgll_csr_plan* plan, const int64_t* rowptr, const int32_t* col,
                       const float* val, const void* X, int64_t ldx, int x_dtype, void* Y, int64_t ldy,
                       int y_dtype, int64_t n_rows, int64_t n_cols, int feat, int reduce, int epilogue,
                       const float* bias, void* workspace, size_t workspace_bytes, const float* row_scale, int accumulate,
                       const void* gate, int64_t ldg, int only_long) {
    DGLL_REQUIRE(n_rows >= 0 && n_cols >= 0 && feat >= 0, "negative size");
    DGLL_REQUIRE(!gate || ldg >= feat, "gate leading dimension smaller than feat");
    DGLL_REQUIRE(accumulate >= 0 && accumulate <= 2, "accumulate: 0, 1 (add Y before the epilogue) or 2 (add the epilogue's result to Y)");
    DGLL_REQUIRE(accumulate != 2 || epilogue == 0, "the increment form (accumulate = 2) takes no bias / ReLU epilogue");
    if (n_rows == 0 || feat == 0) return DGLL_OK;
    DGLL_REQUIRE(rowptr && X && Y, "NULL rowptr/X/Y");
    DGLL_REQUIRE(ldx >= feat && ldy >= feat, "leading dimension smaller than feat");
    DGLL_REQUIRE(ldx < ((int64_t)1 << 31) && n_cols < ((int64_t)1 << 31), "leading dimension / column count must fit 31 bits");
    DGLL_REQUIRE(x_dtype == DGLL_F32 || x_dtype == DGLL_BF16, "x_dtype");
    DGLL_REQUIRE(y_dtype == DGLL_F32 || y_dtype == DGLL_BF16, "y_dtype");
    DGLL_REQUIRE(reduce == DGLL_REDUCE_SUM || reduce == DGLL_REDUCE_MEAN, "reduce");
    DGLL_REQUIRE((epilogue & ~(DGLL_EPI_BIAS | DGLL_EPI_RELU)) == 0, "epilogue");
    DGLL_REQUIRE(!(epilogue & DGLL_EPI_BIAS) || bias, "DGLL_EPI_BIAS needs bias");
    if (x_dtype == DGLL_F32 && y_dtype == DGLL_BF16) {
        set_error("fp32 input with bf16 output is not supported");
        return DGLL_ERR_UNSUPPORTED;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);

    SpmmArgs a{};
    a.rowptr = rowptr; a.col = col; a.val = val; a.X = X; a.Y = Y;
    a.ldx = ldx; a.ldy = ldy; a.n_rows = n_rows; a.feat = feat; a.reduce = reduce; a.epilogue = epilogue; a.bias = bias;
    a.ws_ld = ws_ld_for(feat);
    a.rows_per_wave = 1;
    a.flags = g_tune_flags;
    a.row_scale = row_scale;
    a.accumulate = accumulate;
    a.gate = gate; a.ldg = ldg;
    if (plan) {
        DGLL_REQUIRE(plan->n_rows == n_rows, "plan was built for a different CSR");
        a.threshold = plan->threshold;
        a.n_chunks = plan->n_chunks;
        a.chunk_begin = plan->d_chunk_begin;
        a.chunk_end = plan->d_chunk_end;
        if (plan->n_chunks > 0) {
            const size_t need = dgll_hip_csr_plan_workspace_bytes(plan, feat);
            if (!workspace || workspace_bytes < need) {
                set_error("workspace too small for the plan's long-row partials");
                return DGLL_ERR_WORKSPACE;
            }
            a.ws = static_cast<float*>(workspace);
        }
        // several rows per wavefront amortise wave start-up and smooth the tail: aim for ~96 KiB of gathered bytes per
        // wave (measured on the products shapes: 4 rows/wave is 8-18 % faster than 1; tools/spmm_tune.py)
        const double row_bytes = (double)plan->nnz / (double)std::max<int64_t>(n_rows, 1) * feat *
                                 (x_dtype == DGLL_BF16 ? 2.0 : 4.0);
        int rpw = row_bytes > 0 ? (int)(98304.0 / row_bytes) : 8;
        a.rows_per_wave = std::min(std::max(rpw, 1), 8);
        if (g_tune_rows_per_wave > 0) a.rows_per_wave = g_tune_rows_per_wave;
    }
    if (only_long && (!plan || plan->n_long == 0)) return DGLL_OK;
    const int64_t waves = (n_rows + a.rows_per_wave - 1) / a.rows_per_wave;
    const int64_t row_blocks = only_long ? 0 : (waves + kWavesPerBlock - 1) / kWavesPerBlock;
    int64_t chunk_blocks = (a.n_chunks + kWavesPerBlock - 1) / kWavesPerBlock;
    chunk_blocks = (chunk_blocks + kXcds - 1) / kXcds * kXcds;  // keep (block % 8) == XCD for the row blocks
    DGLL_REQUIRE(row_blocks + chunk_blocks < (int64_t)0x7fffffff, "grid too large");
    a.row_blocks = (uint32_t)row_blocks;
    a.chunk_blocks = (uint32_t)chunk_blocks;

    const int esz = x_dtype == DGLL_BF16 ? 2 : 4;
    const int ysz = y_dtype == DGLL_BF16 ? 2 : 4;
    const int epv = 16 / esz;
    const bool fast = aligned16(X) && aligned16(Y) && (ldx * esz) % 16 == 0 && (ldy * ysz) % 16 == 0 &&
                      ((int64_t)epv * ysz) % 16 == 0 && (!gate || (aligned16(gate) && (ldg * ysz) % 16 == 0));
    if (fast) a.flags |= 2;
    hipError_t err;
    bool rowslot = false;
    if (fast) {
        const int vecs = (feat + epv - 1) / epv;
        int lpr = 4;
        while (lpr < 64 && lpr < vecs) lpr <<= 1;
        // row-per-slot kernel: SHORT rows (<= 24 edges on average) of at most 16 vectors (F <= 128 bf16).  Measured on MI355X
        // (tools/rowslot_ab.py): 6.6 edges per row -- the halo halves of an 8-way partition -- F = 47: 1.00 -> 0.47 ms, F = 100 /
        // 128: -24 %; F = 256 (two slots): +6 %, and at 51 edges per row the wave-per-row kernel wins everywhere (+11 .. +32 %:
        // a slot walks its row U edges at a time, a whole wavefront 64).
        const double avg_len = plan ? (double)plan->nnz / (double)std::max<int64_t>(n_rows, 1) : 1e9;
        // Round 3: at F = 256 (two slots) it wins once rows are VERY short -- 3.2 edges per row, the transposed halo half of an 8-way
        // partition (678 k rows, 2.1 M edges): 0.49 -> 0.39 ms (tools/scaling_trace.py); hence <= 4.5 edges for 32 lanes per row.
        rowslot = !only_long && lpr <= 32 && g_tune_rowslot != 1 &&
                  (g_tune_rowslot == 2 || (lpr <= 16 && avg_len <= 24.0) || (lpr == 32 && avg_len <= 4.5));
        if (rowslot) {
            const int slots = kWave / lpr;
            a.rows_per_wave = std::max(a.rows_per_wave, 1);
            a.rows_per_wave = (a.rows_per_wave + slots - 1) / slots * slots;
            if (g_tune_rows_per_wave > 0) a.rows_per_wave = (g_tune_rows_per_wave + slots - 1) / slots * slots;
        }
    }
    if (rowslot) {   // the grid depends on rows_per_wave: recompute
        const int64_t waves2 = (n_rows + a.rows_per_wave - 1) / a.rows_per_wave;
        const int64_t row_blocks2 = (waves2 + kWavesPerBlock - 1) / kWavesPerBlock;
        DGLL_REQUIRE(row_blocks2 + chunk_blocks < (int64_t)0x7fffffff, "grid too large");
        a.row_blocks = (uint32_t)row_blocks2;
        const int vecs = (feat + epv - 1) / epv;
        int lpr = 4;
        while (lpr < 64 && lpr < vecs) lpr <<= 1;
        dim3 grid((uint32_t)(row_blocks2 + chunk_blocks), (uint32_t)((vecs + lpr - 1) / lpr));
        if (x_dtype == DGLL_F32) err = launch_rowslot_lpr<float, float, 4>(a, lpr, grid, s);
        else if (y_dtype == DGLL_BF16) err = launch_rowslot_lpr<bf16_t, bf16_t, 8>(a, lpr, grid, s);
        else err = launch_rowslot_lpr<bf16_t, float, 8>(a, lpr, grid, s);
    } else if (fast && !only_long && plan && plan->d_flat_row0 && g_tune_flat != 1 && g_tune_unroll == 4 && !(a.flags & 1) &&
               [&]() { const int v = (feat + epv - 1) / epv; return v > 8 && v <= 32; }() &&
               (g_tune_flat == 2 || (x_dtype == DGLL_F32 && (double)plan->nnz / (double)std::max<int64_t>(n_rows, 1) >= 8.0))) {
        // The flattened edge-stream kernel.  Measured on the products-sized bench graph (tools/flat_ab.py, E = 256, interleaved):
        // fp32 F = 100 forward 4.23 -> 3.93 ms (-7 %): chosen for fp32 rows of 16 / 32 lanes; bf16 F = 256 forward 4.35 -> 4.26-4.35,
        // F = 100 / 128 bf16 +3-4 % slower, the weighted + gated + accumulating transposed pass 5.49 -> 5.85 (73-87 registers: 5-6
        // wavefronts per SIMD against 8): not chosen for bf16 (dgll_hip_debug_tune(13, 2) forces it, (13, 1) disables it).
        const int vecs = (feat + epv - 1) / epv;
        const int lpr = vecs <= 16 ? 16 : 32;
        a.flat_row0 = plan->d_flat_row0;
        a.n_flat = plan->n_flat;
        const int64_t flat_blocks = (plan->n_flat + kWavesPerBlock - 1) / kWavesPerBlock;
        DGLL_REQUIRE(flat_blocks + chunk_blocks < (int64_t)0x7fffffff, "grid too large");
        a.row_blocks = (uint32_t)flat_blocks;
        dim3 grid((uint32_t)(flat_blocks + chunk_blocks), (uint32_t)((vecs + lpr - 1) / lpr));
        if (x_dtype == DGLL_F32) err = launch_flat_lpr<float, float, 4>(a, lpr, grid, s);
        else if (y_dtype == DGLL_BF16) err = launch_flat_lpr<bf16_t, bf16_t, 8>(a, lpr, grid, s);
        else err = launch_flat_lpr<bf16_t, float, 8>(a, lpr, grid, s);
    } else if (fast) {
        const int vecs = (feat + epv - 1) / epv;
        int lpr = 4;
        while (lpr < 64 && lpr < vecs) lpr <<= 1;
        dim3 grid((uint32_t)(row_blocks + chunk_blocks), (uint32_t)((vecs + lpr - 1) / lpr));
        if (x_dtype == DGLL_F32) err = launch_lpr<float, float, 4>(a, lpr, grid, s);
        else if (y_dtype == DGLL_BF16) err = launch_lpr<bf16_t, bf16_t, 8>(a, lpr, grid, s);
        else err = launch_lpr<bf16_t, float, 8>(a, lpr, grid, s);
    } else {
        dim3 grid((uint32_t)(row_blocks + chunk_blocks), (uint32_t)((feat + 63) / 64));
        if (x_dtype == DGLL_F32) err = launch_variant<float, float, 1, 64>(a, grid, s);
        else if (y_dtype == DGLL_BF16) err = launch_variant<bf16_t, bf16_t, 1, 64>(a, grid, s);
        else err = launch_variant<bf16_t, float, 1, 64>(a, grid, s);
    }
    if (err != hipSuccess) return hip_fail(err, "spmm_csr_kernel launch");

    if (plan && plan->n_long > 0) {
        dim3 grid((uint32_t)((plan->n_long + kWavesPerBlock - 1) / kWavesPerBlock));
        if (y_dtype == DGLL_F32)
            hipLaunchKernelGGL(spmm_long_finalize_kernel<float>, grid, dim3(kBlock), 0, s, a, plan->d_long_row,
                               plan->d_long_chunk0, plan->n_long);
        else
            hipLaunchKernelGGL(spmm_long_finalize_kernel<bf16_t>, grid, dim3(kBlock), 0, s, a, plan->d_long_row,
                               plan->d_long_chunk0, plan->n_long);
        err = hipGetLastError();
        if (err != hipSuccess) return hip_fail(err, "spmm_long_finalize_kernel launch");
    }
    return DGLL_OK;
}
