// gat_kernel.hpp -- the three gather passes of the fused multi-head GAT layer for gfx950, second generation.
// Included by gat_fwd.hip / gat_bwd_rows.hip / gat_bwd_cols.hip (one translation unit per pass: they compile in parallel).
//
// Reference semantics (under /root/reference/dgll/nn/Convolution/):
//   sparseGatConv.forward  gatconv.py:111-148   e_ij = exp(-leakyrelu(a1.h_i + a2.h_j)); out_i = sum_j e_ij h_j / sum_j e_ij; elu
//   SpecialSpmmFunction    gatconv.py:60-81     the two backward products (here: two more gather passes, nothing per edge stored)
//   SpGAT                  gatconv.py:174-199   `nheads` independent heads -- all of them in ONE launch here
// (the max-subtracted softmax of the dense-adjacency gatConv, gatconv.py:30-54, and attention dropout stay on edge.hip's
// first-generation kernels.)
//
// Where the time of these passes goes (MI355X, 8 heads x 32 bf16, products-sized graph; tools/gat_ab.py): like the SpMM they are
// bound by cache-line fills per edge -- 4 lines for the 512-byte feature row, plus ONE line for every separate per-node array
// that is gathered per edge (the neighbour's scores T[j, :]; in the transposed pass S[i, :] and DD[i, :]).  Hence:
//   * every per-(edge, head) scalar -- the weight w_ij = exp(..) and its derivative factor -- is computed ONCE, by the lane
//     that owns the edge in the coalesced 64-edge index batch, from one vector load of the neighbour's score row, and handed
//     to the gathering lanes through a wave-private LDS record array (first generation: each of the 4 lanes of a head issued
//     its own 4-byte gather and its own exponential: 32 scalar gathers + 32 v_exp per lane and batch instead of 2 + 8);
//   * the backward passes do not reduce <DN_i, h_j> across the lanes of a head per EDGE (two ds_bpermute each):
//         ds_i = sum_j c_ij (dot_ij + dd_i) = head_sum( sum_j c_ij * partial_dot_ij ) + dd_i * sum_j c_ij
//     a lane accumulates its partial dot products weighted by c_ij; the cross-lane sum happens once per ROW;
//   * {s_i, dd_i} are stored side by side (`sd_out`) so that the transposed pass pays one line fill for both, and any of the
//     gathered score arrays may live in the PADDING of the feature rows themselves (`tstride`: a 47-class output row is 94 of
//     128 bytes) -- then the score costs no line fill at all;
//   * the rounds are branch- and mask-free (records of a batch's idle tail are zero, their columns repeat the last valid one)
//     and the next row's index batch is requested before the current row's gathers.
// LDS use: wave-private, written and read by the same wavefront (in-order DS queue: no barrier, no s_waitcnt between rows).
#pragma once
#include <algorithm>
#include <type_traits>

#include "edge_args.hpp"

namespace dgll {

namespace {

constexpr int kRecStride = kWave + 1;   // one pad entry per head row: the 8 heads a half-wave reads land in different banks

// NH consecutive per-head scalars of one node (row stride `stride` floats).  Blocks of 4 or 8 heads are read as float4s (the
// host guarantees whole, 16-byte aligned blocks); 1 or 2 heads per wavefront as guarded scalars.
template <int NH>
__device__ __forceinline__ void load_heads(const float* __restrict__ base, int64_t node, int stride, int heads, int h0, float (&out)[NH]) {
    const float* p = base + node * stride + h0;
    if constexpr (NH % 4 == 0) {
#pragma unroll
        for (int q = 0; q < NH / 4; ++q) {
            const float4 t = reinterpret_cast<const float4*>(p)[q];
            out[4 * q + 0] = t.x; out[4 * q + 1] = t.y; out[4 * q + 2] = t.z; out[4 * q + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < NH; ++k) out[k] = (h0 + k < heads) ? p[k] : 0.0f;
    }
}

template <int NH>
__device__ __forceinline__ void load_heads_uniform(const float* __restrict__ base, int64_t row, int heads, int h0, float (&out)[NH]) {
#pragma unroll
    for (int k = 0; k < NH; ++k) out[k] = base[row * heads + (h0 + k < heads ? h0 + k : heads - 1)];
}

template <int LPH>
__device__ __forceinline__ float head_sum_c(float v) {
#pragma unroll
    for (int off = 1; off < LPH; off <<= 1) v += __shfl_xor(v, off);
    return v;
}

// The same sum on the DPP path (no LDS crossbar trip) for the per-EDGE use of the row-score form: quad_perm [1,0,3,2], [2,3,0,1],
// then row_half_mirror (lane i <-> 7 - i: the other quad of 8) and row_mirror (15 - i: the other half of 16) -- every lane of the
// group ends up with the group's sum.  Groups of 32 / 64 lanes finish through ds_bpermute.
template <int CTRL>
__device__ __forceinline__ float dpp_take(float v) {
    // full row / bank masks and a permutation inside the row: every lane receives a value, `old` is never seen.  The form without an
    // `old` operand lets the compiler fold the move into the add that follows (v_add_f32_dpp): with old = 0 it kept v_mov 0 + v_mov_dpp + v_add.
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, false));
}
template <int LPH>
__device__ __forceinline__ float head_sum_dpp(float v) {
    if constexpr (LPH >= 2) v += dpp_take<0xB1>(v);
    if constexpr (LPH >= 4) v += dpp_take<0x4E>(v);
    if constexpr (LPH >= 8) v += dpp_take<0x141>(v);
    if constexpr (LPH >= 16) v += dpp_take<0x140>(v);
    if constexpr (LPH >= 32) v += __shfl_xor(v, 16);
    if constexpr (LPH >= 64) v += __shfl_xor(v, 32);
    return v;
}

}  // namespace

// KIND 0: forward.  KIND 1: backward over the rows of A (DN, DD, grad_S).  KIND 2: backward over the rows of A^T (grad_H, grad_T).
// See edge.hip for the argument roles of each pass (they are unchanged).  Records (one per edge and head, wave-private LDS):
//   KIND 0: w_ij     KIND 1: w_ij carrying the sign of z_ij (c_ij = w_ij * sign * lrelu'(z_ij) needs only that bit)     KIND 2: { w_ij, c_ij } and dd_i * c_ij
// INROW (one head, scores in the padding of the gathered rows themselves): the lane group's first idle lane reads the 16 bytes
// behind the row's last column WITH the row -- t_j (or {s_i, dd_i} in the transposed pass) arrives with the gather, is broadcast
// inside the lane group, and the record phase (a dependent load per batch, the LDS hand-over) disappears: for 8-lane rows the
// passes are bound by exactly that per-row latency chain.
// TROW (forward and rows passes): the gathered-side score t_j = h_j . a2 is FORMED from the gathered row itself, as the reference does
// (gatconv.py:122-125 builds the logit from the gathered rows): every lane holds its EPV columns of a2 (bf16 storage: as packed pairs
// for v_dot2), takes the partial dot product with the row it has just gathered and sums over the head's lanes on the DPP path; each
// lane then evaluates w_ij itself.  No score row is fetched (the FIFTH cache line per edge of the 8 x 32 bf16 layer), no
// record phase, no LDS hand-over -- for 4 x more exponentials per edge, on a VALU that was a third busy.
template <typename XT, typename YT, int EPV, int LPR, int NH, int U, int KIND, bool INROW = false, bool TROW = false>
// (the narrow in-row rows pass is held at 6 wavefronts per SIMD: the exact dd_i's extra sum took it from 79 to 86 VGPRs and from 6 to
// 5 wavefronts, 2.56 -> 2.88 ms; bounded it fits 77 registers without scratch.  The 8-head form needs 12 bytes of scratch at that bound
// and measured no faster: it runs at 83 registers / 5 wavefronts, 6.02 -> 6.17 ms for the exactness.)
__global__ __launch_bounds__(kBlock, ((KIND == 1 || KIND == 3) && INROW && sizeof(XT) == 2) ? 6 : 1) void gat2_kernel(const EdgeArgs a) {
    typedef VecIO<XT, EPV> IO;
    typedef typename std::conditional<KIND == 2, float2, float>::type rec_t;   // KIND 2: {w_ij, c_ij}, plus dd_i * c_ij in rec1;
                                                                               // KIND 1: w_ij with the sign of z_ij (c_ij = w_ij * f, f = sign or sign * alpha)
    constexpr int SLOTS = kWave / LPR;
    constexpr bool ROWS = KIND == 1 || KIND == 3;                   // the rows pass: KIND 3 = its exact-dd form ALONE (no code of the
    constexpr bool EXACT = KIND == 3;                               // stored-output form: 75 instead of 83 VGPRs, 6 wavefronts per SIMD)
    constexpr int LPH = LPR / NH;                                  // lanes per head
    constexpr bool BF = sizeof(XT) == 2;
    __shared__ rec_t rec_all[kWavesPerBlock][NH * kRecStride];
    __shared__ float rec1_all[KIND == 2 ? kWavesPerBlock : 1][KIND == 2 ? NH * kRecStride : 1];
    __shared__ float attn_all[KIND == 2 ? 2 : 1][KIND == 2 ? LPR * EPV : 1];   // a1 | a2 of this column block (score-gradient epilogue)
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    rec_t* __restrict__ rec = rec_all[wave];
    const int sub = lane % LPR, slot = lane / LPR;
    const int hk = sub / LPH, hs = sub % LPH;                      // this lane's head among the wave's NH heads, its vector in the head
    const int h0 = (int)blockIdx.y * NH;                           // first head of this column block
    const bool col_ok = h0 + hk < a.heads && hs < a.vph;           // a head may use fewer vectors than it has lanes (fo = 48: 6 of 8)
    const int head = col_ok ? h0 + hk : 0;
    const int c0 = col_ok ? head * a.fo + hs * EPV : 0;
    // INROW: the first idle lane of the group fetches the score slot behind the row's last column
    const XT* hcol = static_cast<const XT*>(a.H) + (col_ok ? c0 : ((INROW && hs == a.vph) ? a.feat : 0));
    const int tlane = (lane & ~(LPR - 1)) + a.vph;
    const uint32_t ld32 = (uint32_t)a.ldh;
    const uint32_t lane_off = (uint32_t)((col_ok ? c0 : ((INROW && hs == a.vph) ? a.feat : 0)) * (int)sizeof(XT));
    const rec_t* __restrict__ my_rec = rec + hk * kRecStride;
    float* __restrict__ rec1 = rec1_all[KIND == 2 ? wave : 0];
    const float* __restrict__ my_rec1 = rec1 + hk * kRecStride;

    if constexpr (KIND == 2) {
        if (a.attn1) {      // staged once per workgroup: the epilogue of every row reads them (from global memory it cost 0.6 ms per pass)
            for (int i = threadIdx.x; i < LPR * EPV; i += kBlock) {
                const int hh = h0 + i / (LPH * EPV), cc = i % (LPH * EPV);
                const bool ok = hh < a.heads && cc < a.fo;
                attn_all[0][i] = ok ? a.attn1[hh * a.fo + cc] : 0.0f;
                attn_all[1][i] = ok ? a.attn2[hh * a.fo + cc] : 0.0f;
            }
            __syncthreads();
        }
    }
    // (Round 4, measured and dropped: requesting the NEXT item's first score rows one item ahead, behind the current row's gathers --
    // forward 5.58 -> 5.61 ms, rows pass 6.04 -> 6.15; in the transposed pass the held registers cost a wavefront of occupancy.)
    uint32_t a2p[4] = {0u, 0u, 0u, 0u};                            // TROW: this lane's columns of a2 (idle lanes: zeros -> partial dot 0)
    float a2f[EPV];
#pragma unroll
    for (int i = 0; i < EPV; ++i) a2f[i] = 0.0f;
    if constexpr (TROW) {
        if (col_ok) {
#pragma unroll
            for (int i = 0; i < EPV; ++i) a2f[i] = a.attn2[c0 + i];
        }
        if constexpr (BF) {
#pragma unroll
            for (int q = 0; q < 4; ++q) a2p[q] = pack_bf16x2(a2f[2 * q], a2f[2 * q + 1]);
        }
    }
    WorkItem it = resolve_item(a, wave, 0);
    int col_first = 0;                                             // the item's first index batch (lane = edge), prefetched
    if (it.valid && it.b + lane < it.e) col_first = __builtin_nontemporal_load(a.col + it.b + lane);
    // TROW: s_i of this lane's own head, requested ONE ITEM AHEAD like the index batch (read at the top of its row the load's latency
    // stood in front of every row's gathers: the one wait the plain SpMM of the same shape does not have)
    float s_first = 0.0f;
    const int head_s = h0 + hk < a.heads ? h0 + hk : 0;          // the lane's head even where it holds none of its columns (fo < lanes x EPV)
    if constexpr (TROW) { if (it.valid) s_first = a.S[it.row * a.heads + head_s]; }
  for (int r = 0; !it.done; ++r) {
    // the NEXT item's bounds and first index batch are requested before this item's gathers: a row's dependent chain
    // (row pointers -> indices -> score rows -> gathers) then starts at the score rows
    WorkItem nx;
    nx.done = true; nx.valid = false; nx.first = true; nx.row = nx.b = nx.e = 0; nx.chunk = -1;
    if (r + 1 < a.rows_per_wave) nx = resolve_item(a, wave, r + 1);
    int col_next_item = 0;
    if (nx.valid && nx.b + lane < nx.e) col_next_item = __builtin_nontemporal_load(a.col + nx.b + lane);
    float s_next = 0.0f;
    if constexpr (TROW) { if (nx.valid) s_next = a.S[nx.row * a.heads + head_s]; }
   if (it.valid) {
    const int64_t row = it.row, b = it.b, e = it.e;

    // ---- wave-uniform per-row scalars of the wave's heads
    float su[NH];
    load_heads_uniform<NH>(a.S, row, a.heads, h0, su);
    const float s_mine = s_first;                                  // TROW: s_i of this lane's own head

    // ---- per-row prologue of the backward passes
    float dn[EPV];                     // KIND 1: DN_i (this lane's columns)
    uint32_t rp[4] = {0u, 0u, 0u, 0u}; // bf16: DN_i (KIND 1) or h_j (KIND 2) as packed pairs for v_dot2
    float hj[EPV];                     // KIND 2 fp32: h_j
    float dd = 0.0f;
    if constexpr (ROWS) {
        const float inv_den = 1.0f / a.DEN[row * a.heads + head];
        float g[EPV], o[EPV];
        IO::unpack(col_ok ? IO::load_nt(static_cast<const XT*>(a.G) + row * a.ldg + c0) : IO::zero(), g);   // row-side operands: streamed
        IO::unpack(col_ok ? IO::load_nt(static_cast<const XT*>(a.O) + row * a.ldo + c0) : IO::zero(), o);
        // dd_i = -DN_i . hp_i  (hp_i = the row's pre-activation output).  EXACT form (a.exact_dd, every single-launch use): hp_i is
        // never reconstructed -- DN_i . hp_i = sum_j w_ij (DN_i . h_j) / den_i, and the dot products are formed below anyway: the
        // lanes accumulate  sw = sum_j w_ij * partial_dot_ij  next to  sa = sum_j c_ij * partial_dot_ij  and dd_i falls out in the row
        // epilogue (long rows: the chunk partials carry (sa, sb, sw) and the finalize kernel combines them -- ds_i is bilinear in
        // them).  The same bf16 h_j enter dd_i and the transposed pass, so the cancellation in ds_i is exact to fp32 rounding.
        // LEGACY form (launches over column halves of A; the caller did not declare this launch the only one): hp_i from the stored output
        // row (bf16: rounded to 8 bits; ELU inverted with a logarithm), dd_i known before the gathers.
        constexpr bool legacy = !EXACT;
        float part = 0.0f;
#pragma unroll
        for (int i = 0; i < EPV; ++i) {
            float dhp = g[i], hp = o[i];
            if (a.apply_elu && o[i] <= 0.0f) {  // out = expm1(hp): elu'(hp) = out + 1, hp = log1p(out)
                const float op1 = o[i] + 1.0f;    // saturated ELU (out == -1): gradient 0, and 0 * log(0) must stay 0
                dhp = g[i] * op1;
                // bf16 storage: the hardware logarithm (absolute error ~1e-7 on a value rounded to 8 bits anyway); fp32: log1pf
                if (legacy) hp = op1 > 0.0f ? (BF ? __logf(op1) : log1pf(o[i])) : 0.0f;
            }
            dn[i] = dhp * inv_den;
            // bf16: the dot products below and the transposed pass see DN in storage precision; whatever is formed from DN here must
            // use the SAME rounded values, or the cancellation in ds_i = sum_j c_ij (DN_i . h_j + dd_i) leaves DN's rounding standing
            if constexpr (BF) dn[i] = bf16_to_f32(f32_to_bf16(dn[i]));
            part = fmaf(dn[i], hp, part);
        }
        if (legacy) dd = -head_sum_c<LPH>(part);
        if (slot == 0 && col_ok && it.first && a.accumulate != 1 && a.exact_dd < 3) {   // per-row outputs: written once (row itself / first chunk, first launch)
            VecIO<XT, EPV>::store_nt(static_cast<XT*>(a.Y) + row * a.ldy + c0, dn);
            if (hs == 0 && legacy) {
                if (a.out_b) a.out_b[row * a.heads + head] = dd;
                if (a.sd_out) {   // {s_i, dd_i} side by side where the transposed pass gathers them with ONE line fill per edge
                    a.sd_out[row * a.sd_stride + head] = a.S[row * a.heads + head];
                    a.sd_out[row * a.sd_stride + a.heads + head] = dd;
                }
            }
        }
        if constexpr (BF) {  // the transposed pass re-reads DN in storage precision: use the same rounded values here
#pragma unroll
            for (int q = 0; q < 4; ++q) rp[q] = pack_bf16x2(dn[2 * q], dn[2 * q + 1]);
        }
    }
    float gs_row = 0.0f;               // KIND 2: grad_S of this row's head (score-gradient epilogue), requested before the gather
    if constexpr (KIND == 2) {
        if (a.attn1 && it.chunk < 0) gs_row = a.gs_rows[row * a.heads + head];
        const typename IO::raw_t raw = col_ok ? IO::load_nt(static_cast<const XT*>(a.G) + row * a.ldg + c0) : IO::zero();
        if constexpr (BF) { rp[0] = raw.x; rp[1] = raw.y; rp[2] = raw.z; rp[3] = raw.w; }
        else IO::unpack(raw, hj);
    }

    float acc[EPV];
#pragma unroll
    for (int i = 0; i < EPV; ++i) acc[i] = 0.0f;
    float sa = 0.0f, sb = 0.0f;        // KIND 0: sb = denominator.  KIND 1/2: sa = sum c * partial dot, sb = sum c (or dd * c)
    float sw = 0.0f;                   // KIND 1: sum w * partial dot (the exact dd_i)

    int col_cur = col_first;
    for (int64_t k0 = b; k0 < e; k0 += kWave) {
        const int64_t left = e - k0;
        const int nb = left < kWave ? (int)left : kWave;
        const bool live = lane < nb;
        int col_nxt = 0;                                        // next index batch of this row
        {
            const int64_t kn = k0 + kWave + lane;
            if (kn < e) col_nxt = __builtin_nontemporal_load(a.col + kn);
        }
        // -- record phase: lane = edge.  (Measured alternative for 8 heads, lane = (edge, half) so that one load instruction
        // covers 32 edges with two adjacent lanes per 32-byte score row: forward unchanged, transposed pass 6.4 -> 7.8 ms.)
        if constexpr (!INROW && !TROW) {
            float tv[NH], dv[NH];
            load_heads<NH>(a.T, col_cur, a.tstride, a.heads, h0, tv);        // idle lanes hold column 0: a valid row
            if constexpr (KIND == 2) load_heads<NH>(a.DD, col_cur, a.tstride, a.heads, h0, dv);
#pragma unroll
            for (int k = 0; k < NH; ++k) {
                const float z = su[k] + tv[k];
                float w = __expf(a.sign * lrelu(z, a.alpha));
                w = (live && h0 + k < a.heads) ? w : 0.0f;
                const float cc = w * a.sign * (z > 0.0f ? 1.0f : a.alpha);
                if constexpr (KIND == 0) rec[k * kRecStride + lane] = w;
                if constexpr (ROWS) rec[k * kRecStride + lane] = z > 0.0f ? w : -w;
                if constexpr (KIND == 2) { rec[k * kRecStride + lane] = make_float2(w, cc); rec1[k * kRecStride + lane] = dv[k] * cc; }
            }
        }
        // idle lanes of the last batch repeat its last valid column (same cache lines as a live request; their records are zero)
        const int last_col = __builtin_amdgcn_readlane(col_cur, nb - 1);
        const int gcol = live ? col_cur : last_col;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // -- gather rounds: lane = (slot, columns)
        for (int j = 0; j < nb; j += SLOTS * U) {
            typename IO::raw_t v[U];
            rec_t rr[U];
            float r1[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = j + u * SLOTS + slot;
                const int cj = __shfl(gcol, idx);
                if constexpr (TROW) {
                    // 32-bit byte offset from the uniform base: one full-rate v_mad_u32_u24 and the load's scalar-base form instead of
                    // v_mad_u64_u32 + a 64-bit add per gather (forward 5.08 -> 4.95 ms; the entry point admits matrices under 4 GB and
                    // 2^24 rows only).  The plain SpMM gains nothing from the same change: it waits for memory, this pass for its VALU.
                    const uint32_t off = __umul24((uint32_t)cj, ld32 * (uint32_t)sizeof(XT)) + lane_off;
                    v[u] = *reinterpret_cast<const typename IO::raw_t*>(static_cast<const char*>(a.H) + off);
                } else {
                    v[u] = IO::load(hcol + (uint64_t)(uint32_t)cj * ld32);
                }
                if constexpr (!INROW && !TROW) {
                    rr[u] = my_rec[idx];
                    if constexpr (KIND == 2) r1[u] = my_rec1[idx];
                }
            }
            if constexpr (TROW) {
                static_assert(KIND != 2, "the transposed pass gathers DN rows: their scores cannot be formed from them");
                // w = exp(sign * lrelu(z)) = exp2(z * c_neg + max(z, 0) * (c_pos - c_neg)),  c_pos = sign * log2(e), c_neg = alpha * c_pos:
                // mul + max + fma + v_exp per (edge, head) instead of mul, compare, select, mul, mul, v_exp
                const float c_neg = a.sign * a.alpha * 1.44269504088896341f, c_dif = a.sign * 1.44269504088896341f - c_neg;
                auto scores = [&](auto masked) __attribute__((always_inline)) {
                    float z[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        float pt = 0.0f;
                        if constexpr (BF) {
                            const uint32_t hv[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                pt = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a2p[q]), __builtin_bit_cast(bf16x2_t, hv[q]), pt, false);
                        } else {
                            float f[EPV];
                            IO::unpack(v[u], f);
#pragma unroll
                            for (int i = 0; i < EPV; ++i) pt = fmaf(f[i], a2f[i], pt);
                        }
                        z[u] = s_mine + head_sum_dpp<LPH>(pt);
                    }
                    if constexpr (LPH == 4 && U == 4) {
                        // the head's four lanes hold the same four z: lane hs evaluates edge hs and hands w to the quad (one exponential
                        // and one mask per lane and group instead of four; v_exp_f32 is a quarter-rate instruction)
                        const float zs = hs == 0 ? z[0] : (hs == 1 ? z[1] : (hs == 2 ? z[2] : z[3]));
                        float ws = __builtin_amdgcn_exp2f(fmaf(fmaxf(zs, 0.0f), c_dif, zs * c_neg));
                        if constexpr (decltype(masked)::value) ws = (j + hs * SLOTS + slot < nb) ? ws : 0.0f;
                        const float w4[4] = {dpp_take<0x00>(ws), dpp_take<0x55>(ws), dpp_take<0xAA>(ws), dpp_take<0xFF>(ws)};
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            if constexpr (KIND == 0) rr[u] = w4[u];
                            if constexpr (ROWS) rr[u] = z[u] > 0.0f ? w4[u] : -w4[u];
                        }
                    } else {
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            float w = __builtin_amdgcn_exp2f(fmaf(fmaxf(z[u], 0.0f), c_dif, z[u] * c_neg));
                            if constexpr (decltype(masked)::value) w = (j + u * SLOTS + slot < nb) ? w : 0.0f;
                            if constexpr (KIND == 0) rr[u] = w;
                            if constexpr (ROWS) rr[u] = z[u] > 0.0f ? w : -w;
                        }
                    }
                };
                // every slot of the group holds a live edge (wave-uniform; all groups of a row but its last): no per-edge mask.  (The
                // empty asm keeps the two forms apart: merged into selects they cost every group two per edge.)
                if (j + SLOTS * U <= nb) scores(std::false_type{});
                else { asm volatile(""); scores(std::true_type{}); }
            }
            if constexpr (INROW && KIND == 0 && U == 4 && LPR >= 4) {
                // the forward as in the row-score form: w = exp2(z c_neg + max(z, 0) (c_pos - c_neg)), and ONE exponential per lane and group
                // of four edges -- every lane of the row's group holds the same four z (t_j broadcast from the score lane): lane q of each
                // quad evaluates edge q and hands w round the quad (one-head output layer: 2.20 -> 2.07 ms).  The two backward passes keep
                // the per-edge form below: waiting for all four rows before any arithmetic cost them 8-14 % (2.52 -> 2.87, 2.5 -> 2.7 ms).
                const float k_neg = a.sign * a.alpha * 1.44269504088896341f, k_dif = a.sign * 1.44269504088896341f - k_neg;
                float z[U];
#pragma unroll
                for (int u = 0; u < U; ++u) z[u] = su[0] + __shfl(__uint_as_float(v[u].x), tlane);
                const int q = lane & 3;
                const float zs = q == 0 ? z[0] : (q == 1 ? z[1] : (q == 2 ? z[2] : z[3]));
                float ws = __builtin_amdgcn_exp2f(fmaf(fmaxf(zs, 0.0f), k_dif, zs * k_neg));
                ws = (j + q * SLOTS + slot < nb) ? ws : 0.0f;
                rr[0] = dpp_take<0x00>(ws); rr[1] = dpp_take<0x55>(ws); rr[2] = dpp_take<0xAA>(ws); rr[3] = dpp_take<0xFF>(ws);
            } else if constexpr (INROW) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const float tx = __shfl(__uint_as_float(v[u].x), tlane);       // t_j (KIND 0 / 1) or s_i (KIND 2)
                    const float z = su[0] + tx;
                    float w = __expf(a.sign * lrelu(z, a.alpha));
                    w = (j + u * SLOTS + slot < nb) ? w : 0.0f;
                    const float cc = w * a.sign * (z > 0.0f ? 1.0f : a.alpha);
                    if constexpr (KIND == 0) rr[u] = w;
                    if constexpr (ROWS) rr[u] = z > 0.0f ? w : -w;
                    if constexpr (KIND == 2) {
                        rr[u] = make_float2(w, cc);
                        r1[u] = __shfl(__uint_as_float(v[u].y), tlane) * cc;    // dd_i sits next to s_i
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if constexpr (KIND == 0) {
                    sb += rr[u];
                    float f[EPV];
                    IO::unpack(v[u], f);
#pragma unroll
                    for (int i = 0; i < EPV; ++i) acc[i] = fmaf(rr[u], f[i], acc[i]);
                } else {
                    float dot = 0.0f;
                    if constexpr (BF) {   // <DN_i, h_j> on the packed pairs (v_dot2c_f32_bf16, fp32 accumulate)
                        const uint32_t hv[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            dot = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, rp[q]), __builtin_bit_cast(bf16x2_t, hv[q]), dot, false);
                    }
                    if constexpr (ROWS) {
                        if constexpr (!BF) {
                            float f[EPV];
                            IO::unpack(v[u], f);
#pragma unroll
                            for (int i = 0; i < EPV; ++i) dot = fmaf(dn[i], f[i], dot);
                        }
                        if constexpr (INROW) dot = col_ok ? dot : 0.0f;   // the score lane holds score bits, not features: 0 x NaN
                        // one record per (edge, head): |r| = w_ij, its sign = the sign of z_ij; c_ij = w_ij * f
                        const float wv = fabsf(rr[u]);
                        const float fz = rr[u] > 0.0f ? a.sign : a.sign * a.alpha;
                        const float dw = dot * wv;
                        sw += dw;
                        sa = fmaf(fz, dw, sa);
                        sb = fmaf(fz, wv, sb);
                    } else {
                        float f[EPV];
                        IO::unpack(v[u], f);
                        if constexpr (!BF) {
#pragma unroll
                            for (int i = 0; i < EPV; ++i) dot = fmaf(f[i], hj[i], dot);
                        }
                        if constexpr (INROW) dot = col_ok ? dot : 0.0f;
                        sa = fmaf(dot, rr[u].y, sa);
                        sb += r1[u];
#pragma unroll
                        for (int i = 0; i < EPV; ++i) acc[i] = fmaf(rr[u].x, f[i], acc[i]);
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        col_cur = col_nxt;
    }

    // ---- row epilogues
    if constexpr (KIND == 0) {
#pragma unroll
        for (int i = 0; i < EPV; ++i) acc[i] = slot_sum<LPR>(acc[i]);
        float den = slot_sum<LPR>(sb);
        if (it.chunk >= 0) {  // partial of a long row: (acc, den, local max) -> workspace, combined by gat_long_finalize_kernel
            if (slot == 0 && col_ok) {
                float* wsp = a.ws + it.chunk * a.ws_ld;
#pragma unroll
                for (int i = 0; i < EPV; ++i) wsp[c0 + i] = acc[i];
                if (hs == 0) {
                    wsp[a.ws_vec + head] = den;
                    wsp[a.ws_vec + a.heads + head] = 0.0f;     // local maximum: unused without max subtraction
                }
            }
        } else {
            if (slot == 0 && col_ok) {
                YT* yrow = static_cast<YT*>(a.Y) + row * a.ldy + c0;
                if (a.accumulate) {   // second half of a split adjacency: the first launch left (num, den) of the other columns here
                    den += a.out_a[row * a.heads + head];
#pragma unroll
                    for (int i = 0; i < EPV; ++i) acc[i] += load_scalar<YT>(yrow + i);
                }
                if (!a.raw) {
                    const float inv = 1.0f / den;  // 0/0 -> NaN for edgeless rows, as gatconv.py:139
#pragma unroll
                    for (int i = 0; i < EPV; ++i) {
                        float v = acc[i] * inv;
                        if (a.apply_elu) v = v > 0.0f ? v : (BF ? __expf(v) - 1.0f : expm1f(v));
                        acc[i] = v;
                    }
                }
                VecIO<YT, EPV>::store_nt(yrow, acc);
            }
            // the per-(row, head) scalars are written after every lane has read the previous launch's denominator
            __builtin_amdgcn_wave_barrier();
            if (slot == 0 && col_ok && hs == 0) {
                a.out_a[row * a.heads + head] = den;
            }
        }
    }
    if constexpr (ROWS) {
        const float sa_t = head_sum_c<LPH>(slot_sum<LPR>(sa)), sb_t = slot_sum<LPR>(sb);
        if constexpr (EXACT) {
            const float sw_t = head_sum_c<LPH>(slot_sum<LPR>(sw));
            if (slot == 0 && col_ok && hs == 0) {
                if (it.chunk >= 0) {          // bilinear in the partial sums: the finalize kernel combines (sa, sb, sw) of the chunks
                    float* wsp = a.ws + it.chunk * a.ws_ld + a.ws_vec;
                    wsp[head] = sa_t; wsp[a.heads + head] = sb_t; wsp[2 * a.heads + head] = sw_t;
                } else {
                    gat_finish_scores(a, row, head, sa_t, sb_t, sw_t);
                }
            }
        } else {
            const float ds = sa_t + dd * sb_t;
            if (slot == 0 && col_ok && hs == 0) {
                if (it.chunk >= 0) a.ws[it.chunk * a.ws_ld + a.ws_vec + head] = ds;
                else a.out_a[row * a.heads + head] = (a.accumulate == 1 ? a.out_a[row * a.heads + head] : 0.0f) + ds;
            }
        }
    }
    if constexpr (KIND == 2) {
#pragma unroll
        for (int i = 0; i < EPV; ++i) acc[i] = slot_sum<LPR>(acc[i]);
        const float dt = head_sum_c<LPH>(slot_sum<LPR>(sa)) + slot_sum<LPR>(sb);
        if (slot == 0 && col_ok) {
            if (it.chunk >= 0) {
                float* wsp = a.ws + it.chunk * a.ws_ld;
#pragma unroll
                for (int i = 0; i < EPV; ++i) wsp[c0 + i] = acc[i];
                if (hs == 0) wsp[a.ws_vec + head] = dt;
            } else {
                if (a.attn1) {   // d(scores)/dH: S = H.a1, T = H.a2 per head -- their gradients land here instead of in a GEMM + add
                    const float* p1 = attn_all[0] + sub * EPV;
                    const float* p2 = attn_all[KIND == 2 ? 1 : 0] + sub * EPV;
#pragma unroll
                    for (int i = 0; i < EPV; ++i) acc[i] += gs_row * p1[i] + dt * p2[i];
                }
                VecIO<YT, EPV>::store_nt(static_cast<YT*>(a.Y) + row * a.ldy + c0, acc);
                if (hs == 0) a.out_a[row * a.heads + head] = dt;
            }
        }
    }
   }
    it = nx;
    col_first = col_next_item;
    s_first = s_next;
  }
}

// ---- dispatch: (lanes per row, heads per wavefront) pairs with 1 <= LPR / NH ------------------------------------------------
template <typename XT, typename YT, int EPV, int LPR, int KIND, bool TROW = false>
static bool gat2_launch_nh(const EdgeArgs& a, int nh, dim3 grid, hipStream_t s, bool inrow) {
#define DGLL_GAT2(NHV) hipLaunchKernelGGL((gat2_kernel<XT, YT, EPV, LPR, NHV, 4, KIND, false, TROW>), grid, dim3(kBlock), 0, s, a); return true
    switch (nh) {
        case 1:
            if constexpr (!TROW) {
                if (inrow) { hipLaunchKernelGGL((gat2_kernel<XT, YT, EPV, LPR, 1, 4, KIND, true>), grid, dim3(kBlock), 0, s, a); return true; }
            }
            DGLL_GAT2(1);
        case 2: DGLL_GAT2(2);
        case 4: DGLL_GAT2(4);
        case 8: if constexpr (LPR >= 8) { DGLL_GAT2(8); } return false;
        default: return false;
    }
#undef DGLL_GAT2
}

template <typename XT, typename YT, int EPV, int KIND, bool TROW = false>
static bool gat2_launch_lpr(const EdgeArgs& a, int lpr, int nh, dim3 grid, hipStream_t s, bool inrow) {
    switch (lpr) {
        case 4: return gat2_launch_nh<XT, YT, EPV, 4, KIND, TROW>(a, nh, grid, s, inrow);
        case 8: return gat2_launch_nh<XT, YT, EPV, 8, KIND, TROW>(a, nh, grid, s, inrow);
        case 16: return gat2_launch_nh<XT, YT, EPV, 16, KIND, TROW>(a, nh, grid, s, inrow);
        case 32: return gat2_launch_nh<XT, YT, EPV, 32, KIND, TROW>(a, nh, grid, s, inrow);
        case 64: return gat2_launch_nh<XT, YT, EPV, 64, KIND, TROW>(a, nh, grid, s, inrow);
        default: return false;
    }
}

template <int KIND, bool TROW = false>
static bool gat2_launch_kind(int dtype, int lpr, int nh, dim3 grid, hipStream_t s, const EdgeArgs& a, bool inrow) {
    if (dtype == DGLL_F32) return gat2_launch_lpr<float, float, 4, KIND, TROW>(a, lpr, nh, grid, s, inrow);
    return gat2_launch_lpr<bf16_t, bf16_t, 8, KIND, TROW>(a, lpr, nh, grid, s, inrow);
}

}  // namespace dgll
