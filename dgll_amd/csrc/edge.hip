// edge.hip -- per-edge kernels of the GAT / SpecialSpmm path for gfx950: SDDMM, fused edge-softmax + SpMM
// (forward), its two backward gather passes, and the max-reduce of the SAGE aggregator.
//
// Reference semantics (all under /root/reference/dgll/nn/Convolution/):
//   sparseGatConv.forward  gatconv.py:111-148   e_ij = exp(-leakyrelu(a1.h_i + a2.h_j)); out_i = sum_j e_ij h_j / sum_j e_ij; elu
//   gatConv.forward        gatconv.py:30-54     att = softmax_j(+leakyrelu(.)) over the adjacency's nonzeros; out = att.Wh; elu
//   SpecialSpmmFunction    gatconv.py:60-81     backward: grad_values[e] = <g[row_e,:], b[col_e,:]>  (SDDMM), grad_b = A^T.g
//   SpGAT / GAT            gatconv.py:154-199   `nheads` independent heads, concatenated (one python call per head)
// Here all heads of a layer run in ONE launch: H is [N, heads*fo] (fo = per-head width, padded by the host to a
// power-of-two number of 16-byte vectors), S[i,k] = a1_k.h_i^k and T[j,k] = a2_k.h_j^k are [N, heads] fp32.
//
// Same skeleton as spmm.hip: one wavefront per row, LPR lanes x 16 bytes per gathered feature row, 64/LPR
// neighbour slots, U gathers in flight per lane, a coalesced 64-edge index batch handed out with ds_bpermute,
// fp32 accumulation, no atomics (fixed reduction order, bit-reproducible).  Nothing per-edge is ever stored:
// the backward passes recompute e_ij from S and T.
#include <algorithm>

#include "common.hpp"
#include "edge_args.hpp"

namespace dgll {

// ------------------------------------------------------------------------------------------------ SDDMM
// edge_out[k] = sum_f G[row(k), f] * H[col[k], f]            (gatconv.py:76-78)
template <typename XT, int EPV, int LPR, int U>
__global__ __launch_bounds__(kBlock) void sddmm_kernel(const EdgeArgs a) {
    typedef VecIO<XT, EPV> IO;
    constexpr int SLOTS = kWave / LPR;
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int sub = lane % LPR, slot = lane / LPR;
    const int c0 = sub * EPV;
    const bool col_ok = c0 < a.feat;
    const int64_t row = (int64_t)blockIdx.x * kWavesPerBlock + wave;
    if (row >= a.n_rows) return;
    const int64_t b = uniform64(a.rowptr[row]), e = uniform64(a.rowptr[row + 1]);
    float g[EPV];
    {
        typename IO::raw_t r = col_ok ? IO::load(static_cast<const XT*>(a.G) + row * a.ldg + c0) : IO::zero();
        IO::unpack(r, g);
    }
    const XT* hcol = static_cast<const XT*>(a.H) + (col_ok ? c0 : 0);
    for_each_batch(a.col, b, e, lane, [&](int nb, int cur_col, int64_t k0) {
        for (int j = 0; j < nb; j += SLOTS * U) {
            int idx[U];
            typename IO::raw_t v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                idx[u] = j + u * SLOTS + slot;
                const int c = __shfl(cur_col, idx[u] < nb ? idx[u] : nb - 1);
                v[u] = IO::load(hcol + (uint64_t)(uint32_t)c * (uint32_t)a.ldh);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float f[EPV];
                IO::unpack(v[u], f);
                float d = 0.0f;
#pragma unroll
                for (int i = 0; i < EPV; ++i) d = fmaf(g[i], f[i], d);
                d = head_sum(d, LPR);
                if (sub == 0 && idx[u] < nb) a.edge_out[k0 + idx[u]] = d;
            }
        }
    });
}

// ------------------------------------------------------------------------------------------------ GAT forward
// out[i, head cols] = act( sum_j w_ij * scale_ij * H[j, cols] / sum_j w_ij ),  w_ij = exp(sign*lrelu(S[i]+T[j]) - M[i])
template <typename XT, typename YT, int EPV, int LPR, int U>
__global__ __launch_bounds__(kBlock) void gat_fwd_kernel(const EdgeArgs a, int lph) {
    typedef VecIO<XT, EPV> IO;
    constexpr int SLOTS = kWave / LPR;
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int sub = lane % LPR, slot = lane / LPR;
    const int c0 = ((int)blockIdx.y * LPR + sub) * EPV;
    const bool col_ok = c0 < a.feat;
    const int head = col_ok ? c0 / a.fo : 0;
  for (int r = 0; r < a.rows_per_wave; ++r) {   // several consecutive rows per wavefront (amortises wave start-up)
    const WorkItem it = resolve_item(a, wave, r);
    if (it.done) return;
    if (!it.valid) continue;
    const int64_t row = it.row, b = it.b, e = it.e;
    const float s_i = a.S[row * a.heads + head];
    const XT* hcol = static_cast<const XT*>(a.H) + (col_ok ? c0 : 0);

    float m_i = 0.0f;
    if (a.use_max) {  // gatConv semantics: subtract the row maximum of the scores (softmax), gatconv.py:36
        float mx = -INFINITY;
        for_each_batch(a.col, b, e, lane, [&](int nb, int cur_col, int64_t) {
            for (int j = 0; j < nb; j += SLOTS) {  // uniform trip count: every lane takes part in the shuffle
                const int idx = j + slot;
                const int c = __shfl(cur_col, idx < nb ? idx : nb - 1);
                mx = fmaxf(mx, a.sign * lrelu(s_i + a.T[(int64_t)c * a.heads + head], a.alpha));
            }
        });
        m_i = slot_max<LPR>(mx);
    }

    float acc[EPV];
#pragma unroll
    for (int i = 0; i < EPV; ++i) acc[i] = 0.0f;
    float den = 0.0f;
    for_each_batch(a.col, b, e, lane, [&](int nb, int cur_col, int64_t k0) {
        for (int j = 0; j < nb; j += SLOTS * U) {
            bool ok[U];
            int idx[U];
            float t[U], sc[U];
            typename IO::raw_t v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                idx[u] = j + u * SLOTS + slot;
                ok[u] = idx[u] < nb;
                if (!ok[u]) idx[u] = nb - 1;
                const int c = __shfl(cur_col, idx[u]);
                t[u] = a.T[(int64_t)c * a.heads + head];
                sc[u] = a.edge_scale ? a.edge_scale[(k0 + idx[u]) * a.heads + head] : 1.0f;
                v[u] = IO::load(hcol + (uint64_t)(uint32_t)c * (uint32_t)a.ldh);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float w = __expf(a.sign * lrelu(s_i + t[u], a.alpha) - m_i);
                w = ok[u] ? w : 0.0f;
                den += w;
                const float wn = w * sc[u];
                float f[EPV];
                IO::unpack(ok[u] ? v[u] : IO::zero(), f);
#pragma unroll
                for (int i = 0; i < EPV; ++i) acc[i] = fmaf(wn, f[i], acc[i]);
            }
        }
    });
#pragma unroll
    for (int i = 0; i < EPV; ++i) acc[i] = slot_sum<LPR>(acc[i]);
    den = slot_sum<LPR>(den);
    if (it.chunk >= 0) {  // partial of a long row: (acc, den, local max) -> workspace, combined by gat_long_finalize_kernel
        if (slot == 0 && col_ok) {
            float* w = a.ws + it.chunk * a.ws_ld;
#pragma unroll
            for (int i = 0; i < EPV; ++i) w[c0 + i] = acc[i];
            if ((sub % lph) == 0) {
                w[a.ws_vec + head] = den;
                w[a.ws_vec + a.heads + head] = m_i;
            }
        }
        return;
    }
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
                if (a.apply_elu) v = v > 0.0f ? v : expm1f(v);
                acc[i] = v;
            }
        }
        VecIO<YT, EPV>::store(yrow, acc);
    }
    // the per-(row, head) scalars are written after every lane has read the previous launch's denominator
    __builtin_amdgcn_wave_barrier();
    if (slot == 0 && col_ok && (sub % lph) == 0) {
        a.out_a[row * a.heads + head] = den;
        if (a.out_b) a.out_b[row * a.heads + head] = m_i;
    }
  }
}

// ------------------------------------------------------------------------------------------------ GAT backward, pass 1 (rows of A)
// Per row i:  dhp = g * elu'(hp);  DN[i,:] = dhp / den_i;  DD[i,k] = -sum_{f in k} dhp*hp / den_i;
//             ds[i,k] = sum_j dz_ij,   dz_ij = (scale_ij * <DN_i, h_j>_k + DD[i,k]) * w_ij * sign * lrelu'(S_i+T_j)
template <typename XT, int EPV, int LPR, int U>
__global__ __launch_bounds__(kBlock) void gat_bwd_rows_kernel(const EdgeArgs a, int lph) {
    typedef VecIO<XT, EPV> IO;
    constexpr int SLOTS = kWave / LPR;
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int sub = lane % LPR, slot = lane / LPR;
    const int c0 = ((int)blockIdx.y * LPR + sub) * EPV;
    const bool col_ok = c0 < a.feat;
    const int head = col_ok ? c0 / a.fo : 0;
  for (int r = 0; r < a.rows_per_wave; ++r) {   // several consecutive rows per wavefront (amortises wave start-up)
    const WorkItem it = resolve_item(a, wave, r);
    if (it.done) return;
    if (!it.valid) continue;
    const int64_t row = it.row, b = it.b, e = it.e;
    const float s_i = a.S[row * a.heads + head];
    const float m_i = a.M ? a.M[row * a.heads + head] : 0.0f;
    const float inv_den = 1.0f / a.DEN[row * a.heads + head];

    float dn[EPV];
    float dd;
    {
        float g[EPV], o[EPV];
        IO::unpack(col_ok ? IO::load(static_cast<const XT*>(a.G) + row * a.ldg + c0) : IO::zero(), g);
        IO::unpack(col_ok ? IO::load(static_cast<const XT*>(a.O) + row * a.ldo + c0) : IO::zero(), o);
        float part = 0.0f;
#pragma unroll
        for (int i = 0; i < EPV; ++i) {
            float dhp = g[i], hp = o[i];
            if (a.apply_elu && o[i] <= 0.0f) {  // out = expm1(hp): elu'(hp) = out + 1, hp = log1p(out)
                const float op1 = o[i] + 1.0f;    // saturated ELU (out == -1): gradient 0, and 0 * log(0) must stay 0
                dhp = g[i] * op1;
                hp = op1 > 0.0f ? log1pf(o[i]) : 0.0f;
            }
            dn[i] = dhp * inv_den;
            // bf16: dd_i = -DN_i . hp_i from the ROUNDED DN_i the dot products (and the transposed pass) use -- see gat_kernel.hpp
            if (sizeof(XT) == 2) dn[i] = bf16_to_f32(f32_to_bf16(dn[i]));
            part = fmaf(dn[i], hp, part);
        }
        dd = -head_sum(part, lph);
        if (slot == 0 && col_ok && it.first && !a.accumulate) {   // per-row outputs: written once (row itself / first chunk, first launch)
            VecIO<XT, EPV>::store(static_cast<XT*>(a.Y) + row * a.ldy + c0, dn);
            if ((sub % lph) == 0) a.out_b[row * a.heads + head] = dd;
        }
    }
    uint32_t dnp[4] = {0u, 0u, 0u, 0u};   // DN_i as packed bf16 pairs for the dot2 path (exact: dn was rounded above)
    if constexpr (sizeof(XT) == 2) {
#pragma unroll
        for (int q = 0; q < 4; ++q) dnp[q] = pack_bf16x2(dn[2 * q], dn[2 * q + 1]);
    }

    const XT* hcol = static_cast<const XT*>(a.H) + (col_ok ? c0 : 0);
    float ds = 0.0f;
    for_each_batch(a.col, b, e, lane, [&](int nb, int cur_col, int64_t k0) {
        for (int j = 0; j < nb; j += SLOTS * U) {
            bool ok[U];
            int idx[U];
            float t[U], sc[U];
            typename IO::raw_t v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                idx[u] = j + u * SLOTS + slot;
                ok[u] = idx[u] < nb;
                if (!ok[u]) idx[u] = nb - 1;
                const int c = __shfl(cur_col, idx[u]);
                t[u] = a.T[(int64_t)c * a.heads + head];
                sc[u] = a.edge_scale ? a.edge_scale[(k0 + idx[u]) * a.heads + head] : 1.0f;
                v[u] = IO::load(hcol + (uint64_t)(uint32_t)c * (uint32_t)a.ldh);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float dot = 0.0f;
                if constexpr (sizeof(XT) == 2) {   // bf16: <DN_i, h_j> straight on the packed pairs (v_dot2c_f32_bf16, fp32 accumulate):
                    const uint32_t hv[4] = {v[u].x, v[u].y, v[u].z, v[u].w};   // 4 instructions instead of 8 unpacks + 8 FMAs
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        dot = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, dnp[q]), __builtin_bit_cast(bf16x2_t, hv[q]), dot, false);
                } else {
                    float f[EPV];
                    IO::unpack(v[u], f);
#pragma unroll
                    for (int i = 0; i < EPV; ++i) dot = fmaf(dn[i], f[i], dot);
                }
                dot = head_sum(dot, lph);
                const float z = s_i + t[u];
                const float w = __expf(a.sign * lrelu(z, a.alpha) - m_i);
                const float dz = (sc[u] * dot + dd) * w * a.sign * (z > 0.0f ? 1.0f : a.alpha);
                ds += ok[u] ? dz : 0.0f;
            }
        }
    });
    ds = slot_sum<LPR>(ds);
    if (slot == 0 && col_ok && (sub % lph) == 0) {
        if (it.chunk >= 0) a.ws[it.chunk * a.ws_ld + a.ws_vec + head] = ds;
        else a.out_a[row * a.heads + head] = (a.accumulate ? a.out_a[row * a.heads + head] : 0.0f) + ds;
    }
  }
}

// ------------------------------------------------------------------------------------------------ GAT backward, pass 2 (rows of A^T)
// Per source node j:  dH[j,:] = sum_i w_ij * scale_ij * DN[i,:];   dt[j,k] = sum_i dz_ij   (dz recomputed, see pass 1)
// Here rowptr/col describe A^T (rows = j, columns = i); H is DN (gathered by i); G is the node's own h_j row;
// S holds the per-node score of the ROW side of this pass (T of the forward), T the gathered side (S of the forward).
template <typename XT, typename YT, int EPV, int LPR, int U>
__global__ __launch_bounds__(kBlock) void gat_bwd_cols_kernel(const EdgeArgs a, int lph) {
    typedef VecIO<XT, EPV> IO;
    constexpr int SLOTS = kWave / LPR;
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int sub = lane % LPR, slot = lane / LPR;
    const int c0 = ((int)blockIdx.y * LPR + sub) * EPV;
    const bool col_ok = c0 < a.feat;
    const int head = col_ok ? c0 / a.fo : 0;
  for (int r = 0; r < a.rows_per_wave; ++r) {   // several consecutive rows per wavefront (amortises wave start-up)
    const WorkItem it = resolve_item(a, wave, r);
    if (it.done) return;
    if (!it.valid) continue;
    const int64_t row = it.row, b = it.b, e = it.e;
    const float t_j = a.S[row * a.heads + head];
    float hj[EPV];
    IO::unpack(col_ok ? IO::load(static_cast<const XT*>(a.G) + row * a.ldg + c0) : IO::zero(), hj);
    const XT* dncol = static_cast<const XT*>(a.H) + (col_ok ? c0 : 0);

    float acc[EPV];
#pragma unroll
    for (int i = 0; i < EPV; ++i) acc[i] = 0.0f;
    float dt = 0.0f;
    for_each_batch(a.col, b, e, lane, [&](int nb, int cur_col, int64_t k0) {
        for (int j = 0; j < nb; j += SLOTS * U) {
            bool ok[U];
            int idx[U];
            float s[U], m[U], dd[U], sc[U];
            typename IO::raw_t v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                idx[u] = j + u * SLOTS + slot;
                ok[u] = idx[u] < nb;
                if (!ok[u]) idx[u] = nb - 1;
                const int64_t c = __shfl(cur_col, idx[u]);
                s[u] = a.T[c * a.heads + head];
                m[u] = a.M ? a.M[c * a.heads + head] : 0.0f;
                dd[u] = a.DD[c * a.heads + head];
                sc[u] = a.edge_scale ? a.edge_scale[a.perm[k0 + idx[u]] * a.heads + head] : 1.0f;
                v[u] = IO::load(dncol + (uint64_t)(uint32_t)c * (uint32_t)a.ldh);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float f[EPV];
                IO::unpack(ok[u] ? v[u] : IO::zero(), f);
                float dot = 0.0f;
#pragma unroll
                for (int i = 0; i < EPV; ++i) dot = fmaf(f[i], hj[i], dot);
                dot = head_sum(dot, lph);
                const float z = s[u] + t_j;
                float w = __expf(a.sign * lrelu(z, a.alpha) - m[u]);
                w = ok[u] ? w : 0.0f;
                dt += (sc[u] * dot + dd[u]) * w * a.sign * (z > 0.0f ? 1.0f : a.alpha);
                const float wn = w * sc[u];
#pragma unroll
                for (int i = 0; i < EPV; ++i) acc[i] = fmaf(wn, f[i], acc[i]);
            }
        }
    });
#pragma unroll
    for (int i = 0; i < EPV; ++i) acc[i] = slot_sum<LPR>(acc[i]);
    dt = slot_sum<LPR>(dt);
    if (slot == 0 && col_ok) {
        if (it.chunk >= 0) {
            float* w = a.ws + it.chunk * a.ws_ld;
#pragma unroll
            for (int i = 0; i < EPV; ++i) w[c0 + i] = acc[i];
            if ((sub % lph) == 0) w[a.ws_vec + head] = dt;
        } else {
            VecIO<YT, EPV>::store(static_cast<YT*>(a.Y) + row * a.ldy + c0, acc);
            if ((sub % lph) == 0) a.out_a[row * a.heads + head] = dt;
        }
    }
  }
}

// Second pass for long rows of the three GAT kernels: combine the chunk partials in chunk order.
//   kind 0 (forward): softmax-merge (local max m_c, denominator d_c, vector v_c): M = max m_c, den = sum d_c e^{m_c-M},
//                     out = act(sum v_c e^{m_c-M} / den); writes out, rowsum, rowmax
//   kind 1 (backward rows): out_a[row, head] = sum of scalar partials
//   kind 2 (backward cols): Y[row, :] = sum of vector partials, out_a[row, head] = sum of scalar partials
template <typename YT>
__global__ __launch_bounds__(kBlock) void gat_long_finalize_kernel(const EdgeArgs a, const int64_t* __restrict__ long_row,
                                                                   const int32_t* __restrict__ long_chunk0, int kind) {
    const int64_t li = blockIdx.x;
    const int64_t row = long_row[li];
    const int cb = long_chunk0[li], ce = long_chunk0[li + 1];
    // per-(row, head) scalars are written after the whole block has read the previous launch's denominators
    int my_head = -1;
    float my_den = 0.0f, my_max = 0.0f;
    for (int f = (int)threadIdx.x; f < a.feat + a.heads; f += kBlock) {
        const bool vec = f < a.feat;
        const int head = vec ? f / a.fo : f - a.feat;
        const int off = vec ? f : a.ws_vec + head;
        if (kind == 0) {
            float M = -INFINITY;
            for (int c = cb; c < ce; ++c) M = fmaxf(M, a.ws[(int64_t)c * a.ws_ld + a.ws_vec + a.heads + head]);
            float den = 0.0f, v = 0.0f;
            for (int c = cb; c < ce; ++c) {
                const float* w = a.ws + (int64_t)c * a.ws_ld;
                const float sc = __expf(w[a.ws_vec + a.heads + head] - M);
                den += w[a.ws_vec + head] * sc;
                if (vec) v += w[f] * sc;
            }
            if (a.accumulate) den += a.out_a[row * a.heads + head];
            if (vec) {
                YT* y = static_cast<YT*>(a.Y) + row * a.ldy + f;
                if (a.accumulate) v += load_scalar<YT>(y);
                float o = a.raw ? v : v / den;
                if (!a.raw && a.apply_elu) o = o > 0.0f ? o : expm1f(o);
                store_one<YT>(y, o);
            } else {
                my_head = head; my_den = den; my_max = M;
            }
        } else if (kind == 1 && a.exact_dd) {
            // rows pass, exact dd: the chunks carry (sa, sb, sw); ds_i = sa + dd_i * sb with dd_i = -sw / den_i (gat_kernel.hpp)
            if (!vec) {
                float sa = 0.0f, sb = 0.0f, sw = 0.0f;
                for (int c = cb; c < ce; ++c) {
                    const float* w = a.ws + (int64_t)c * a.ws_ld + a.ws_vec;
                    sa += w[head]; sb += w[a.heads + head]; sw += w[2 * a.heads + head];
                }
                gat_finish_scores(a, row, head, sa, sb, sw);      // finalises, or parks / adds the sums in part3 (split launches)
            }
        } else {
            float sacc = 0.0f;
            for (int c = cb; c < ce; ++c) sacc += a.ws[(int64_t)c * a.ws_ld + off];
            if (vec) {
                if (kind == 2) {
                    if (a.attn1) {      // the scores' own contribution to grad_H (see edge_args.hpp): the head's dt first
                        float dt = 0.0f;
                        for (int c = cb; c < ce; ++c) dt += a.ws[(int64_t)c * a.ws_ld + a.ws_vec + head];
                        sacc += a.gs_rows[row * a.heads + head] * a.attn1[f] + dt * a.attn2[f];
                    }
                    store_one<YT>(static_cast<YT*>(a.Y) + row * a.ldy + f, sacc);
                }
            } else {
                my_head = head;
                my_den = ((kind == 1 && a.accumulate == 1) ? a.out_a[row * a.heads + head] : 0.0f) + sacc;
            }
        }
    }
    __syncthreads();
    if (my_head >= 0 && !(kind == 1 && a.exact_dd)) {       // (the exact rows pass wrote its scalars above)
        a.out_a[row * a.heads + my_head] = my_den;
        if (kind == 0 && a.out_b) a.out_b[row * a.heads + my_head] = my_max;
    }
}

// The same second pass with ONE WAVEFRONT per long row (heads <= 64): lane k < heads merges the per-head scalars and
// broadcasts them by shuffle; every lane then combines four columns at a time with float4 reads of the chunk partials.
// Most long rows have two or three chunks, so a whole workgroup per row (above) is mostly launch overhead: 0.83 -> ~0.2 ms
// for the 8 x 32 forward on the products-shaped graph.
template <typename YT>
__global__ __launch_bounds__(kBlock) void gat_long_finalize_wave_kernel(const EdgeArgs a, const int64_t* __restrict__ long_row,
                                                                        const int32_t* __restrict__ long_chunk0, int64_t n_long,
                                                                        int kind) {
    const int lane = lane_id();
    const int64_t li = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (li >= n_long) return;
    const int64_t row = long_row[li];
    const int cb = long_chunk0[li], ce = long_chunk0[li + 1];
    // ---- per-head scalars (lane = head)
    float h_max = 0.0f, h_den = 0.0f;
    if (lane < a.heads) {
        // Chunk partials are read EIGHT AT A TIME and combined in chunk order (the sums are the sequential ones, bit for bit): a hub of
        // ten thousand edges has dozens of chunks, and one dependent load per chunk made a launch's run time the latency chain of
        // its longest row (the SpMM's finalize kernel: 45 -> 12 us per launch with the same change).
        if (kind == 0) {
            float M = -INFINITY;
            float den = 0.0f;
            for (int c = cb; c < ce; c += 8) {
                float m8[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) m8[u] = a.ws[(int64_t)(c + u < ce ? c + u : ce - 1) * a.ws_ld + a.ws_vec + a.heads + lane];
#pragma unroll
                for (int u = 0; u < 8; ++u) M = fmaxf(M, m8[u]);                 // (a repeated last chunk changes no maximum)
            }
            for (int c = cb; c < ce; c += 8) {
                float d8[8], m8[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float* w = a.ws + (int64_t)(c + u < ce ? c + u : ce - 1) * a.ws_ld + a.ws_vec;
                    d8[u] = w[lane]; m8[u] = w[a.heads + lane];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (c + u < ce) den += d8[u] * __expf(m8[u] - M);
            }
            if (a.accumulate) den += a.out_a[row * a.heads + lane];
            h_max = M; h_den = den;
        } else if (kind == 1 && a.exact_dd) {
            float sa = 0.0f, sb = 0.0f, sw = 0.0f;
            for (int c = cb; c < ce; c += 8) {
                float a8[8], b8[8], w8[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float* w = a.ws + (int64_t)(c + u < ce ? c + u : ce - 1) * a.ws_ld + a.ws_vec;
                    a8[u] = w[lane]; b8[u] = w[a.heads + lane]; w8[u] = w[2 * a.heads + lane];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (c + u < ce) { sa += a8[u]; sb += b8[u]; sw += w8[u]; }
            }
            gat_finish_scores(a, row, lane, sa, sb, sw);                       // finalises, or parks / adds the sums in part3
        } else {
            float sacc = 0.0f;
            for (int c = cb; c < ce; c += 8) {
                float s8[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) s8[u] = a.ws[(int64_t)(c + u < ce ? c + u : ce - 1) * a.ws_ld + a.ws_vec + lane];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (c + u < ce) sacc += s8[u];
            }
            h_den = ((kind == 1 && a.accumulate == 1) ? a.out_a[row * a.heads + lane] : 0.0f) + sacc;
        }
    }
    // ---- vector part: four columns per lane (fo is a multiple of 4, so they share a head)
    if (kind != 1) {
        const int trips = (a.feat + kWave * 4 - 1) / (kWave * 4);
        for (int it = 0; it < trips; ++it) {
            const int f = it * kWave * 4 + lane * 4;
            const bool live = f < a.feat;
            const int head = live ? f / a.fo : 0;
            const float M = __shfl(h_max, head), den = __shfl(h_den, head);      // every lane takes part
            if (!live) continue;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int c = cb; c < ce; c += 8) {
                float4 p8[8];
                float m8[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float* w = a.ws + (int64_t)(c + u < ce ? c + u : ce - 1) * a.ws_ld;
                    p8[u] = *reinterpret_cast<const float4*>(w + f);
                    m8[u] = kind == 0 ? w[a.ws_vec + a.heads + head] : 0.0f;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (c + u >= ce) continue;
                    const float sc = kind == 0 ? __expf(m8[u] - M) : 1.0f;
                    const float4 p = p8[u];
                    v.x = fmaf(p.x, sc, v.x); v.y = fmaf(p.y, sc, v.y); v.z = fmaf(p.z, sc, v.z); v.w = fmaf(p.w, sc, v.w);
                }
            }
            float o[4] = {v.x, v.y, v.z, v.w};
            YT* y = static_cast<YT*>(a.Y) + row * a.ldy + f;
            if (kind == 2 && a.attn1) {      // the scores' own contribution to grad_H (edge_args.hpp); `den` is the head's dt here
                const float gs = a.gs_rows[row * a.heads + head];
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] += gs * a.attn1[f + i] + den * a.attn2[f + i];
            }
            if (kind == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (a.accumulate) o[i] += load_scalar<YT>(y + i);
                    if (!a.raw) {
                        o[i] = o[i] / den;
                        if (a.apply_elu) o[i] = o[i] > 0.0f ? o[i] : expm1f(o[i]);
                    }
                }
            }
            VecIO<YT, 4>::store(y, o);
        }
    }
    // per-(row, head) scalars: written after every lane has read the previous launch's denominators
    __builtin_amdgcn_wave_barrier();
    if (lane < a.heads && !(kind == 1 && a.exact_dd)) {       // (the exact rows pass wrote its scalars above)
        a.out_a[row * a.heads + lane] = h_den;
        if (kind == 0 && a.out_b) a.out_b[row * a.heads + lane] = h_max;
    }
}

// ------------------------------------------------------------------------------------------------ segment max
// Y[i,f] = max_k X[col[k], f], arg[i,f] = col of the (first) maximum; empty rows give 0 / -1.   (sageconv.py:37-38)
template <typename XT, int EPV, int LPR, int U>
__global__ __launch_bounds__(kBlock) void segment_max_kernel(const EdgeArgs a) {
    typedef VecIO<XT, EPV> IO;
    constexpr int SLOTS = kWave / LPR;
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int sub = lane % LPR, slot = lane / LPR;
    const int c0 = ((int)blockIdx.y * LPR + sub) * EPV;
    const bool col_ok = c0 < a.feat;
    const int64_t row = (int64_t)blockIdx.x * kWavesPerBlock + wave;
    if (row >= a.n_rows) return;
    const int64_t b = uniform64(a.rowptr[row]), e = uniform64(a.rowptr[row + 1]);
    const XT* hcol = static_cast<const XT*>(a.H) + (col_ok ? c0 : 0);
    float best[EPV];
    int arg[EPV];
#pragma unroll
    for (int i = 0; i < EPV; ++i) { best[i] = -INFINITY; arg[i] = -1; }
    for_each_batch(a.col, b, e, lane, [&](int nb, int cur_col, int64_t) {
        for (int j = 0; j < nb; j += SLOTS * U) {
            bool ok[U];
            int c[U];
            typename IO::raw_t v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = j + u * SLOTS + slot;
                ok[u] = idx < nb;
                c[u] = __shfl(cur_col, ok[u] ? idx : nb - 1);
                v[u] = IO::load(hcol + (uint64_t)(uint32_t)c[u] * (uint32_t)a.ldh);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float f[EPV];
                IO::unpack(v[u], f);
#pragma unroll
                for (int i = 0; i < EPV; ++i)
                    if (ok[u] && (f[i] > best[i] || (f[i] == best[i] && c[u] < arg[i]) || arg[i] < 0)) { best[i] = f[i]; arg[i] = c[u]; }
            }
        }
    });
#pragma unroll
    for (int off = LPR; off < kWave; off <<= 1) {
#pragma unroll
        for (int i = 0; i < EPV; ++i) {
            const float ob = __shfl_xor(best[i], off);
            const int oa = __shfl_xor(arg[i], off);
            if (oa >= 0 && (arg[i] < 0 || ob > best[i] || (ob == best[i] && oa < arg[i]))) { best[i] = ob; arg[i] = oa; }
        }
    }
    if (slot == 0 && col_ok) {
#pragma unroll
        for (int i = 0; i < EPV; ++i)
            if (arg[i] < 0) best[i] = 0.0f;
        VecIO<XT, EPV>::store(static_cast<XT*>(a.Y) + row * a.ldy + c0, best);
#pragma unroll
        for (int i = 0; i < EPV; ++i) a.arg_out[row * a.ldy + c0 + i] = arg[i];
    }
}

// ------------------------------------------------------------------------------------------------ segment max, backward
// grad[j, f] = sum over the destination rows i that list j (once per distinct i) of  (arg[i, f] == j) ? g[i, f] : 0
// -- the gradient of a max goes to the arg-max source row only (torch.max, sageconv.py:37-38).  Driven by the TRANSPOSED
// structure (row j of `rowptr/col` lists the destinations i, ascending), so every output row is produced by one
// wavefront in a fixed order: no atomics, bit-reproducible.  A pair (i, j) stored twice is counted once.
template <typename XT, int EPV, int LPR, int U>
__global__ __launch_bounds__(kBlock) void segment_max_bwd_kernel(const EdgeArgs a, const int32_t* __restrict__ arg, int64_t ldarg) {
    typedef VecIO<XT, EPV> IO;
    constexpr int SLOTS = kWave / LPR;
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int sub = lane % LPR, slot = lane / LPR;
    const int c0 = ((int)blockIdx.y * LPR + sub) * EPV;
    const bool col_ok = c0 < a.feat;
    const int64_t j = (int64_t)blockIdx.x * kWavesPerBlock + wave;      // source row whose gradient this wavefront writes
    if (j >= a.n_rows) return;
    const int64_t b = uniform64(a.rowptr[j]), e = uniform64(a.rowptr[j + 1]);
    const XT* gcol = static_cast<const XT*>(a.G) + (col_ok ? c0 : 0);
    const int32_t* acol = arg + (col_ok ? c0 : 0);
    float acc[EPV];
#pragma unroll
    for (int i = 0; i < EPV; ++i) acc[i] = 0.0f;
    for_each_batch(a.col, b, e, lane, [&](int nb, int cur_col, int64_t k0) {
        // a repeated (i, j) pair sits in adjacent slots of the sorted transposed row: keep the first
        int prev = __shfl_up(cur_col, 1);
        if (lane == 0) prev = k0 > b ? a.col[k0 - 1] : -1;
        const int keep = (lane < nb && cur_col != prev) ? 1 : 0;
        for (int jj = 0; jj < nb; jj += SLOTS * U) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = jj + u * SLOTS + slot;
                const int src = idx < nb ? idx : nb - 1;
                const int i = __shfl(cur_col, src);
                const int k = __shfl(keep, src);
                if (idx < nb && k) {
                    float f[EPV];
                    IO::unpack(IO::load(gcol + (uint64_t)(uint32_t)i * (uint32_t)a.ldg), f);
                    const int32_t* ap = acol + (int64_t)i * ldarg;
#pragma unroll
                    for (int q = 0; q < EPV; ++q)
                        if (ap[q] == (int32_t)j) acc[q] += f[q];
                }
            }
        }
    });
#pragma unroll
    for (int off = LPR; off < kWave; off <<= 1) {
#pragma unroll
        for (int i = 0; i < EPV; ++i) acc[i] += __shfl_xor(acc[i], off);
    }
    if (slot == 0 && col_ok) VecIO<XT, EPV>::store(static_cast<XT*>(a.Y) + j * a.ldy + c0, acc);
}

// ---- host-side dispatch helpers --------------------------------------------------------------------------
static int pick_lpr(int vecs) {
    int lpr = 4;
    while (lpr < 64 && lpr < vecs) lpr <<= 1;
    return lpr;
}

static bool vec_ok(const void* p, int64_t ld, int esz) { return aligned16(p) && (ld * esz) % 16 == 0; }

#define DGLL_LPR_SWITCH(lpr, CALL)   \
    switch (lpr) {                   \
        case 4: { CALL(4); } break;  \
        case 8: { CALL(8); } break;  \
        case 16: { CALL(16); } break;\
        case 32: { CALL(32); } break;\
        default: { CALL(64); } break;\
    }

}  // namespace dgll

int g_tune_gat_gen = 0;      // dgll_hip_debug_tune(9, v): 1 = first-generation GAT kernels only, 2 = second generation without the in-row form

using namespace dgll;

DGLL_API int dgll_hip_sddmm_csr(void* stream, const int64_t* rowptr, const int32_t* col, const void* G, int64_t ldg,
                                const void* B, int64_t ldb, int dtype, float* edge_out, int64_t n_rows, int feat) {
    if (n_rows <= 0) return DGLL_OK;
    DGLL_REQUIRE(rowptr && col && G && B && edge_out, "NULL argument");
    DGLL_REQUIRE(dtype == DGLL_F32 || dtype == DGLL_BF16, "dtype");
    const int esz = dtype == DGLL_BF16 ? 2 : 4, epv = 16 / esz;
    DGLL_REQUIRE(feat > 0 && feat <= 64 * epv, "sddmm handles up to 64 vectors per row in one launch (split columns on the host)");
    DGLL_REQUIRE(vec_ok(G, ldg, esz) && vec_ok(B, ldb, esz) && ldg >= ((feat + epv - 1) / epv) * epv && ldb >= ((feat + epv - 1) / epv) * epv,
                 "sddmm operands must be 16-byte aligned with padded leading dimensions");
    EdgeArgs a{};
    a.rowptr = rowptr; a.col = col; a.G = G; a.ldg = ldg; a.H = B; a.ldh = ldb; a.edge_out = edge_out;
    a.n_rows = n_rows; a.feat = feat;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int lpr = pick_lpr((feat + epv - 1) / epv);
    dim3 grid((uint32_t)((n_rows + kWavesPerBlock - 1) / kWavesPerBlock));
#define CALL(L)                                                                                                       \
    if (dtype == DGLL_F32) hipLaunchKernelGGL((sddmm_kernel<float, 4, L, 4>), grid, dim3(kBlock), 0, s, a);           \
    else hipLaunchKernelGGL((sddmm_kernel<bf16_t, 8, L, 4>), grid, dim3(kBlock), 0, s, a);
    DGLL_LPR_SWITCH(lpr, CALL)
#undef CALL
    DGLL_HIP_TRY(hipGetLastError());
    return DGLL_OK;
}

static int gat_ws_vec(int heads, int fo) { return (heads * fo + 7) & ~7; }
static int gat_ws_ld(int heads, int fo) { return (gat_ws_vec(heads, fo) + 3 * heads + 7) & ~7; }   // three scalars per head (exact rows pass)

DGLL_API size_t dgll_hip_gat_workspace_bytes(const dgll_csr_plan* plan, int heads, int fo) {
    if (!plan || plan->n_chunks == 0 || heads <= 0 || fo <= 0) return 0;
    return (size_t)plan->n_chunks * (size_t)gat_ws_ld(heads, fo) * sizeof(float);
}

// Attach the long-row schedule of `plan` (may be NULL) to the launch arguments and size the grid.
static int gat_schedule(EdgeArgs& a, const dgll_csr_plan* plan, int64_t n_rows, void* workspace, size_t workspace_bytes,
                        dim3* grid, int esz) {
    a.threshold = 0; a.n_chunks = 0; a.chunk_blocks = 0; a.ws = nullptr; a.rows_per_wave = 1;
    a.ws_vec = gat_ws_vec(a.heads, a.fo); a.ws_ld = gat_ws_ld(a.heads, a.fo);
    if (plan) {
        DGLL_REQUIRE(plan->n_rows == n_rows, "plan was built for a different CSR");
        a.threshold = plan->threshold;
        a.n_chunks = plan->n_chunks;
        a.chunk_begin = plan->d_chunk_begin; a.chunk_end = plan->d_chunk_end; a.chunk_row = plan->d_chunk_row;
        if (plan->n_chunks > 0) {
            const size_t need = dgll_hip_gat_workspace_bytes(plan, a.heads, a.fo);
            if (!workspace || workspace_bytes < need) {
                set_error("workspace too small for the plan's long-row partials");
                return DGLL_ERR_WORKSPACE;
            }
            a.ws = static_cast<float*>(workspace);
        }
        a.chunk_blocks = (uint32_t)((plan->n_chunks + kWavesPerBlock - 1) / kWavesPerBlock);
        // as in spmm.hip: ~96 KiB of gathered bytes per wavefront
        const double row_bytes = (double)plan->nnz / (double)std::max<int64_t>(n_rows, 1) * a.feat * (double)esz;
        a.rows_per_wave = std::min(std::max(row_bytes > 0 ? (int)(98304.0 / row_bytes) : 8, 1), 8);
    }
    const int64_t waves = (n_rows + a.rows_per_wave - 1) / a.rows_per_wave;
    grid->x = a.chunk_blocks + (uint32_t)((waves + kWavesPerBlock - 1) / kWavesPerBlock);
    return DGLL_OK;
}

template <typename YT>
static int gat_finalize(const EdgeArgs& a, const dgll_csr_plan* plan, int kind, hipStream_t s) {
    if (!plan || plan->n_long == 0) return DGLL_OK;
    constexpr uintptr_t kStore = 4 * sizeof(YT);      // the wavefront variant stores four columns at once
    const bool rows_aligned = (a.ldy * (int64_t)sizeof(YT)) % kStore == 0 && (reinterpret_cast<uintptr_t>(a.Y) % kStore) == 0;
    if (a.heads <= kWave && (kind == 1 || rows_aligned))
        hipLaunchKernelGGL(gat_long_finalize_wave_kernel<YT>, dim3((uint32_t)((plan->n_long + kWavesPerBlock - 1) / kWavesPerBlock)),
                           dim3(kBlock), 0, s, a, plan->d_long_row, plan->d_long_chunk0, plan->n_long, kind);
    else
        hipLaunchKernelGGL(gat_long_finalize_kernel<YT>, dim3((uint32_t)plan->n_long), dim3(kBlock), 0, s, a, plan->d_long_row,
                           plan->d_long_chunk0, kind);
    DGLL_HIP_TRY(hipGetLastError());
    return DGLL_OK;
}

static int gat_common(EdgeArgs& a, const int64_t* rowptr, const int32_t* col, int64_t n_rows, int heads, int fo, int dtype,
                      float alpha, int mode, int apply_elu) {
    DGLL_REQUIRE(rowptr && col, "NULL CSR");
    DGLL_REQUIRE(dtype == DGLL_F32 || dtype == DGLL_BF16, "dtype");
    DGLL_REQUIRE(mode == 0 || mode == 1, "mode");
    const int epv = dtype == DGLL_BF16 ? 8 : 4;
    DGLL_REQUIRE(heads > 0 && fo > 0, "heads/fo must be positive");
    DGLL_REQUIRE(fo % epv == 0, "per-head width must be a multiple of the 16-byte vector (pad on the host)");
    a.rowptr = rowptr; a.col = col; a.n_rows = n_rows; a.heads = heads; a.fo = fo; a.feat = heads * fo;
    a.alpha = alpha; a.sign = mode == 0 ? -1.0f : 1.0f; a.use_max = mode; a.apply_elu = apply_elu;
    a.vph = fo / epv; a.tstride = heads; a.sd_out = nullptr; a.sd_stride = 0;
    return DGLL_OK;
}

// Launch geometry of the second-generation kernels (gat_kernel.hpp): `nh` heads per wavefront on `lpr` = nh * lanes-per-head
// lanes per row.  They cover sparseGatConv's form -- exp(-leakyrelu), no max subtraction, no attention-dropout multipliers --
// for any per-head width that is a multiple of the 16-byte vector.  False: the first-generation kernels run.
static bool gat2_pick(const EdgeArgs& a, int* lpr, int* nh, uint32_t* grid_y) {
    if (g_tune_gat_gen == 1 || a.edge_scale || a.use_max || a.M) return false;
    int lph = 1;
    while (lph < a.vph) lph <<= 1;
    if (lph > kWave) return false;
    // blocks of 4 / 8 heads are read as float4s: whole blocks, 16-byte aligned
    // (row-score form, a.T == NULL: no score row is read at all)
    const bool vec = a.heads % 4 == 0 && (!a.T || (a.tstride % 4 == 0 && aligned16(a.T))) && (!a.DD || aligned16(a.DD));
    int n = 1;
    for (int cand = 8; cand >= 1; cand >>= 1) {
        if (cand * lph > kWave) continue;
        if (cand > 2 && !(vec && a.heads % cand == 0)) continue;
        if (cand > 1 && cand / 2 >= a.heads) continue;      // would leave half the wavefront's heads idle
        n = cand;
        break;
    }
    while (n * lph < 4) lph <<= 1;                           // at least 4 lanes per row slot (idle lanes inside a head)
    *lpr = n * lph; *nh = n; *grid_y = (uint32_t)((a.heads + n - 1) / n);
    return true;
}

// One head whose gathered-side scores sit right behind the last column of the gathered rows (same stride): the gather itself
// can bring them along (gat2_kernel INROW).  `second`: an array that must sit one float after `first` (dd after s), or NULL.
static bool gat2_inrow(const EdgeArgs& a, int lpr, int nh, int esz, const float* first, const float* second) {
    if (g_tune_gat_gen == 2 || nh != 1 || a.heads != 1 || a.vph >= lpr) return false;
    const char* slot = static_cast<const char*>(a.H) + (size_t)a.feat * esz;
    return reinterpret_cast<const char*>(first) == slot && (int64_t)a.tstride * 4 == a.ldh * esz &&
           (!second || second == first + 1) && (a.ldh - a.feat) * esz >= (second ? 8 : 4);
}

// Geometry of the first-generation kernels: per-head width a power-of-two number of vectors.
static int gat1_pick(const EdgeArgs& a, int epv, int* lph, int* lpr, uint32_t* grid_y) {
    *lph = a.fo / epv;
    DGLL_REQUIRE((*lph & (*lph - 1)) == 0 && *lph <= 64,
                 "per-head width / vector must be a power of two <= 64 for the max-subtracted / dropout form (pad on the host)");
    if (a.tstride != a.heads || a.sd_out) {
        set_error("strided score arrays need the second-generation kernels (mode 0, no attention dropout)");
        return DGLL_ERR_UNSUPPORTED;
    }
    const int vecs = a.feat / epv;
    *lpr = pick_lpr(vecs);
    if (*lpr < *lph) *lpr = *lph;
    *grid_y = (uint32_t)((vecs + *lpr - 1) / *lpr);
    return DGLL_OK;
}

// the row-score kernels address the gathered rows by 32-bit byte offsets formed with a 24-bit multiply
static bool rowscore_addressable(int64_t n_cols, int64_t ldh, int dtype) {
    const int64_t row_bytes = ldh * (dtype == DGLL_BF16 ? 2 : 4);
    return n_cols > 0 && n_cols <= (1 << 24) && row_bytes < (1 << 24) && n_cols * row_bytes <= (int64_t)0xffffffffll;
}

static int gat_fwd_impl(void* stream, const dgll_csr_plan* plan, const int64_t* rowptr, const int32_t* col,
                        const void* H, int64_t ldh, const float* S, const float* T, int t_stride, const float* edge_scale, void* out,
                        int64_t ldo, int dtype, float* rowsum, float* rowmax, int64_t n_rows, int heads, int fo,
                        float alpha, int apply_elu, int mode, void* workspace, size_t workspace_bytes, int raw, int accumulate,
                        const float* attn2 = nullptr) {
    if (n_rows <= 0) return DGLL_OK;
    EdgeArgs a{};
    int rc = gat_common(a, rowptr, col, n_rows, heads, fo, dtype, alpha, mode, apply_elu);
    if (rc != DGLL_OK) return rc;
    DGLL_REQUIRE(H && S && (T || attn2) && out && rowsum, "NULL argument");
    DGLL_REQUIRE(!attn2 || (!T && mode == 0 && !edge_scale), "the row-score form takes a2 INSTEAD of T (sparseGatConv's form, no attention dropout)");
    a.attn2 = attn2;
    DGLL_REQUIRE(mode == 0 || rowmax, "mode 1 needs a rowmax output");
    const int esz = dtype == DGLL_BF16 ? 2 : 4, epv = 16 / esz;
    DGLL_REQUIRE(vec_ok(H, ldh, esz) && vec_ok(out, ldo, esz) && ldh >= a.feat && ldo >= a.feat, "H/out must be 16-byte aligned");
    a.H = H; a.ldh = ldh; a.S = S; a.T = T; a.tstride = t_stride > 0 ? t_stride : heads; a.edge_scale = edge_scale; a.Y = out; a.ldy = ldo;
    a.out_a = rowsum; a.out_b = mode == 1 ? rowmax : nullptr;
    DGLL_REQUIRE(mode == 0 || (!raw && !accumulate), "split (raw / accumulate) launches support mode 0 only");
    a.raw = raw; a.accumulate = accumulate;
    dim3 grid(1, 1, 1);
    rc = gat_schedule(a, plan, n_rows, workspace, workspace_bytes, &grid, esz);
    if (rc != DGLL_OK) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    int lpr, nh, lph;
    if (attn2) {
        if (!gat2_pick(a, &lpr, &nh, &grid.y) || !gat2_launch_0r(dtype, lpr, nh, grid, s, a)) {
            set_error("no row-score GAT kernel for this head layout");
            return DGLL_ERR_UNSUPPORTED;
        }
    } else if (gat2_pick(a, &lpr, &nh, &grid.y)) {
        if (!gat2_launch_0(dtype, lpr, nh, grid, s, a, gat2_inrow(a, lpr, nh, esz, a.T, nullptr))) { set_error("no second-generation GAT kernel for this head layout"); return DGLL_ERR_UNSUPPORTED; }
    } else {
        rc = gat1_pick(a, epv, &lph, &lpr, &grid.y);
        if (rc != DGLL_OK) return rc;
#define CALL(L)                                                                                                                \
    if (dtype == DGLL_F32) hipLaunchKernelGGL((gat_fwd_kernel<float, float, 4, L, 4>), grid, dim3(kBlock), 0, s, a, lph);      \
    else hipLaunchKernelGGL((gat_fwd_kernel<bf16_t, bf16_t, 8, L, 4>), grid, dim3(kBlock), 0, s, a, lph);
        DGLL_LPR_SWITCH(lpr, CALL)
#undef CALL
    }
    DGLL_HIP_TRY(hipGetLastError());
    return dtype == DGLL_F32 ? gat_finalize<float>(a, plan, 0, s) : gat_finalize<bf16_t>(a, plan, 0, s);
}

DGLL_API int dgll_hip_gat_fwd(void* stream, const dgll_csr_plan* plan, const int64_t* rowptr, const int32_t* col,
                              const void* H, int64_t ldh, const float* S, const float* T, const float* edge_scale, void* out,
                              int64_t ldo, int dtype, float* rowsum, float* rowmax, int64_t n_rows, int heads, int fo,
                              float alpha, int apply_elu, int mode, void* workspace, size_t workspace_bytes) {
    return gat_fwd_impl(stream, plan, rowptr, col, H, ldh, S, T, 0, edge_scale, out, ldo, dtype, rowsum, rowmax, n_rows, heads, fo,
                        alpha, apply_elu, mode, workspace, workspace_bytes, 0, 0);
}

DGLL_API int dgll_hip_gat_fwd_ex(void* stream, const dgll_csr_plan* plan, const int64_t* rowptr, const int32_t* col,
                                 const void* H, int64_t ldh, const float* S, const float* T, const float* edge_scale, void* out,
                                 int64_t ldo, int dtype, float* rowsum, int64_t n_rows, int heads, int fo, float alpha,
                                 int apply_elu, void* workspace, size_t workspace_bytes, int raw, int accumulate) {
    return gat_fwd_impl(stream, plan, rowptr, col, H, ldh, S, T, 0, edge_scale, out, ldo, dtype, rowsum, nullptr, n_rows, heads, fo,
                        alpha, apply_elu, 0, workspace, workspace_bytes, raw, accumulate);
}

DGLL_API int dgll_hip_gat_fwd_strided(void* stream, const dgll_csr_plan* plan, const int64_t* rowptr, const int32_t* col,
                                      const void* H, int64_t ldh, const float* S, const float* T, int t_stride, void* out,
                                      int64_t ldo, int dtype, float* rowsum, int64_t n_rows, int heads, int fo, float alpha,
                                      int apply_elu, void* workspace, size_t workspace_bytes) {
    DGLL_REQUIRE(t_stride >= heads, "t_stride must be at least heads");
    return gat_fwd_impl(stream, plan, rowptr, col, H, ldh, S, T, t_stride, nullptr, out, ldo, dtype, rowsum, nullptr, n_rows, heads,
                        fo, alpha, apply_elu, 0, workspace, workspace_bytes, 0, 0);
}

DGLL_API int dgll_hip_gat_fwd_rowscore(void* stream, const dgll_csr_plan* plan, const int64_t* rowptr, const int32_t* col,
                                       const void* H, int64_t ldh, const float* S, const float* attn2, void* out, int64_t ldo, int dtype,
                                       float* rowsum, int64_t n_rows, int64_t n_cols, int heads, int fo, float alpha, int apply_elu,
                                       void* workspace, size_t workspace_bytes, int raw, int accumulate) {
    DGLL_REQUIRE(attn2, "attn2 (a2 of every head, laid out like a row of H) is required");
    if (!rowscore_addressable(n_cols, ldh, dtype)) {
        set_error("dgll_hip_gat_fwd_rowscore: H must have at most 2^24 rows and 4 GB (use dgll_hip_gat_fwd_strided)");
        return DGLL_ERR_UNSUPPORTED;
    }
    return gat_fwd_impl(stream, plan, rowptr, col, H, ldh, S, nullptr, 0, nullptr, out, ldo, dtype, rowsum, nullptr, n_rows, heads, fo,
                        alpha, apply_elu, 0, workspace, workspace_bytes, raw, accumulate, attn2);
}

// Pass 1 of the backward (rows of A or of one column-half of A): DN, DD and grad_S.  accumulate: 0 = first (or only) launch:
// writes DN, DD, grad_S, dd_i from the stored output row; 1 = a further launch over another column half (grad_S +=); 3 = DECLARED
// the only launch over these rows (second-generation kernels: exact dd_i from the pass's own dot products, gat_kernel.hpp).
static int gat_bwd_rows_impl(void* stream, const dgll_csr_plan* plan, const int64_t* rowptr, const int32_t* col,
                             const void* H, int64_t ldh, const float* S, const float* T, int t_stride, const float* edge_scale,
                             const void* out, int64_t ldo, const void* grad_out, int64_t ldg, int dtype,
                             const float* rowsum, const float* rowmax, void* dn, int64_t ldn, float* dd, float* sd_out,
                             int sd_stride, float* grad_S, int64_t n_rows, int heads, int fo, float alpha, int apply_elu,
                             int mode, int accumulate, void* workspace, size_t workspace_bytes, float* part3 = nullptr,
                             const float* attn2 = nullptr) {
    if (n_rows <= 0) return DGLL_OK;
    EdgeArgs a{};
    int rc = gat_common(a, rowptr, col, n_rows, heads, fo, dtype, alpha, mode, apply_elu);
    if (rc != DGLL_OK) return rc;
    DGLL_REQUIRE(H && S && (T || attn2) && out && grad_out && rowsum && dn && (dd || sd_out) && grad_S, "NULL argument");
    DGLL_REQUIRE(!attn2 || (!T && mode == 0 && !edge_scale && accumulate == 3),
                 "the row-score form takes a2 INSTEAD of T: sparseGatConv's form, one launch over the rows (exact dd_i)");
    a.attn2 = attn2;
    DGLL_REQUIRE(mode == 0 || rowmax, "mode 1 needs the forward's rowmax");
    const int esz = dtype == DGLL_BF16 ? 2 : 4, epv = 16 / esz;
    DGLL_REQUIRE(vec_ok(H, ldh, esz) && vec_ok(out, ldo, esz) && vec_ok(grad_out, ldg, esz) && vec_ok(dn, ldn, esz),
                 "matrices must be 16-byte aligned with padded leading dimensions");
    a.H = H; a.ldh = ldh; a.S = S; a.T = T; a.tstride = t_stride > 0 ? t_stride : heads; a.M = mode == 1 ? rowmax : nullptr;
    a.DEN = rowsum; a.edge_scale = edge_scale;
    a.G = grad_out; a.ldg = ldg; a.O = out; a.ldo = ldo; a.Y = dn; a.ldy = ldn; a.out_a = grad_S; a.out_b = dd;
    a.sd_out = sd_out; a.sd_stride = sd_stride;
    a.accumulate = accumulate;
    dim3 grid(1, 1, 1);
    rc = gat_schedule(a, plan, n_rows, workspace, workspace_bytes, &grid, esz);
    if (rc != DGLL_OK) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    int lpr, nh, lph;
    if (gat2_pick(a, &lpr, &nh, &grid.y)) {
        // 3: declared the only launch over these rows; 4 / 5 / 6: first / middle / last launch of a split exact pass (part3 carries
        // the partial sums): dd_i from the pass's own dot products
        a.exact_dd = accumulate == 3 ? 1 : (accumulate >= 4 && accumulate <= 6) ? accumulate - 2 : 0;
        DGLL_REQUIRE(a.exact_dd < 2 || part3, "a split exact rows pass needs the [n_rows, 3 * heads] partial-sum buffer");
        a.part3 = part3;
        a.accumulate = accumulate == 1 ? 1 : 0;
        const bool inrow = !attn2 && gat2_inrow(a, lpr, nh, esz, a.T, nullptr);
        const bool ok = attn2 ? gat2_launch_3r(dtype, lpr, nh, grid, s, a)
                              : (a.exact_dd ? gat2_launch_3(dtype, lpr, nh, grid, s, a, inrow) : gat2_launch_1(dtype, lpr, nh, grid, s, a, inrow));
        if (!ok) { set_error("no second-generation GAT kernel for this head layout"); return DGLL_ERR_UNSUPPORTED; }
    } else {
        DGLL_REQUIRE(!attn2, "the row-score form needs the second-generation kernels");
        rc = gat1_pick(a, epv, &lph, &lpr, &grid.y);
        if (rc != DGLL_OK) return rc;
        DGLL_REQUIRE(dd, "the first-generation rows pass writes dd");
        a.accumulate = (accumulate == 1 || accumulate == 5 || accumulate == 6) ? 1 : 0;    // first-generation kernels: dd from the stored output row in every mode
#define CALL(L)                                                                                                              \
    if (dtype == DGLL_F32) hipLaunchKernelGGL((gat_bwd_rows_kernel<float, 4, L, 2>), grid, dim3(kBlock), 0, s, a, lph);      \
    else hipLaunchKernelGGL((gat_bwd_rows_kernel<bf16_t, 8, L, 2>), grid, dim3(kBlock), 0, s, a, lph);
        DGLL_LPR_SWITCH(lpr, CALL)
#undef CALL
    }
    DGLL_HIP_TRY(hipGetLastError());
    return gat_finalize<float>(a, plan, 1, s);
}

DGLL_API int dgll_hip_gat_bwd_rows(void* stream, const dgll_csr_plan* plan, const int64_t* rowptr, const int32_t* col,
                                   const void* H, int64_t ldh, const float* S, const float* T, const float* edge_scale,
                                   const void* out, int64_t ldo, const void* grad_out, int64_t ldg, int dtype,
                                   const float* rowsum, const float* rowmax, void* dn, int64_t ldn, float* dd, float* grad_S,
                                   int64_t n_rows, int heads, int fo, float alpha, int apply_elu, int mode, int accumulate,
                                   void* workspace, size_t workspace_bytes) {
    DGLL_REQUIRE(dd, "NULL dd");
    return gat_bwd_rows_impl(stream, plan, rowptr, col, H, ldh, S, T, 0, edge_scale, out, ldo, grad_out, ldg, dtype, rowsum, rowmax,
                             dn, ldn, dd, nullptr, 0, grad_S, n_rows, heads, fo, alpha, apply_elu, mode, accumulate, workspace,
                             workspace_bytes);
}

// dgll_hip_gat_bwd_rows for a rows pass SPLIT over column halves of A (the partitioned path: owned-columns CSR, then halo-columns
// CSR) with the exact dd_i: phase 4 = first launch (writes DN; partial sums -> partial3), 5 = a middle launch, 6 = the last
// (adds partial3, writes dd and grad_S).  partial3: caller-owned fp32 [n_rows, 3 * heads], the same buffer for every phase.
DGLL_API int dgll_hip_gat_bwd_rows_split(void* stream, const dgll_csr_plan* plan, const int64_t* rowptr, const int32_t* col,
                                         const void* H, int64_t ldh, const float* S, const float* T, const void* out, int64_t ldo,
                                         const void* grad_out, int64_t ldg, int dtype, const float* rowsum, void* dn, int64_t ldn,
                                         float* dd, float* grad_S, int64_t n_rows, int heads, int fo, float alpha, int apply_elu,
                                         int phase, float* partial3, void* workspace, size_t workspace_bytes) {
    DGLL_REQUIRE(dd && partial3 && phase >= 4 && phase <= 6, "dgll_hip_gat_bwd_rows_split: phase 4 (first) / 5 (middle) / 6 (last), non-NULL dd and partial3");
    return gat_bwd_rows_impl(stream, plan, rowptr, col, H, ldh, S, T, 0, nullptr, out, ldo, grad_out, ldg, dtype, rowsum, nullptr,
                             dn, ldn, dd, nullptr, 0, grad_S, n_rows, heads, fo, alpha, apply_elu, 0, phase, workspace,
                             workspace_bytes, partial3);
}

// Pass 2 of the backward over a transposed structure (rows = source nodes j, columns = destination rows i):
// grad_H[j] = sum_i w_ij scale_ij DN[i], grad_T[j] = sum_i dz_ij.  Hrow / T_row belong to the rows of this pass, DN / S_col /
// DD / M_col to its columns (`col_stride` floats per node for S_col and dd_col; 0 = heads).
static int gat_bwd_cols_impl(void* stream, const dgll_csr_plan* t_plan, const int64_t* t_rowptr, const int32_t* t_col,
                             const int64_t* t_perm, const void* dn, int64_t ldn, const void* Hrow, int64_t ldh,
                             const float* T_row, const float* S_col, const float* dd_col, int col_stride, const float* rowmax_col,
                             const float* edge_scale, void* grad_H, int64_t ldgh, float* grad_T, int dtype,
                             int64_t n_rows_t, int heads, int fo, float alpha, int mode, void* workspace,
                             size_t workspace_bytes, const float* attn1 = nullptr, const float* attn2 = nullptr,
                             const float* grad_S_rows = nullptr) {
    if (n_rows_t <= 0) return DGLL_OK;
    EdgeArgs t{};
    int rc = gat_common(t, t_rowptr, t_col, n_rows_t, heads, fo, dtype, alpha, mode, 0);
    if (rc != DGLL_OK) return rc;
    DGLL_REQUIRE(dn && Hrow && T_row && S_col && dd_col && grad_H && grad_T, "NULL argument");
    DGLL_REQUIRE(mode == 0 || rowmax_col, "mode 1 needs the forward's rowmax");
    DGLL_REQUIRE(!edge_scale || t_perm, "edge_scale needs the transpose permutation");
    const int esz = dtype == DGLL_BF16 ? 2 : 4, epv = 16 / esz;
    DGLL_REQUIRE(vec_ok(dn, ldn, esz) && vec_ok(Hrow, ldh, esz) && vec_ok(grad_H, ldgh, esz),
                 "matrices must be 16-byte aligned with padded leading dimensions");
    t.perm = t_perm; t.H = dn; t.ldh = ldn; t.G = Hrow; t.ldg = ldh; t.S = T_row; t.T = S_col; t.DD = dd_col;
    t.tstride = col_stride > 0 ? col_stride : heads;
    DGLL_REQUIRE(!attn1 || (attn2 && grad_S_rows), "the score-gradient epilogue needs a1, a2 and the rows' grad_S");
    t.attn1 = attn1; t.attn2 = attn2; t.gs_rows = grad_S_rows;
    t.M = mode == 1 ? rowmax_col : nullptr; t.edge_scale = edge_scale; t.Y = grad_H; t.ldy = ldgh; t.out_a = grad_T;
    t.out_b = nullptr;
    dim3 grid(1, 1, 1);
    rc = gat_schedule(t, t_plan, n_rows_t, workspace, workspace_bytes, &grid, esz);
    if (rc != DGLL_OK) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    int lpr, nh, lph;
    if (gat2_pick(t, &lpr, &nh, &grid.y)) {
        if (!gat2_launch_2(dtype, lpr, nh, grid, s, t, gat2_inrow(t, lpr, nh, esz, t.T, t.DD))) { set_error("no second-generation GAT kernel for this head layout"); return DGLL_ERR_UNSUPPORTED; }
    } else {
        rc = gat1_pick(t, epv, &lph, &lpr, &grid.y);
        if (rc != DGLL_OK) return rc;
        DGLL_REQUIRE(!attn1, "the score-gradient epilogue needs the second-generation kernels");
#define CALL(L)                                                                                                                  \
    if (dtype == DGLL_F32) hipLaunchKernelGGL((gat_bwd_cols_kernel<float, float, 4, L, 2>), grid, dim3(kBlock), 0, s, t, lph);   \
    else hipLaunchKernelGGL((gat_bwd_cols_kernel<bf16_t, bf16_t, 8, L, 2>), grid, dim3(kBlock), 0, s, t, lph);
        DGLL_LPR_SWITCH(lpr, CALL)
#undef CALL
    }
    DGLL_HIP_TRY(hipGetLastError());
    return dtype == DGLL_F32 ? gat_finalize<float>(t, t_plan, 2, s) : gat_finalize<bf16_t>(t, t_plan, 2, s);
}

DGLL_API int dgll_hip_gat_bwd_cols(void* stream, const dgll_csr_plan* t_plan, const int64_t* t_rowptr, const int32_t* t_col,
                                   const int64_t* t_perm, const void* dn, int64_t ldn, const void* Hrow, int64_t ldh,
                                   const float* T_row, const float* S_col, const float* dd_col, const float* rowmax_col,
                                   const float* edge_scale, void* grad_H, int64_t ldgh, float* grad_T, int dtype,
                                   int64_t n_rows_t, int heads, int fo, float alpha, int mode, void* workspace,
                                   size_t workspace_bytes) {
    return gat_bwd_cols_impl(stream, t_plan, t_rowptr, t_col, t_perm, dn, ldn, Hrow, ldh, T_row, S_col, dd_col, 0, rowmax_col,
                             edge_scale, grad_H, ldgh, grad_T, dtype, n_rows_t, heads, fo, alpha, mode, workspace, workspace_bytes);
}

DGLL_API int dgll_hip_gat_bwd(void* stream, const dgll_csr_plan* plan, const dgll_csr_plan* t_plan,
                              const int64_t* rowptr, const int32_t* col,          /* A   */
                              const int64_t* t_rowptr, const int32_t* t_col, const int64_t* t_perm, /* A^T */
                              const void* H, int64_t ldh, const float* S, const float* T, const float* edge_scale,
                              const void* out, int64_t ldo, const void* grad_out, int64_t ldg, int dtype,
                              const float* rowsum, const float* rowmax,
                              void* dn_scratch, int64_t ldn, float* dd_scratch,
                              void* grad_H, int64_t ldgh, float* grad_S, float* grad_T,
                              int64_t n_rows, int64_t n_cols, int heads, int fo, float alpha, int apply_elu, int mode,
                              void* workspace, size_t workspace_bytes) {
    DGLL_REQUIRE(t_rowptr && t_col, "NULL transposed CSR");
    int rc = dgll_hip_gat_bwd_rows(stream, plan, rowptr, col, H, ldh, S, T, edge_scale, out, ldo, grad_out, ldg, dtype, rowsum,
                                   rowmax, dn_scratch, ldn, dd_scratch, grad_S, n_rows, heads, fo, alpha, apply_elu, mode, 3,
                                   workspace, workspace_bytes);
    if (rc != DGLL_OK) return rc;
    // the same scratch is reused: pass 2 is stream-ordered after pass 1
    return dgll_hip_gat_bwd_cols(stream, t_plan, t_rowptr, t_col, t_perm, dn_scratch, ldn, H, ldh, T, S, dd_scratch, rowmax,
                                 edge_scale, grad_H, ldgh, grad_T, dtype, n_cols, heads, fo, alpha, mode, workspace,
                                 workspace_bytes);
}

// Both backward passes of the sparseGatConv form (mode 0, no attention dropout) with strided score arrays:
//   T            gathered by the rows pass at T[j * t_stride + head] -- a compact [n_cols, heads] array (t_stride = heads) or
//                a slot in the padding of the H rows themselves (then the score costs no extra cache line per edge);
//   sd_scratch   fp32, `sd_stride` floats per destination row: the rows pass leaves {s_i[0:heads], dd_i[0:heads]} side by side
//                there and the transposed pass gathers both with one line fill -- a separate [n_rows, 2 * heads] buffer
//                (sd_stride = 2 * heads) or a slot in the padding of the dn_scratch rows.
DGLL_API int dgll_hip_gat_bwd_rows_strided(void* stream, const dgll_csr_plan* plan, const int64_t* rowptr, const int32_t* col,
                                           const void* H, int64_t ldh, const float* S, const float* T, int t_stride,
                                           const void* out, int64_t ldo, const void* grad_out, int64_t ldg, int dtype,
                                           const float* rowsum, void* dn_scratch, int64_t ldn, float* sd_scratch, int sd_stride,
                                           float* grad_S, int64_t n_rows, int heads, int fo, float alpha, int apply_elu,
                                           void* workspace, size_t workspace_bytes) {
    DGLL_REQUIRE(sd_scratch && sd_stride >= 2 * heads && t_stride >= heads, "bad strided score arguments");
    return gat_bwd_rows_impl(stream, plan, rowptr, col, H, ldh, S, T, t_stride, nullptr, out, ldo, grad_out, ldg, dtype, rowsum,
                             nullptr, dn_scratch, ldn, nullptr, sd_scratch, sd_stride, grad_S, n_rows, heads, fo, alpha,
                             apply_elu, 0, 3, workspace, workspace_bytes);      // the strided form is always the only launch over its rows
}

DGLL_API int dgll_hip_gat_bwd_rows_rowscore(void* stream, const dgll_csr_plan* plan, const int64_t* rowptr, const int32_t* col,
                                            const void* H, int64_t ldh, const float* S, const float* attn2,
                                            const void* out, int64_t ldo, const void* grad_out, int64_t ldg, int dtype,
                                            const float* rowsum, void* dn_scratch, int64_t ldn, float* sd_scratch, int sd_stride,
                                            float* grad_S, int64_t n_rows, int64_t n_cols, int heads, int fo, float alpha, int apply_elu,
                                            void* workspace, size_t workspace_bytes) {
    DGLL_REQUIRE(sd_scratch && sd_stride >= 2 * heads && attn2, "bad row-score arguments");
    if (!rowscore_addressable(n_cols, ldh, dtype)) {
        set_error("dgll_hip_gat_bwd_rows_rowscore: H must have at most 2^24 rows and 4 GB (use dgll_hip_gat_bwd_rows_strided)");
        return DGLL_ERR_UNSUPPORTED;
    }
    return gat_bwd_rows_impl(stream, plan, rowptr, col, H, ldh, S, nullptr, 0, nullptr, out, ldo, grad_out, ldg, dtype, rowsum,
                             nullptr, dn_scratch, ldn, nullptr, sd_scratch, sd_stride, grad_S, n_rows, heads, fo, alpha,
                             apply_elu, 0, 3, workspace, workspace_bytes, nullptr, attn2);
}

DGLL_API int dgll_hip_gat_bwd_cols_strided(void* stream, const dgll_csr_plan* t_plan, const int64_t* t_rowptr,
                                           const int32_t* t_col, const void* dn_scratch, int64_t ldn, const void* H,
                                           int64_t ldh, const float* T_rows, const float* sd_scratch, int sd_stride,
                                           void* grad_H, int64_t ldgh, float* grad_T, int dtype, int64_t n_cols, int heads,
                                           int fo, float alpha, void* workspace, size_t workspace_bytes,
                                           const float* attn1, const float* attn2, const float* grad_S_rows) {
    DGLL_REQUIRE(sd_scratch && sd_stride >= 2 * heads && T_rows, "bad strided score arguments");
    return gat_bwd_cols_impl(stream, t_plan, t_rowptr, t_col, nullptr, dn_scratch, ldn, H, ldh, T_rows, sd_scratch,
                             sd_scratch + heads, sd_stride, nullptr, nullptr, grad_H, ldgh, grad_T, dtype, n_cols, heads, fo,
                             alpha, 0, workspace, workspace_bytes, attn1, attn2, grad_S_rows);
}

DGLL_API int dgll_hip_gat_bwd_strided(void* stream, const dgll_csr_plan* plan, const dgll_csr_plan* t_plan,
                                      const int64_t* rowptr, const int32_t* col, const int64_t* t_rowptr, const int32_t* t_col,
                                      const void* H, int64_t ldh, const float* S, const float* T, int t_stride,
                                      const float* T_rows, const void* out, int64_t ldo, const void* grad_out, int64_t ldg,
                                      int dtype, const float* rowsum, void* dn_scratch, int64_t ldn, float* sd_scratch,
                                      int sd_stride, void* grad_H, int64_t ldgh, float* grad_S, float* grad_T, int64_t n_rows,
                                      int64_t n_cols, int heads, int fo, float alpha, int apply_elu, void* workspace,
                                      size_t workspace_bytes) {
    DGLL_REQUIRE(t_rowptr && t_col, "NULL transposed CSR");
    int rc = dgll_hip_gat_bwd_rows_strided(stream, plan, rowptr, col, H, ldh, S, T, t_stride, out, ldo, grad_out, ldg, dtype,
                                           rowsum, dn_scratch, ldn, sd_scratch, sd_stride, grad_S, n_rows, heads, fo, alpha,
                                           apply_elu, workspace, workspace_bytes);
    if (rc != DGLL_OK) return rc;
    // the same workspace is reused: pass 2 is stream-ordered after pass 1
    return dgll_hip_gat_bwd_cols_strided(stream, t_plan, t_rowptr, t_col, dn_scratch, ldn, H, ldh, T_rows, sd_scratch, sd_stride,
                                         grad_H, ldgh, grad_T, dtype, n_cols, heads, fo, alpha, workspace, workspace_bytes,
                                         nullptr, nullptr, nullptr);
}

DGLL_API int dgll_hip_segment_max(void* stream, const int64_t* rowptr, const int32_t* col, const void* X, int64_t ldx,
                                  void* Y, int32_t* arg, int64_t ldy, int dtype, int64_t n_rows, int feat) {
    if (n_rows <= 0 || feat <= 0) return DGLL_OK;
    DGLL_REQUIRE(rowptr && col && X && Y && arg, "NULL argument");
    DGLL_REQUIRE(dtype == DGLL_F32 || dtype == DGLL_BF16, "dtype");
    const int esz = dtype == DGLL_BF16 ? 2 : 4, epv = 16 / esz;
    const int vecs = (feat + epv - 1) / epv;
    DGLL_REQUIRE(vec_ok(X, ldx, esz) && vec_ok(Y, ldy, esz) && ldx >= vecs * epv && ldy >= vecs * epv,
                 "segment_max operands must be 16-byte aligned with padded leading dimensions");
    EdgeArgs a{};
    a.rowptr = rowptr; a.col = col; a.H = X; a.ldh = ldx; a.Y = Y; a.ldy = ldy; a.arg_out = arg; a.n_rows = n_rows;
    a.feat = vecs * epv;  // padded columns are computed too (they live inside the padded leading dimension)
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int lpr = pick_lpr(vecs);
    dim3 grid((uint32_t)((n_rows + kWavesPerBlock - 1) / kWavesPerBlock), (uint32_t)((vecs + lpr - 1) / lpr));
#define CALL(L)                                                                                                          \
    if (dtype == DGLL_F32) hipLaunchKernelGGL((segment_max_kernel<float, 4, L, 4>), grid, dim3(kBlock), 0, s, a);        \
    else hipLaunchKernelGGL((segment_max_kernel<bf16_t, 8, L, 4>), grid, dim3(kBlock), 0, s, a);
    DGLL_LPR_SWITCH(lpr, CALL)
#undef CALL
    DGLL_HIP_TRY(hipGetLastError());
    return DGLL_OK;
}

DGLL_API int dgll_hip_segment_max_bwd(void* stream, const int64_t* t_rowptr, const int32_t* t_col, const void* G, int64_t ldg,
                                      const int32_t* arg, int64_t ldarg, void* grad, int64_t ldgrad, int dtype,
                                      int64_t n_src, int feat) {
    if (n_src <= 0 || feat <= 0) return DGLL_OK;
    DGLL_REQUIRE(t_rowptr && t_col && G && arg && grad, "NULL argument");
    DGLL_REQUIRE(dtype == DGLL_F32 || dtype == DGLL_BF16, "dtype");
    const int esz = dtype == DGLL_BF16 ? 2 : 4, epv = 16 / esz;
    const int vecs = (feat + epv - 1) / epv;
    DGLL_REQUIRE(vec_ok(G, ldg, esz) && vec_ok(grad, ldgrad, esz) && ldg >= vecs * epv && ldgrad >= vecs * epv && ldarg >= vecs * epv,
                 "segment_max_bwd operands must be 16-byte aligned with padded leading dimensions");
    EdgeArgs a{};
    a.rowptr = t_rowptr; a.col = t_col; a.G = G; a.ldg = ldg; a.Y = grad; a.ldy = ldgrad; a.n_rows = n_src;
    a.feat = vecs * epv;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int lpr = pick_lpr(vecs);
    dim3 grid((uint32_t)((n_src + kWavesPerBlock - 1) / kWavesPerBlock), (uint32_t)((vecs + lpr - 1) / lpr));
#define CALL(L)                                                                                                                  \
    if (dtype == DGLL_F32) hipLaunchKernelGGL((segment_max_bwd_kernel<float, 4, L, 4>), grid, dim3(kBlock), 0, s, a, arg, ldarg);  \
    else hipLaunchKernelGGL((segment_max_bwd_kernel<bf16_t, 8, L, 4>), grid, dim3(kBlock), 0, s, a, arg, ldarg);
    DGLL_LPR_SWITCH(lpr, CALL)
#undef CALL
    DGLL_HIP_TRY(hipGetLastError());
    return DGLL_OK;
}
