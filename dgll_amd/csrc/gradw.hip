// gradw.hip -- the weight gradients of a dense transform next to the aggregation:  dW1 = X1^T . G  and  dW2 = X2^T . G
// (sageconv.py:41,72-75 and gcnconv.py:30 leave these to autograd: two library GEMMs that each read G again).
// bf16 operands, fp32 accumulation on the matrix cores, G read ONCE for both products.
//
// The reduction runs over the ROWS of both operands (M ~ 2.4 M), which is the slow dimension in memory: an MFMA lane has
// to hold 8 consecutive rows of ONE column.  Layout trick instead of an LDS transposition: lane i of a half-wave loads the
// DWORD holding columns (2 i, 2 i + 1) of eight consecutive rows -- every load instruction reads two whole 128-byte row
// segments -- and two v_perm_b32 per row pair split the dwords into the fragment of column 2 i and the fragment of column
// 2 i + 1.  An MFMA over the "even" fragments of X and G therefore produces the outputs (2 i, 2 j) of a 64 x 64 super-tile,
// the other three combinations the rest; the final store undoes the interleave.  Plain loads that the compiler counts
// itself; the dwords come from an LDS tile the workgroup fills with line-shaped loads (see the kernel).
//
// Work split: S row slabs x T workgroup types.  A workgroup is 8 waves = 4 column slabs of 64 of the concatenated
// operand [X1 | X2] x 2 halves of G's 256 columns; each wave owns a 64 x 128 output tile (128 accumulator registers) and
// walks its slab 64 rows per step (four 16-row MFMA sub-steps), two steps of loads in flight behind the MFMAs of the current one.
// Slab partials go to a workspace [S][Kc][N] fp32 and are summed in slab order by a second kernel: deterministic.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

#include "common.hpp"

namespace dgll {

typedef __attribute__((ext_vector_type(8))) __bf16 gw_bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float gw_f32x16_t;
typedef __attribute__((ext_vector_type(4))) int gw_i32x4_t;
typedef __attribute__((ext_vector_type(4))) uint32_t gw_u32x4_t;

struct GradWArgs {
    const void* X[2];      // bf16 [M, ldx]
    int64_t ldx[2];
    int K[2];              // columns of each operand (K[1] = 0: single product)
    const void* G;         // bf16 [M, ldg]
    int64_t ldg;
    int N;
    int64_t M;
    int64_t rows_per_slab; // multiple of 16
    int n_slabs;
    int kslabs[2];         // 64-column slabs of each operand
    float* partial;        // [n_slabs][Kc = 64 * (kslabs[0] + kslabs[1])][Np = 256] fp32
};

typedef __amdgpu_buffer_rsrc_t gw_rsrc_t;
__device__ __forceinline__ gw_rsrc_t gw_rsrc(const void* base, uint64_t bytes) {      // raw dword access, hardware range check
    const uint32_t n = bytes > 0xffffffffull ? 0xffffffffu : (uint32_t)bytes;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, (int)n, 0x00020000);
}

// eight consecutive rows of one dword column -> the fragments of its even and of its odd bf16 column
__device__ __forceinline__ void gw_split(const uint32_t (&r)[8], gw_bf16x8_t& even, gw_bf16x8_t& odd) {
    gw_u32x4_t e, o;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        e[p] = __builtin_amdgcn_perm(r[2 * p + 1], r[2 * p], 0x05040100u);   // lo16(r[2p]) | lo16(r[2p+1]) << 16
        o[p] = __builtin_amdgcn_perm(r[2 * p + 1], r[2 * p], 0x07060302u);   // hi16(r[2p]) | hi16(r[2p+1]) << 16
    }
    even = __builtin_bit_cast(gw_bf16x8_t, e);
    odd = __builtin_bit_cast(gw_bf16x8_t, o);
}

// One step = kGwU sub-steps of 16 rows.  The workgroup loads a sub-step's 16 KB ONCE, line shaped (a wave instruction
// covers whole 128-byte row segments), parks it in LDS and every wave picks its dword columns from there.  (First version: every wave loaded
// its own dwords straight from global memory -- each byte through the L1 three times; 2.9 TB/s, and slower with more
// loads in flight.)  LDS tile of a step: X part [4 column slabs][16 rows][128 B], G part [16 rows][512 B]; two tiles.
constexpr int kGwU = 4;                       // 16-row sub-steps per step: one barrier per 64 rows (2 and 4 measured equal; 1: -8 %)
constexpr int kGwSub = 16 * 512 * 2;          // LDS bytes of a 16-row sub-tile (X part + G part)
constexpr int kGwTile = kGwU * kGwSub;

struct GwStage { gw_u32x4_t x[kGwU], g[kGwU]; };

__global__ __launch_bounds__(512) void gradw_splitk_kernel(const GradWArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];   // 2 x kGwTile
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l32 = lane & 31;
    const int slab = blockIdx.x, type = blockIdx.y;
    const int ks_total = a.kslabs[0] + a.kslabs[1];
    const int64_t r_begin = (int64_t)slab * a.rows_per_slab;
    int64_t r_end = r_begin + a.rows_per_slab;
    if (r_end > a.M) r_end = a.M;
    if (r_begin >= r_end) return;                             // whole workgroup: uniform
    const int steps = (int)((r_end - r_begin + 16 * kGwU - 1) / (16 * kGwU));
    const uint32_t ldg_b = (uint32_t)(a.ldg * 2);

    // ---- loader role: wave w fetches column slab w / 2 of the operand, rows 8 (w % 2) + lane / 8, and rows 2 w, 2 w + 1 of G
    const int lks = type * 4 + (wave >> 1);
    const bool l_has_x = lks < ks_total;
    const bool l_second = lks >= a.kslabs[0];
    const uint32_t l_lda_b = (uint32_t)((l_second ? a.ldx[1] : a.ldx[0]) * 2);
    const char* lX = static_cast<const char*>(l_second ? a.X[1] : a.X[0]);
    const int l_kcol0 = (l_second ? lks - a.kslabs[0] : lks) * 64;
    // descriptors based at the slab's first row; records end with the slab's last row: the ragged last step reads zeros
    // The records end inside the slab's LAST row at that row's own (16-byte rounded) valid columns, not a whole pitch later: an
    // operand may be a column slice of a wider matrix (x[:, 512:602] of a 608-wide allocation), and a pitch counted from the slice
    // pointer would reach k0 * 2 bytes past the allocation in the matrix's last row.  Rows before the last may still
    // be read past their valid columns -- into their own padding / the next row: finite garbage that only feeds outputs nobody reads.
    const uint32_t l_valid_b = (uint32_t)((((l_second ? a.K[1] : a.K[0]) * 2 + 15) / 16) * 16);
    const uint32_t g_valid_b = (uint32_t)(((a.N * 2 + 15) / 16) * 16);
    const gw_rsrc_t rx = gw_rsrc(lX + r_begin * l_lda_b,
                                 l_has_x ? (uint64_t)(r_end - r_begin - 1) * l_lda_b + (l_valid_b < l_lda_b ? l_valid_b : l_lda_b) : 0);
    const gw_rsrc_t rg = gw_rsrc(static_cast<const char*>(a.G) + r_begin * ldg_b,
                                 (uint64_t)(r_end - r_begin - 1) * ldg_b + (g_valid_b < ldg_b ? g_valid_b : ldg_b));
    const int lx_row = 8 * (wave & 1) + (lane >> 3);
    uint32_t lx_off = (uint32_t)lx_row * l_lda_b + (uint32_t)(l_kcol0 * 2 + (lane & 7) * 16);
    const int lg_row = 2 * wave + half;
    uint32_t lg_off = (uint32_t)lg_row * ldg_b + (uint32_t)(l32 * 16);
    const int lx_lds = (wave >> 1) * 2048 + lx_row * 128 + (lane & 7) * 16;
    const int lg_lds = 16 * 512 + lg_row * 512 + l32 * 16;
    // (a 16-byte load that straddles the end of a row's valid columns reads the padding / the next row: those columns only
    //  feed outputs the reduce kernel never reads; past the slab's last row the descriptor returns zeros)
    auto fetch = [&](GwStage& st) {
#pragma unroll
        for (int u = 0; u < kGwU; ++u) {
            st.x[u] = __builtin_bit_cast(gw_u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)lx_off, 0, 0));
            st.g[u] = __builtin_bit_cast(gw_u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rg, (int)lg_off, 0, 0));
            lx_off += 16 * l_lda_b;
            lg_off += 16 * ldg_b;
        }
    };
    auto park = [&](const GwStage& st, int buf) {
#pragma unroll
        for (int u = 0; u < kGwU; ++u) {
            *reinterpret_cast<gw_u32x4_t*>(lds + buf * kGwTile + u * kGwSub + lx_lds) = st.x[u];
            *reinterpret_cast<gw_u32x4_t*>(lds + buf * kGwTile + u * kGwSub + lg_lds) = st.g[u];
        }
    };

    // ---- compute role: wave w owns column slab w / 2 (64 columns of [X1 | X2]) x half w % 2 of G's columns
    const int ks = lks, nh = wave & 1;
    const bool active = ks < ks_total && nh * 128 < a.N;
    const int cx_lds = (wave >> 1) * 2048 + (8 * half) * 128 + l32 * 4;
    const int cg_lds = 16 * 512 + (8 * half) * 512 + nh * 256 + l32 * 4;

    gw_f32x16_t acc[2][2][2];                                 // [X parity][G 64-column slab][G parity]
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i >> 2][(i >> 1) & 1][i & 1][r] = 0.0f;

    auto compute = [&](int buf) {
        if (!active) return;                                  // wave-uniform
#pragma unroll
      for (int u = 0; u < kGwU; ++u) {
        const char* t = lds + buf * kGwTile + u * kGwSub;
        uint32_t ra[8], rb0[8], rb1[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            ra[j] = *reinterpret_cast<const uint32_t*>(t + cx_lds + j * 128);
            rb0[j] = *reinterpret_cast<const uint32_t*>(t + cg_lds + j * 512);
            rb1[j] = *reinterpret_cast<const uint32_t*>(t + cg_lds + j * 512 + 128);
        }
        gw_bf16x8_t xa[2], gb[2][2];
        gw_split(ra, xa[0], xa[1]);
        gw_split(rb0, gb[0][0], gb[0][1]);
        gw_split(rb1, gb[1][0], gb[1][1]);
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int ns = 0; ns < 2; ++ns)
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    acc[p][ns][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[p], gb[ns][q], acc[p][ns][q], 0, 0, 0);
      }
    };

    // two register stages: the loads of step s + 2 are issued while step s computes; a stage is parked one step later
    GwStage S0, S1;
    fetch(S0);                                                // step 0
    fetch(S1);                                                // step 1 (past the slab: zeros)
    park(S0, 0);
    __syncthreads();
    for (int s = 0; s < steps; s += 2) {
        fetch(S0);                                            // step s + 2
        compute(0);                                           // step s
        park(S1, 1);                                          // step s + 1
        __syncthreads();
        fetch(S1);                                            // step s + 3
        compute(1);                                           // step s + 1 (zeros if steps is odd)
        park(S0, 0);                                          // step s + 2
        __syncthreads();
    }
    if (!active) return;

    // D[i][j]: j = lane % 32 -> G column, i = (r & 3) + 8 (r >> 2) + 4 half -> X column, both inside their parity class
    float* P = a.partial + ((int64_t)slab * ks_total * 64 + (int64_t)ks * 64) * 256 + nh * 128;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int ns = 0; ns < 2; ++ns)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kc = 2 * ((r & 3) + 8 * (r >> 2) + 4 * half) + p;
                // the two parities of a lane's G columns are adjacent in memory: one 8-byte store
                float2 v = make_float2(acc[p][ns][0][r], acc[p][ns][1][r]);
                *reinterpret_cast<float2*>(P + (int64_t)kc * 256 + ns * 64 + 2 * l32) = v;
            }
}

// dW[k][n] = sum over slabs in a FIXED order (deterministic): a 256-thread block owns 64 consecutive columns of one row of
// [X1 | X2]^T . G; its four waves each add every fourth slab in slab order, then the four partial sums are added in wave order.
__global__ __launch_bounds__(256) void gradw_reduce_kernel(const float* __restrict__ partial, int n_slabs, int kc_total, int kslabs0,
                                                           int K0, int K1, int N, float* __restrict__ dW0, int64_t ld0,
                                                           float* __restrict__ dW1, int64_t ld1, int transposed) {
    __shared__ float part[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int kc = blockIdx.x >> 2, n = (blockIdx.x & 3) * 64 + lane;
    const bool second = kc >= kslabs0 * 64;
    const int k = second ? kc - kslabs0 * 64 : kc;
    if (k >= (second ? K1 : K0) || (blockIdx.x & 3) * 64 >= N) return;         // block-uniform
    float s = 0.0f;
    if (n < N) {
        const float* p = partial + (int64_t)kc * 256 + n;
        const int64_t stride = (int64_t)kc_total * 256;
        for (int t = w; t < n_slabs; t += 4) s += p[t * stride];
    }
    part[w][lane] = s;
    __syncthreads();
    if (w == 0 && n < N) {
        const float v = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
        // transposed: the caller wants G^T . X (its [N, K] destination) -- the products of one X with two G's (below)
        float* d = second ? dW1 : dW0;
        const int64_t ld = second ? ld1 : ld0;
        d[transposed ? (int64_t)n * ld + k : (int64_t)k * ld + n] = v;
    }
}

}  // namespace dgll

using namespace dgll;

// Workspace the caller provides for n_slabs slab partials: dgll_hip_grad_weight_workspace(K1, K2, n_slabs) bytes.
DGLL_API int64_t dgll_hip_grad_weight_workspace(int K1, int K2, int n_slabs) {
    const int64_t ks = (K1 + 63) / 64 + (K2 + 63) / 64;
    return (int64_t)n_slabs * ks * 64 * 256 * (int64_t)sizeof(float);
}

static int grad_weight_bf16_impl(void* stream, const void* X1, int64_t ldx1, int K1, const void* X2, int64_t ldx2, int K2,
                                 const void* G, int64_t ldg, int N, int64_t M, void* workspace, int64_t workspace_bytes,
                                 int n_slabs, float* dW1, int64_t lddw1, float* dW2, int64_t lddw2, bool transposed) {
    DGLL_REQUIRE(M >= 0 && N >= 0 && K1 >= 0 && K2 >= 0, "negative size");
    DGLL_REQUIRE(dW1 && K1 > 0 && N > 0 && (M == 0 || (X1 && G)), "NULL operand");    // an empty reduction may come with NULL inputs
    DGLL_REQUIRE(K1 <= 256 && K2 <= 256 && N <= 256, "dgll_hip_grad_weight_bf16 handles K1, K2, N <= 256");
    DGLL_REQUIRE((M == 0 || (K2 == 0) == (X2 == nullptr)) && (K2 == 0 || dW2), "second operand incomplete");
    DGLL_REQUIRE(n_slabs >= 1 && n_slabs <= 4096, "n_slabs");
    DGLL_REQUIRE(workspace && workspace_bytes >= dgll_hip_grad_weight_workspace(K1, K2, n_slabs), "workspace too small");
    if (M == 0) {                                             // empty reduction: zeros
        hipStream_t s = static_cast<hipStream_t>(stream);
        const int r1 = transposed ? N : K1, r2 = transposed ? (K2 ? N : 0) : K2;
        for (int k = 0; k < r1; ++k) DGLL_HIP_TRY(hipMemsetAsync(dW1 + (int64_t)k * lddw1, 0, (size_t)(transposed ? K1 : N) * sizeof(float), s));
        for (int k = 0; k < r2; ++k) DGLL_HIP_TRY(hipMemsetAsync(dW2 + (int64_t)k * lddw2, 0, (size_t)(transposed ? K2 : N) * sizeof(float), s));
        return DGLL_OK;
    }
    // 16-byte loads: rows start on 16-byte boundaries
    DGLL_REQUIRE(aligned16(X1) && (ldx1 & 7) == 0 && ldx1 >= K1, "X1: 16-byte aligned rows (leading dimension a multiple of 8)");
    DGLL_REQUIRE(!X2 || (aligned16(X2) && (ldx2 & 7) == 0 && ldx2 >= K2), "X2: 16-byte aligned rows (leading dimension a multiple of 8)");
    DGLL_REQUIRE(aligned16(G) && (ldg & 7) == 0 && ldg >= N, "G: 16-byte aligned rows (leading dimension a multiple of 8)");
    DGLL_REQUIRE(transposed ? (lddw1 >= K1 && (!X2 || lddw2 >= K2)) : (lddw1 >= N && (!X2 || lddw2 >= N)), "dW leading dimension");
    hipStream_t s = static_cast<hipStream_t>(stream);
    GradWArgs a{};
    a.X[0] = X1; a.ldx[0] = ldx1; a.K[0] = K1;
    a.X[1] = X2; a.ldx[1] = ldx2; a.K[1] = K2;
    a.G = G; a.ldg = ldg; a.N = N; a.M = M;
    a.kslabs[0] = (K1 + 63) / 64; a.kslabs[1] = (K2 + 63) / 64;
    a.n_slabs = n_slabs;
    const int64_t per = (M + n_slabs - 1) / n_slabs;
    a.rows_per_slab = std::max<int64_t>(16 * kGwU, (per + 16 * kGwU - 1) / (16 * kGwU) * (16 * kGwU));
    // the loader's byte offsets inside a slab are 32-bit and run up to three prefetch steps (16 * kGwU rows each) past the
    // slab's end before the out-of-range test zeroes them: that overshoot must not wrap either
    DGLL_REQUIRE((a.rows_per_slab + 4 * 16 * kGwU) * std::max(std::max(ldx1, ldx2), ldg) * 2 < (int64_t)1 << 32,
                 "slab (plus the prefetch overshoot) larger than 4 GiB: use more slabs");
    a.partial = static_cast<float*>(workspace);
    const int ks_total = a.kslabs[0] + a.kslabs[1];
    // slabs past the end of M (tiny M) would leave their partials unwritten: shrink the slab count instead
    const int used = (int)((M + a.rows_per_slab - 1) / a.rows_per_slab);
    a.n_slabs = used;
    // waves idle in the upper half of G's columns (N <= 128) leave their partial columns unwritten: the reduce kernel never reads n >= N
    dim3 grid((uint32_t)used, (uint32_t)((ks_total + 3) / 4));
    // the attribute is per device: raise it once for each device a launch is issued on (single-process multi-device use)
    static bool raised_on[64] = {};
    int dev = 0;
    DGLL_HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !raised_on[dev]) {
        DGLL_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&gradw_splitk_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kGwTile));
        if (dev >= 0 && dev < 64) raised_on[dev] = true;
    }
    hipLaunchKernelGGL(gradw_splitk_kernel, grid, dim3(512), 2 * kGwTile, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "gradw_splitk_kernel launch");
    hipLaunchKernelGGL(gradw_reduce_kernel, dim3((uint32_t)(ks_total * 64 * 4)), dim3(256), 0, s, a.partial, used, ks_total * 64,
                       a.kslabs[0], K1, K2, N, dW1, lddw1, dW2, lddw2, transposed ? 1 : 0);
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "gradw_reduce_kernel launch");
    return DGLL_OK;
}

DGLL_API int dgll_hip_grad_weight_bf16(void* stream, const void* X1, int64_t ldx1, int K1, const void* X2, int64_t ldx2, int K2,
                                       const void* G, int64_t ldg, int N, int64_t M, void* workspace, int64_t workspace_bytes,
                                       int n_slabs, float* dW1, int64_t lddw1, float* dW2, int64_t lddw2) {
    return grad_weight_bf16_impl(stream, X1, ldx1, K1, X2, ldx2, K2, G, ldg, N, M, workspace, workspace_bytes, n_slabs, dW1, lddw1,
                                 dW2, lddw2, false);
}

// The same launch with the outputs stored transposed: dW1 [N, lddw1] = G^T . X1 and dW2 [N, lddw2] = G^T . X2.  This is the pair
// "one X, two gradients" of the narrowing SAGE layer (out = h.Ws + A(h.Wn): dWs = h^T . g and dWn = h^T . (A^T g) share h): call it
// with X1 = g, X2 = A^T g, G = h and the wide operand h is read once for both products instead of once per product.
DGLL_API int dgll_hip_grad_weight_bf16_tr(void* stream, const void* X1, int64_t ldx1, int K1, const void* X2, int64_t ldx2, int K2,
                                          const void* G, int64_t ldg, int N, int64_t M, void* workspace, int64_t workspace_bytes,
                                          int n_slabs, float* dW1, int64_t lddw1, float* dW2, int64_t lddw2) {
    return grad_weight_bf16_impl(stream, X1, ldx1, K1, X2, ldx2, K2, G, ldg, N, M, workspace, workspace_bytes, n_slabs, dW1, lddw1,
                                 dW2, lddw2, true);
}
