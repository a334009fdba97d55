// common.hpp -- shared device/host helpers of libdgll_hip.so (gfx950 only, wave64).
#pragma once
#include <hip/hip_runtime.h>

#include "host_common.hpp"

// Load-balancing schedule of one CSR structure (opaque to callers; include/dgll_hip.h).  Rows longer than `threshold`
// nonzeros are cut into chunks that run as independent work items and are reduced in a fixed order afterwards.
struct dgll_csr_plan {
    int device = 0;
    int64_t n_rows = 0, nnz = 0;
    int threshold = 256;
    int64_t n_long = 0, n_chunks = 0;
    int64_t* d_long_row = nullptr;    // [n_long]      row id of each long row (ascending)
    int32_t* d_long_chunk0 = nullptr; // [n_long + 1]  first chunk of each long row
    int64_t* d_chunk_begin = nullptr; // [n_chunks]    first edge of the chunk
    int64_t* d_chunk_end = nullptr;   // [n_chunks]    one past its last edge
    int64_t* d_chunk_row = nullptr;   // [n_chunks]    row the chunk belongs to
    // cost-balanced wave schedule of the flattened kernel (spmm_csr_flat_kernel): wave w owns rows [d_flat_row0[w], d_flat_row0[w + 1]),
    // about flat_edges edges (a row is charged like 4 edges); nullptr: no such schedule (host-only plan)
    int flat_edges = 0;
    int64_t n_flat = 0;
    int64_t* d_flat_row0 = nullptr;   // [n_flat + 1]
};

namespace dgll {

constexpr int kWave = 64;           // CDNA4 wavefront width
constexpr int kBlock = 256;         // 4 waves per workgroup, one per SIMD
constexpr int kWavesPerBlock = kBlock / kWave;
constexpr int kXcds = 8;            // MI355X: 8 XCDs, block b is dispatched to XCD b % 8

typedef uint16_t bf16_t;            // raw bfloat16 bits
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;

// ---- error plumbing (set_error / DGLL_REQUIRE: host_common.hpp) ---------------------------------------
int hip_fail(hipError_t e, const char* what);

#define DGLL_HIP_TRY(expr)                                          \
    do {                                                            \
        hipError_t _e = (expr);                                     \
        if (_e != hipSuccess) return ::dgll::hip_fail(_e, #expr);   \
    } while (0)

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- bf16 <-> f32 ------------------------------------------------------------------------------------
__device__ __forceinline__ float bf16_lo(uint32_t packed) { return __uint_as_float(packed << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t packed) { return __uint_as_float(packed & 0xffff0000u); }
__device__ __forceinline__ float bf16_to_f32(bf16_t b) { return __uint_as_float((uint32_t)b << 16); }
// round-to-nearest-even pack of two floats: one v_cvt_pk_bf16_f32 on gfx950
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return (bf16_t)(pack_bf16x2(f, 0.0f) & 0xffffu); }

// make a wave-uniform 64-bit value provably uniform (SGPR pair) for the compiler
__device__ __forceinline__ int64_t uniform64(int64_t v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & (kWave - 1)); }

// XCD-aware block remap: give every XCD (private 4 MiB L2) a CONTIGUOUS range of logical blocks, so rows
// that are close in the graph (and tend to share neighbours) share an L2.  Bijective for any grid size.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t bid, uint32_t nblocks) {
    const uint32_t q = nblocks / kXcds, r = nblocks % kXcds;   // XCD x owns q (+1 if x < r) logical blocks
    const uint32_t xcd = bid % kXcds, idx = bid / kXcds;
    const uint32_t start = xcd * q + (xcd < r ? xcd : r);
    return start + idx;
}

// ---- typed 16-byte / scalar row-vector access ----------------------------------------------------------
// VecIO<T, EPV>: EPV elements of storage type T handled by one lane.  EPV * sizeof(T) is 16 bytes on the
// fast path (global_load_dwordx4) and sizeof(T) on the any-alignment path.
template <typename T, int EPV> struct VecIO;

// streaming forms of a 16-byte access: data that is touched once by a launch (a row of its own operand or result, as opposed
// to the gathered rows that are re-read many times) should not displace the gathered rows in L2
__device__ __forceinline__ uint4 load16_nt(const void* p) {
    typedef __attribute__((ext_vector_type(4))) unsigned int u4;
    const u4 v = __builtin_nontemporal_load(reinterpret_cast<const u4*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void store16_nt(void* p, uint4 d) {
    typedef __attribute__((ext_vector_type(4))) unsigned int u4;
    const u4 v = {d.x, d.y, d.z, d.w};
    __builtin_nontemporal_store(v, reinterpret_cast<u4*>(p));
}

template <> struct VecIO<float, 4> {
    typedef uint4 raw_t;
    static __device__ __forceinline__ raw_t zero() { return make_uint4(0, 0, 0, 0); }
    static __device__ __forceinline__ raw_t load(const float* p) { return *reinterpret_cast<const uint4*>(p); }
    static __device__ __forceinline__ raw_t load_nt(const float* p) { return load16_nt(p); }
    static __device__ __forceinline__ void store_nt(float* p, const float (&f)[4]) {
        store16_nt(p, make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3])));
    }
    static __device__ __forceinline__ void unpack(const raw_t& r, float (&f)[4]) {
        f[0] = __uint_as_float(r.x); f[1] = __uint_as_float(r.y); f[2] = __uint_as_float(r.z); f[3] = __uint_as_float(r.w);
    }
    static __device__ __forceinline__ void store(float* p, const float (&f)[4]) {
        *reinterpret_cast<float4*>(p) = make_float4(f[0], f[1], f[2], f[3]);
    }
};
template <> struct VecIO<float, 1> {
    typedef float raw_t;
    static __device__ __forceinline__ raw_t zero() { return 0.0f; }
    static __device__ __forceinline__ raw_t load(const float* p) { return *p; }
    static __device__ __forceinline__ void unpack(const raw_t& r, float (&f)[1]) { f[0] = r; }
    static __device__ __forceinline__ void store(float* p, const float (&f)[1]) { *p = f[0]; }
};
template <> struct VecIO<bf16_t, 8> {
    typedef uint4 raw_t;
    static __device__ __forceinline__ raw_t zero() { return make_uint4(0, 0, 0, 0); }
    static __device__ __forceinline__ raw_t load(const bf16_t* p) { return *reinterpret_cast<const uint4*>(p); }
    static __device__ __forceinline__ raw_t load_nt(const bf16_t* p) { return load16_nt(p); }
    static __device__ __forceinline__ void unpack(const raw_t& r, float (&f)[8]) {
        f[0] = bf16_lo(r.x); f[1] = bf16_hi(r.x); f[2] = bf16_lo(r.y); f[3] = bf16_hi(r.y);
        f[4] = bf16_lo(r.z); f[5] = bf16_hi(r.z); f[6] = bf16_lo(r.w); f[7] = bf16_hi(r.w);
    }
    static __device__ __forceinline__ void store(bf16_t* p, const float (&f)[8]) {
        *reinterpret_cast<uint4*>(p) = make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]),
                                                  pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7]));
    }
    // streaming form: an output row is written once and not read again by this launch -- keep it from displacing the
    // gathered feature rows in L2
    static __device__ __forceinline__ void store_nt(bf16_t* p, const float (&f)[8]) {
        typedef __attribute__((ext_vector_type(4))) unsigned int u4;
        const u4 v = {pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7])};
        __builtin_nontemporal_store(v, reinterpret_cast<u4*>(p));
    }
};
template <> struct VecIO<bf16_t, 1> {
    typedef bf16_t raw_t;
    static __device__ __forceinline__ raw_t zero() { return 0; }
    static __device__ __forceinline__ raw_t load(const bf16_t* p) { return *p; }
    static __device__ __forceinline__ void unpack(const raw_t& r, float (&f)[1]) { f[0] = bf16_to_f32(r); }
    static __device__ __forceinline__ void store(bf16_t* p, const float (&f)[1]) { *p = f32_to_bf16(f[0]); }
};
// wide stores used when the output type differs from the gather type
template <> struct VecIO<float, 8> {
    static __device__ __forceinline__ void store(float* p, const float (&f)[8]) {
        reinterpret_cast<float4*>(p)[0] = make_float4(f[0], f[1], f[2], f[3]);
        reinterpret_cast<float4*>(p)[1] = make_float4(f[4], f[5], f[6], f[7]);
    }
};
template <> struct VecIO<bf16_t, 4> {
    static __device__ __forceinline__ void store(bf16_t* p, const float (&f)[4]) {
        *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]));
    }
};

template <typename T> __device__ __forceinline__ void store_one(T* p, float v);
template <> __device__ __forceinline__ void store_one<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void store_one<bf16_t>(bf16_t* p, float v) { *p = f32_to_bf16(v); }

}  // namespace dgll
