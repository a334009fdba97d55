// Probe: the ceiling of the path the F = 256 SpMM runs on.  A gather of 512-byte rows (32 lanes x 16 B per row, two row slots per
// wavefront, U = 4 loads in flight per lane, indices handed out from a coalesced 64-entry batch -- the shape and occupancy of
// spmm_csr_kernel<bf16, bf16, 8, 32, ...>) from tables that are L2-resident (2 MB), spread over the eight L2s (24 MB),
// Infinity-Cache-resident (128 MB of the 256 MB MALL) and HBM-resident (1.25 GB = the products-sized feature matrix at F = 256),
// with uniform-random and hub-skewed indices.  Prints TB/s of gathered rows; run under `rocprofv3 --pmc FETCH_SIZE` for the
// bytes the L2s requested from the fabric per launch.
//   hipcc --offload-arch=gfx950 -O3 -o gather_mall.bin gather_mall.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kDeg = 50;            // edges per output row (the bench graph's average degree)

__device__ __forceinline__ float lo(uint32_t p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float hi(uint32_t p) { return __uint_as_float(p & 0xffff0000u); }

template <int U>
__global__ __launch_bounds__(256) void gather_rows(const uint4* __restrict__ table, const int32_t* __restrict__ idx, int64_t n_rows,
                                                   uint4* __restrict__ out, int rows_per_wave) {
    const int lane = threadIdx.x & 63, slot = lane >> 5, l32 = lane & 31;
    const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    for (int rr = 0; rr < rows_per_wave; ++rr) {
        const int64_t row = wave * rows_per_wave + rr;
        if (row >= n_rows) return;
        const int32_t* e = idx + row * kDeg;
        const int mine = __builtin_nontemporal_load(e + (lane < kDeg ? lane : kDeg - 1));     // one coalesced batch (kDeg <= 64)
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int k = 0; k < kDeg; k += 2 * U) {
            uint4 v[U];
            bool ok[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = k + 2 * u + slot;
                ok[u] = j < kDeg;
                const int id = __shfl(mine, ok[u] ? j : kDeg - 1);
                v[u] = table[(int64_t)id * 32 + l32];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float m = ok[u] ? 1.0f : 0.0f;
                acc[0] += m * lo(v[u].x); acc[1] += m * hi(v[u].x); acc[2] += m * lo(v[u].y); acc[3] += m * hi(v[u].y);
                acc[4] += m * lo(v[u].z); acc[5] += m * hi(v[u].z); acc[6] += m * lo(v[u].w); acc[7] += m * hi(v[u].w);
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] += __shfl_xor(acc[i], 32);
        if (slot == 0) {
            uint4 o;
            o.x = (__float_as_uint(acc[0]) >> 16) | (__float_as_uint(acc[1]) & 0xffff0000u);
            o.y = (__float_as_uint(acc[2]) >> 16) | (__float_as_uint(acc[3]) & 0xffff0000u);
            o.z = (__float_as_uint(acc[4]) >> 16) | (__float_as_uint(acc[5]) & 0xffff0000u);
            o.w = (__float_as_uint(acc[6]) >> 16) | (__float_as_uint(acc[7]) & 0xffff0000u);
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            u32x4 ov = {o.x, o.y, o.z, o.w};
            __builtin_nontemporal_store(ov, reinterpret_cast<u32x4*>(out + row * 32 + l32));
        }
    }
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static inline uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

int main(int argc, char** argv) {
    const int64_t n_rows = argc > 1 ? atoll(argv[1]) : 1280000;       // 64 M edges = 32.8 GB of gathered rows per launch
    const int64_t n_edges = n_rows * kDeg;
    int32_t* d_idx; uint4* d_out;
    (void)hipMalloc(&d_idx, n_edges * 4);
    (void)hipMalloc(&d_out, n_rows * 512);
    std::vector<int32_t> h((size_t)n_edges);
    printf("gather of 512-byte rows, %lld rows x %d edges = %.1f M edges (%.1f GB gathered per launch), U = 4 / 8, 8 waves per SIMD wanted\n",
           (long long)n_rows, kDeg, n_edges / 1e6, n_edges * 512.0 / 1e9);
    const double sizes_mb[] = {2.0, 24.0, 128.0, 1250.0};
    for (double mb : sizes_mb) {
        const int64_t t_rows = (int64_t)(mb * 1e6 / 512);
        uint4* d_table; (void)hipMalloc(&d_table, t_rows * 512); (void)hipMemset(d_table, 0x3c, t_rows * 512);
        for (int dist = 0; dist < 2; ++dist) {
            for (int64_t i = 0; i < n_edges; ++i) {
                const double u = (double)(rnd() >> 11) / 9007199254740992.0;
                h[(size_t)i] = (int32_t)((dist == 0 ? u : u * u * u * u) * (double)t_rows);      // skew: a tenth of the rows draw 56 % of the edges
                if (h[(size_t)i] >= t_rows) h[(size_t)i] = (int32_t)t_rows - 1;
            }
            (void)hipMemcpy(d_idx, h.data(), n_edges * 4, hipMemcpyHostToDevice);
            for (int U : {4, 8}) {
                const int rpw = 8;
                const int64_t waves = (n_rows + rpw - 1) / rpw;
                const dim3 grid((uint32_t)((waves + 3) / 4));
                hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
                auto launch = [&]() {
                    if (U == 4) gather_rows<4><<<grid, 256>>>(d_table, d_idx, n_rows, d_out, rpw);
                    else gather_rows<8><<<grid, 256>>>(d_table, d_idx, n_rows, d_out, rpw);
                };
                launch();
                (void)hipEventRecord(a);
                for (int r = 0; r < 3; ++r) launch();
                (void)hipEventRecord(b); (void)hipEventSynchronize(b);
                float ms; (void)hipEventElapsedTime(&ms, a, b);
                ms /= 3;
                printf("table %7.1f MB  %-12s U=%d : %.3f ms  %.2f TB/s gathered  (+ %.2f GB written, %.2f GB of indices)\n", mb,
                       dist == 0 ? "uniform" : "hub-skewed", U, ms, n_edges * 512.0 / 1e9 / ms, n_rows * 512.0 / 1e9, n_edges * 4.0 / 1e9);
            }
        }
        (void)hipFree(d_table);
    }
    return 0;
}
