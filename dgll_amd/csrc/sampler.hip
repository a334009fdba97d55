// sampler.hip -- native (host) neighbour sampler that is BIT-EXACT with the reference's Python sampler.
//
// The reference samples with the interpreter's global Mersenne Twister: for every seed node, in order,
// `random.sample(neighbors, fanout)` when the node has more than `fanout` neighbours, else all of them
// (/root/reference/dgll/sampling/base_sampler.py:45-58).  BASELINE's north star asks for sampled node/edge IDs that are
// bit-identical to the reference under `random.seed(s)`.  This file restates, in C++, exactly what CPython 3.10's
// random.sample does with the generator so that the SAME ids come out ~100x faster:
//   * MT19937 genrand_uint32 (the generator behind random.getrandbits);
//   * getrandbits(k) = genrand_uint32() >> (32 - k) for k <= 32;
//   * _randbelow_with_getrandbits(n): k = n.bit_length(); draw until r < n;
//   * sample(): n <= setsize -> "pool" algorithm (swap the drawn slot with the last live slot), else rejection
//     sampling against the set of already selected POSITIONS; setsize = 21 (+ 4**ceil(log(3k, 4)) for k > 5) is
//     computed by the Python caller with the interpreter's own math so no floating-point corner can differ.
// The generator state is imported from `random.getstate()` and exported back, so Python code before and after a call
// sees the stream exactly as if the reference's pure-Python loop had run.  (The algorithm is CPython-version
// specific; tests/test_sampler.py checks it against the running interpreter and against the goldens.)
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "host_common.hpp"   // host code only: this file also builds with g++ -fsanitize=thread (tests/c_abi/Makefile)

namespace {

struct MT19937 {
    uint32_t* mt;   // 624 words, caller-owned (CPython's untempered state vector)
    int idx;
    uint32_t out[624];   // the tempered outputs of the current block: next() is a load; twist + tempering run once per
                         // 624 draws in loops the compiler vectorises
    bool fresh = false;  // `out` matches `mt`
    void temper() {
        for (int i = 0; i < 624; ++i) {
            uint32_t y = mt[i];
            y ^= (y >> 11);
            y ^= (y << 7) & 0x9d2c5680U;
            y ^= (y << 15) & 0xefc60000U;
            y ^= (y >> 18);
            out[i] = y;
        }
        fresh = true;
    }
    void twist() {
        constexpr int N = 624, M = 397;
        constexpr uint32_t MATRIX_A = 0x9908b0dfU, UPPER = 0x80000000U, LOWER = 0x7fffffffU;
        int kk;
        for (kk = 0; kk < N - M; kk++) {
            uint32_t y = (mt[kk] & UPPER) | (mt[kk + 1] & LOWER);
            mt[kk] = mt[kk + M] ^ (y >> 1) ^ ((0U - (y & 1U)) & MATRIX_A);
        }
        for (; kk < N - 1; kk++) {
            uint32_t y = (mt[kk] & UPPER) | (mt[kk + 1] & LOWER);
            mt[kk] = mt[kk + (M - N)] ^ (y >> 1) ^ ((0U - (y & 1U)) & MATRIX_A);
        }
        uint32_t y = (mt[N - 1] & UPPER) | (mt[0] & LOWER);
        mt[N - 1] = mt[M - 1] ^ (y >> 1) ^ ((0U - (y & 1U)) & MATRIX_A);
    }
    inline uint32_t next() {
        if (__builtin_expect(idx >= 624, 0)) { twist(); temper(); idx = 0; }
        else if (__builtin_expect(!fresh, 0)) temper();
        return out[idx++];
    }
    // random._randbelow_with_getrandbits for 0 < n < 2**32
    inline uint32_t randbelow(uint32_t n) {
        const int sh = __builtin_clz(n);              // 32 - n.bit_length()
        uint32_t r = next() >> sh;
        while (r >= n) r = next() >> sh;
        return r;
    }
};

}  // namespace

// One hop of base_sampler.py:45-58 for `n_seeds` seeds over a CSR copy of DGraph.edges (indptr/indices).
// fanout < 0 means None (take every neighbour).  out_src/out_dst need sum(min(deg, fanout)) slots (`capacity`);
// out_counts[n_seeds] receives the number of neighbours kept per seed occurrence.
//
// Two phases.  (1) SEQUENTIAL, in seed order: consume the generator exactly as random.sample does and record, for every
// kept neighbour, its POSITION in the seed's adjacency list -- the pool algorithm is run on positions (pool[j] starts
// as j), which selects the same elements as running it on the values.  This phase touches only the generator, the
// degree of each seed and a small scratch array, so it is cache-resident.  (2) PARALLEL (std::thread): translate positions to
// neighbour ids, the memory-bound part (random reads into a multi-hundred-megabyte index array).
namespace {
int sample_hop(MT19937& rng, const int64_t* indptr, const int64_t* indices, const int64_t* seeds, int64_t n_seeds, int64_t fanout,
               int64_t setsize, int64_t* out_src, int64_t* out_dst, int64_t* out_counts, int64_t capacity, int64_t* n_out,
               int64_t max_threads) {
    static const bool profile = std::getenv("DGLL_SAMPLER_PROFILE") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    std::vector<uint32_t> pool;      // pool algorithm, on positions: kept equal to the identity between seeds (only the
                                     // entries a sample overwrote are restored, not all n re-initialised)
    std::vector<uint32_t> touched;   // pool positions a sample overwrote
    std::vector<uint32_t> stamp;     // rejection branch: position j is selected iff stamp[j] == epoch (no per-seed clear)
    uint32_t epoch = 0;
    std::vector<int64_t> offset((size_t)n_seeds + 1, 0);
    // ---- phase 1: positions (stored in out_src for now)
    int64_t at = 0;
    for (int64_t s = 0; s < n_seeds; ++s) {
        const int64_t v = seeds[s];
        const int64_t n = indptr[v + 1] - indptr[v];
        const int64_t take = (fanout < 0 || n <= fanout) ? n : fanout;
        if (at + take > capacity) {
            dgll::set_error("sampler output capacity exceeded");
            return DGLL_ERR_WORKSPACE;
        }
        if (take == n) {                                  // all neighbours, no draw (base_sampler.py:49-54)
            for (int64_t i = 0; i < n; ++i) out_src[at + i] = i;
        } else if (n <= setsize) {                        // pool algorithm
            DGLL_REQUIRE(n < (int64_t)0x7fffffff, "degree too large");
            // Branch-free form of `j = randbelow(n - i)` + swap: every generator output is examined once; a rejected
            // one (r >= n - i) performs the same loads and stores with no effect and does not advance i.  The retry loop
            // of randbelow mispredicts up to every second output, which cost more than the arithmetic.  The pool is kept
            // equal to the identity between seeds and is long enough for any k-bit output.
            const size_t need = (size_t)1 << (32 - __builtin_clz((uint32_t)n));
            if (need > pool.size()) {
                const size_t old = pool.size();
                pool.resize(need);
                for (size_t i = old; i < need; ++i) pool[i] = (uint32_t)i;
            }
            if ((size_t)take > touched.size()) touched.resize((size_t)take);
            uint32_t* pl = pool.data();
            int64_t i = 0;
            while (i < take) {
                const uint32_t m = (uint32_t)(n - i);
                const uint32_t r = rng.next() >> __builtin_clz(m);
                const bool ok = r < m;
                const uint32_t pr = pl[r], last = pl[m - 1];
                out_src[at + i] = pr;
                pl[r] = ok ? last : pr;
                touched[(size_t)i] = r;
                i += ok;
            }
            for (int64_t t = 0; t < take; ++t) pl[touched[(size_t)t]] = touched[(size_t)t];
        } else {                                          // rejection against the selected positions
            DGLL_REQUIRE(n < (int64_t)0x7fffffff, "degree too large");
            const size_t need = (size_t)1 << (32 - __builtin_clz((uint32_t)n));     // any k-bit output indexes the stamps
            if (need > stamp.size()) stamp.resize(need, 0);
            if (++epoch == 0) { std::fill(stamp.begin(), stamp.end(), 0u); epoch = 1; }
            const int sh = __builtin_clz((uint32_t)n);
            uint32_t* st = stamp.data();
            int64_t i = 0;
            while (i < take) {                            // same idea: one pass over the outputs, no data-dependent branch
                const uint32_t r = rng.next() >> sh;
                const bool ok = (r < (uint32_t)n) & (st[r] != epoch);
                out_src[at + i] = r;
                st[r] = epoch;                            // r >= n: never consulted; r < n: selected now or before
                i += ok;
            }
        }
        out_counts[s] = take;
        offset[s] = at;
        at += take;
    }
    offset[n_seeds] = at;
    *n_out = at;
    const auto t_phase1 = std::chrono::steady_clock::now();
    if (!out_dst) {   // positions only: the caller translates later (dgll_host_translate_neighbors), e.g. on another thread
        if (profile)
            std::fprintf(stderr, "[dgll sampler] %lld seeds -> %lld edges: sequential draw phase %.2f ms, translation deferred\n",
                         (long long)n_seeds, (long long)at, std::chrono::duration<double, std::milli>(t_phase1 - t_begin).count());
        return DGLL_OK;
    }
    // ---- phase 2: positions -> neighbour ids, destination ids
    auto translate = [&](int64_t s0, int64_t s1) {
        for (int64_t s = s0; s < s1; ++s) {
            const int64_t v = seeds[s];
            const int64_t* nb = indices + indptr[v];
            for (int64_t k = offset[s]; k < offset[s + 1]; ++k) {
                out_src[k] = nb[out_src[k]];
                out_dst[k] = v;
            }
        }
    };
    const unsigned hw = std::thread::hardware_concurrency();
    const int64_t n_threads = at < (1 << 16) ? 1 : std::min<int64_t>(std::min<int64_t>(hw ? hw : 1, 16), std::max<int64_t>(max_threads, 1));
    if (n_threads <= 1) {
        translate(0, n_seeds);
    } else {   // plain std::thread (no OpenMP runtime next to torch's): contiguous seed ranges, disjoint output ranges
        std::vector<std::thread> workers;
        for (int64_t t = 0; t < n_threads; ++t)
            workers.emplace_back(translate, n_seeds * t / n_threads, n_seeds * (t + 1) / n_threads);
        for (auto& w : workers) w.join();
    }
    if (profile) {
        const auto t_end = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[dgll sampler] %lld seeds -> %lld edges: sequential draw phase %.2f ms, translation (%lld threads) %.2f ms\n",
                     (long long)n_seeds, (long long)at, std::chrono::duration<double, std::milli>(t_phase1 - t_begin).count(),
                     (long long)n_threads, std::chrono::duration<double, std::milli>(t_end - t_phase1).count());
    }
    return DGLL_OK;
}

// CPython's random.seed(int) (Modules/_randommodule.c: random_seed -> init_by_array over the 32-bit little-endian words of
// abs(seed); init_genrand(19650218) first), restated; leaves the generator "exhausted" (index 624) as CPython does.
void mt_init_by_array(uint32_t* mt, const uint32_t* key, int64_t key_len) {
    constexpr int N = 624;
    mt[0] = 19650218U;
    for (int i = 1; i < N; i++) mt[i] = 1812433253U * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    int64_t i = 1, j = 0;
    for (int64_t k = (N > key_len ? N : key_len); k; k--) {
        mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525U)) + key[j] + (uint32_t)j;
        i++; j++;
        if (i >= N) { mt[0] = mt[N - 1]; i = 1; }
        if (j >= key_len) j = 0;
    }
    for (int k = N - 1; k; k--) {
        mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941U)) - (uint32_t)i;
        i++;
        if (i >= N) { mt[0] = mt[N - 1]; i = 1; }
    }
    mt[0] = 0x80000000U;
}
}  // namespace

DGLL_API int dgll_host_sample_neighbors(uint32_t* mt_state, int* mt_index, const int64_t* indptr, const int64_t* indices,
                                        const int64_t* seeds, int64_t n_seeds, int64_t fanout, int64_t setsize,
                                        int64_t* out_src, int64_t* out_dst, int64_t* out_counts, int64_t capacity,
                                        int64_t* n_out) {
    DGLL_REQUIRE(mt_state && mt_index && indptr && indices && (seeds || n_seeds == 0) && out_src && out_counts && n_out, "NULL argument");
    DGLL_REQUIRE(*mt_index >= 0 && *mt_index <= 624, "bad generator index");
    MT19937 rng{mt_state, *mt_index};
    const int code = sample_hop(rng, indptr, indices, seeds, n_seeds, fanout, setsize, out_src, out_dst, out_counts, capacity, n_out, 16);
    if (code == DGLL_OK) *mt_index = rng.idx;
    return code;
}

// random.seed(int) -> the 624-word state + index the calls above take: `key` = the 32-bit little-endian words of abs(seed)
// ([0] for seed 0).  With it a sampler stream can live OUTSIDE the interpreter's global generator.
DGLL_API int dgll_host_mt_seed(const uint32_t* key, int64_t key_len, uint32_t* mt_state, int* mt_index) {
    DGLL_REQUIRE(key && key_len > 0 && mt_state && mt_index, "NULL argument");
    mt_init_by_array(mt_state, key, key_len);
    *mt_index = 624;
    return DGLL_OK;
}

// A whole mini-batch under ITS OWN generator: what the reference's loop (dgllsampler.py:10-21 over base_sampler.py:45-58) draws
// when `random.seed(seed)` is called right before the batch.  Batches seeded individually are independent of each other, so
// several of them can be drawn CONCURRENTLY (one call per host thread; nothing here touches shared state), each still
// bit-identical to the reference loop under the same seed -- the sequential single-stream mode (dgll_host_sample_neighbors on
// the interpreter's generator) stays the default-compatible one.
//   hops are given in SAMPLING order (the reference's reversed(fanouts)): hop h draws around the sources of hop h-1 (hop 0
//   around `seeds`), duplicates kept.  fanouts[h] < 0 = every neighbour.  out_src[h] / out_dst[h] / out_counts[h]: caller-owned,
//   capacity[h] edges / (number of hop seeds) counts.  defer_last != 0: the LAST hop keeps neighbour POSITIONS in out_src and
//   leaves out_dst untouched (dgll_host_translate_neighbors, or a device-side translation, finishes it).
//   max_threads bounds the helper threads of the id translation inside this call (1 when many batches run side by side).
DGLL_API int dgll_host_sample_batch_seeded(const uint32_t* key, int64_t key_len, const int64_t* indptr, const int64_t* indices,
                                           const int64_t* seeds, int64_t n_seeds, const int64_t* fanouts, const int64_t* setsizes,
                                           int n_hops, int64_t* const* out_src, int64_t* const* out_dst, int64_t* const* out_counts,
                                           const int64_t* capacity, int64_t* n_out, int defer_last, int max_threads) {
    DGLL_REQUIRE(key && key_len > 0 && indptr && indices && (seeds || n_seeds == 0) && fanouts && setsizes && n_hops > 0 && out_src &&
                 out_dst && out_counts && capacity && n_out, "NULL argument");
    std::vector<uint32_t> state(624);
    mt_init_by_array(state.data(), key, key_len);
    MT19937 rng{state.data(), 624};
    const int64_t* hop_seeds = seeds;
    int64_t n_hop_seeds = n_seeds;
    for (int h = 0; h < n_hops; ++h) {
        const bool defer = defer_last && h == n_hops - 1;
        const int code = sample_hop(rng, indptr, indices, hop_seeds, n_hop_seeds, fanouts[h], setsizes[h], out_src[h],
                                    defer ? nullptr : out_dst[h], out_counts[h], capacity[h], &n_out[h], max_threads);
        if (code != DGLL_OK) return code;
        hop_seeds = out_src[h];
        n_hop_seeds = n_out[h];
    }
    return DGLL_OK;
}


// Phase 2 of dgll_host_sample_neighbors on its own: src_inout holds, per seed occurrence, the POSITIONS of the kept neighbours
// in that seed's adjacency list (what the call above leaves when out_dst is NULL); they are replaced by the neighbour ids and
// out_dst receives the seed of every edge.  Touches no generator state, so it may run on another thread while the next batch
// is being drawn.
DGLL_API int dgll_host_translate_neighbors(const int64_t* indptr, const int64_t* indices, const int64_t* seeds, int64_t n_seeds,
                                           const int64_t* counts, int64_t* src_inout, int64_t* out_dst) {
    DGLL_REQUIRE(indptr && indices && (seeds || n_seeds == 0) && counts && src_inout && out_dst, "NULL argument");
    std::vector<int64_t> offset((size_t)n_seeds + 1, 0);
    for (int64_t s = 0; s < n_seeds; ++s) offset[s + 1] = offset[s] + counts[s];
    const int64_t total = offset[n_seeds];
    auto translate = [&](int64_t s0, int64_t s1) {
        for (int64_t s = s0; s < s1; ++s) {
            const int64_t v = seeds[s];
            const int64_t* nb = indices + indptr[v];
            for (int64_t k = offset[s]; k < offset[s + 1]; ++k) {
                src_inout[k] = nb[src_inout[k]];
                out_dst[k] = v;
            }
        }
    };
    const unsigned hw = std::thread::hardware_concurrency();
    const int64_t n_threads = total < (1 << 16) ? 1 : std::min<int64_t>(hw ? hw : 1, 16);
    if (n_threads <= 1) {
        translate(0, n_seeds);
    } else {
        std::vector<std::thread> workers;
        for (int64_t t = 0; t < n_threads; ++t)
            workers.emplace_back(translate, n_seeds * t / n_threads, n_seeds * (t + 1) / n_threads);
        for (auto& w : workers) w.join();
    }
    return DGLL_OK;
}
