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
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
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

// ---- native producer: a pool of sampler threads behind one in-order hand-over ---------------------------------------------------
// MiniBatchPipeline(sampler_threads=K) drew batches on K PYTHON threads: the draw itself ran without the interpreter lock, but every
// batch's bookkeeping (buffers, prefix sums, narrowing, queue hand-overs) took it, next to the loading and the consuming thread --
// more than 8 threads made the epoch SLOWER (bench.py, Reddit shape: 782 batches/s at 8 threads, 635 at 16).  Here the K threads
// are native: a worker claims the next batch index, waits for that batch's slot (batch i uses slot i % n_slots: slots come free in
// order, so the earliest batch can always proceed), draws the batch under ITS OWN seed -- random.seed((base << 40) | (epoch << 20) |
// batch), bit for bit what the Python path and the reference's loop draw -- straight into the slot's (pinned, caller-owned) buffers in
// the layout the loading stage uploads with ONE copy: [seeds | source ids of hop 0 .. L-2 | row pointers of hop 0 .. L-1], plus the
// outermost hop's neighbour POSITIONS narrowed to 16 / 32 bits.  Python only dequeues finished batches, in order.
// The buffer offsets are the caller's (they are what its loading stage uses); nothing here touches the GPU or the interpreter.
constexpr int kPoolMaxHops = 8;

struct dgll_sampler_pool {
    const int64_t* indptr = nullptr;
    const int64_t* indices = nullptr;
    const int64_t* train = nullptr;
    int64_t n_train = 0, batch_size = 0, n_batches = 0;
    int n_hops = 0;
    int64_t fanouts[kPoolMaxHops] = {0}, setsizes[kPoolMaxHops] = {0};
    uint64_t base_seed = 0, epoch = 0;
    int64_t off_seeds = 0, off_src[kPoolMaxHops] = {0}, off_ptr[kPoolMaxHops] = {0}, staged_entries = 0;
    int64_t cap[kPoolMaxHops] = {0};          // upper bound of hop h's edges for a full batch
    int pos_bytes = 8;
    struct Slot {
        int64_t* staged = nullptr;
        void* pos = nullptr;
        int state = 0;                        // 0 free, 1 being filled, 2 ready, 3 handed out
        int64_t batch = -1;
        int64_t rows[kPoolMaxHops] = {0}, n_src[kPoolMaxHops] = {0};
        double sample_ms = 0.0;
    };
    std::vector<Slot> slots;
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable cv_free, cv_ready;
    int64_t next_batch = 0, next_out = 0;
    bool stop = false;
    int error = DGLL_OK;
    std::string error_text;
};

namespace {
void pool_worker(dgll_sampler_pool* p) {
    const int L = p->n_hops;
    std::vector<std::vector<int64_t>> dst(L), cnt(L);
    std::vector<int64_t> outer;               // the outermost hop's positions as the draw leaves them (int64), narrowed afterwards
    int64_t rows_cap = p->batch_size;
    for (int h = 0; h < L; ++h) {
        cnt[h].resize((size_t)rows_cap);
        if (h + 1 < L) dst[h].resize((size_t)p->cap[h]);      // inner hops are translated to ids (the next hop's seeds): dst is scratch
        rows_cap = p->cap[h];
    }
    outer.resize((size_t)p->cap[L - 1]);
    for (;;) {
        int64_t i;
        dgll_sampler_pool::Slot* sl;
        {
            std::unique_lock<std::mutex> lk(p->mu);
            if (p->stop || p->next_batch >= p->n_batches) return;
            i = p->next_batch++;
            sl = &p->slots[(size_t)(i % (int64_t)p->slots.size())];
            p->cv_free.wait(lk, [&] { return p->stop || sl->state == 0; });
            if (p->stop) return;
            sl->state = 1;
            sl->batch = i;
        }
        const auto t0 = std::chrono::steady_clock::now();
        const int64_t first = i * p->batch_size;
        const int64_t n_seeds = std::min<int64_t>(p->batch_size, p->n_train - first);
        int64_t* st = sl->staged;
        std::memcpy(st + p->off_seeds, p->train + first, (size_t)n_seeds * sizeof(int64_t));
        // random.seed(int): the 32-bit little-endian words of the seed, [0] for 0 (fast_sampler._seed_key)
        const uint64_t seed = (p->base_seed << 40) | (p->epoch << 20) | (uint64_t)i;
        uint32_t key[2] = {(uint32_t)(seed & 0xffffffffu), (uint32_t)(seed >> 32)};
        const int64_t key_len = key[1] ? 2 : 1;
        int64_t* src[kPoolMaxHops];
        int64_t* dstp[kPoolMaxHops];
        int64_t* cntp[kPoolMaxHops];
        for (int h = 0; h < L; ++h) {
            src[h] = h + 1 < L ? st + p->off_src[h] : outer.data();
            dstp[h] = h + 1 < L ? dst[h].data() : outer.data();       // (never written for the deferred outermost hop)
            cntp[h] = cnt[h].data();
        }
        int64_t n_out[kPoolMaxHops] = {0};
        int code = dgll_host_sample_batch_seeded(key, key_len, p->indptr, p->indices, st + p->off_seeds, n_seeds, p->fanouts, p->setsizes,
                                                 L, src, dstp, cntp, p->cap, n_out, /*defer_last=*/1, /*max_threads=*/1);
        if (code == DGLL_OK) {
            int64_t rows = n_seeds;
            for (int h = 0; h < L; ++h) {                     // row pointers of hop h: prefix sums of the kept-neighbour counts
                int64_t* ptr = st + p->off_ptr[h];
                int64_t acc = 0;
                ptr[0] = 0;
                for (int64_t r = 0; r < rows; ++r) { acc += cnt[h][(size_t)r]; ptr[r + 1] = acc; }
                sl->rows[h] = rows;
                sl->n_src[h] = n_out[h];
                rows = n_out[h];
            }
            const int64_t n_pos = n_out[L - 1];
            if (p->pos_bytes == 2) {
                uint16_t* o = static_cast<uint16_t*>(sl->pos);
                for (int64_t k = 0; k < n_pos; ++k) o[k] = (uint16_t)outer[(size_t)k];
            } else if (p->pos_bytes == 4) {
                uint32_t* o = static_cast<uint32_t*>(sl->pos);
                for (int64_t k = 0; k < n_pos; ++k) o[k] = (uint32_t)outer[(size_t)k];
            } else {
                std::memcpy(sl->pos, outer.data(), (size_t)n_pos * sizeof(int64_t));
            }
        }
        sl->sample_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        {
            std::lock_guard<std::mutex> lk(p->mu);
            if (code != DGLL_OK && p->error == DGLL_OK) {
                p->error = code;
                p->error_text = dgll_hip_last_error();
                p->stop = true;
            }
            sl->state = 2;
        }
        p->cv_ready.notify_all();
        if (code != DGLL_OK) { p->cv_free.notify_all(); return; }
    }
}
}  // namespace

DGLL_API int dgll_host_sampler_pool_create(dgll_sampler_pool** out, const int64_t* indptr, const int64_t* indices, const int64_t* train_nodes,
                                           int64_t n_train, int64_t batch_size, const int64_t* fanouts, const int64_t* setsizes, int n_hops,
                                           uint64_t base_seed, uint64_t epoch, int n_threads, int n_slots, int64_t* const* staged_bufs,
                                           int64_t staged_entries, int64_t off_seeds, const int64_t* off_src, const int64_t* off_ptr,
                                           void* const* pos_bufs, int pos_bytes) {
    DGLL_REQUIRE(out && indptr && indices && (train_nodes || n_train == 0) && fanouts && setsizes && staged_bufs && off_src && off_ptr && pos_bufs,
                 "NULL argument");
    DGLL_REQUIRE(n_hops > 0 && n_hops <= kPoolMaxHops && batch_size > 0 && n_train >= 0 && n_threads > 0, "bad size");
    DGLL_REQUIRE(n_slots >= n_threads + 1, "the pool needs more slots than threads (a finished batch holds its slot until it is released)");
    DGLL_REQUIRE(pos_bytes == 2 || pos_bytes == 4 || pos_bytes == 8, "positions leave as 2-, 4- or 8-byte integers");
    DGLL_REQUIRE(base_seed < (1ull << 24) && epoch < (1ull << 20), "base_seed < 2**24 and epoch < 2**20 (the seed is one 64-bit word)");
    const int64_t n_batches = (n_train + batch_size - 1) / batch_size;
    DGLL_REQUIRE(n_batches < (1ll << 20), "batch index must stay below 2**20 (fast_sampler.batch_seed)");
    auto* p = new dgll_sampler_pool();
    p->indptr = indptr; p->indices = indices; p->train = train_nodes; p->n_train = n_train; p->batch_size = batch_size;
    p->n_batches = n_batches; p->n_hops = n_hops; p->base_seed = base_seed; p->epoch = epoch;
    p->off_seeds = off_seeds; p->staged_entries = staged_entries; p->pos_bytes = pos_bytes;
    int64_t rows = batch_size;
    for (int h = 0; h < n_hops; ++h) {
        if (fanouts[h] <= 0) { delete p; dgll::set_error("the pool needs integer fan-outs"); return DGLL_ERR_INVALID; }
        p->fanouts[h] = fanouts[h]; p->setsizes[h] = setsizes[h];
        p->off_src[h] = h + 1 < n_hops ? off_src[h] : 0; p->off_ptr[h] = off_ptr[h];
        p->cap[h] = rows * fanouts[h];
        // the caller's layout must hold the upper bounds
        const bool fits = off_ptr[h] + rows + 1 <= staged_entries && (h + 1 == n_hops || off_src[h] + p->cap[h] <= staged_entries);
        if (!fits || off_seeds + batch_size > staged_entries) { delete p; dgll::set_error("staged buffer layout too small for the upper bounds"); return DGLL_ERR_INVALID; }
        rows = p->cap[h];
    }
    p->slots.resize((size_t)n_slots);
    for (int k = 0; k < n_slots; ++k) {
        if (!staged_bufs[k] || !pos_bufs[k]) { delete p; dgll::set_error("NULL slot buffer"); return DGLL_ERR_INVALID; }
        p->slots[(size_t)k].staged = staged_bufs[k];
        p->slots[(size_t)k].pos = pos_bufs[k];
    }
    for (int t = 0; t < n_threads; ++t) p->workers.emplace_back(pool_worker, p);
    *out = p;
    return DGLL_OK;
}

// The next batch IN ORDER (blocks until it is drawn): 0 = `out` filled (out[0] batch index, out[1] slot, out[2] hops L, out[3 .. 3+L) rows of
// hop h, out[3+L .. 3+2L) edges of hop h, sample_ms the worker's time for it); 1 = the epoch is over; < 0 = a worker failed (the error text is
// this thread's dgll_hip_last_error()).  One consumer thread.
DGLL_API int dgll_host_sampler_pool_next(dgll_sampler_pool* p, int64_t* out, double* sample_ms) {
    DGLL_REQUIRE(p && out, "NULL argument");
    std::unique_lock<std::mutex> lk(p->mu);
    if (p->next_out >= p->n_batches) return 1;
    auto& sl = p->slots[(size_t)(p->next_out % (int64_t)p->slots.size())];
    p->cv_ready.wait(lk, [&] { return p->error != DGLL_OK || p->stop || (sl.state == 2 && sl.batch == p->next_out); });
    if (p->error != DGLL_OK) { dgll::set_error("sampler pool: " + p->error_text); return p->error; }
    if (!(sl.state == 2 && sl.batch == p->next_out)) return 1;       // stopped before this batch was drawn
    sl.state = 3;
    out[0] = sl.batch; out[1] = p->next_out % (int64_t)p->slots.size(); out[2] = p->n_hops;
    for (int h = 0; h < p->n_hops; ++h) { out[3 + h] = sl.rows[h]; out[3 + p->n_hops + h] = sl.n_src[h]; }
    if (sample_ms) *sample_ms = sl.sample_ms;
    p->next_out++;
    return DGLL_OK;
}

// The slot's buffers may be overwritten (its upload has completed).
DGLL_API int dgll_host_sampler_pool_release(dgll_sampler_pool* p, int slot) {
    DGLL_REQUIRE(p && slot >= 0 && slot < (int)p->slots.size(), "bad slot");
    {
        std::lock_guard<std::mutex> lk(p->mu);
        DGLL_REQUIRE(p->slots[(size_t)slot].state == 3, "slot was not handed out");
        p->slots[(size_t)slot].state = 0;
    }
    p->cv_free.notify_all();
    return DGLL_OK;
}

// Stops the workers (a batch in progress is finished first), joins them, frees the pool.  The slot buffers stay the caller's.
DGLL_API int dgll_host_sampler_pool_destroy(dgll_sampler_pool* p) {
    if (!p) return DGLL_OK;
    {
        std::lock_guard<std::mutex> lk(p->mu);
        p->stop = true;
    }
    p->cv_free.notify_all();
    p->cv_ready.notify_all();
    for (auto& w : p->workers) w.join();
    delete p;
    return DGLL_OK;
}
