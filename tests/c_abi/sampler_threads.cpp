// sampler_threads.cpp -- sanitizer driver for the threaded host code of dgll_amd/csrc/sampler.hip (plain g++ build, no GPU).
//
// What MiniBatchPipeline(sampler_threads=K) does natively: K host threads call dgll_host_sample_batch_seeded at the same time
// on ONE shared read-only adjacency, each for its own batches, each with its own output arrays.  Here: 8 threads x 24 batches
// on a hub-heavy graph; every batch is also drawn single-threaded first and the threaded result must equal it entry by entry.
// Then dgll_host_sample_neighbors with its helper threads (phase 2 translation over disjoint output ranges) and
// dgll_host_translate_neighbors.  Built with -fsanitize=thread (data races) and -fsanitize=address,undefined (bounds).
// Exit code 0 = results equal; the sanitizers report on stderr and set their own exit code.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/dgll_hip.h"

namespace dgll {
static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }   // api.hip's definition, restated for the host-only build
}  // namespace dgll
extern "C" const char* dgll_hip_last_error(void) { return dgll::g_err.c_str(); }

namespace {
struct Graph {
    std::vector<int64_t> indptr, indices;
};

uint64_t lcg(uint64_t& s) {
    s = s * 6364136223846793005ULL + 1442695040888963407ULL;
    return s >> 33;
}

Graph make_graph(int64_t n) {
    Graph g;
    uint64_t s = 12345;
    g.indptr.push_back(0);
    for (int64_t v = 0; v < n; ++v) {
        int64_t deg = (v % 97 == 0) ? 3000 : (v % 9 == 0 ? 0 : (int64_t)(lcg(s) % 40));   // hubs, empty rows, short rows
        for (int64_t k = 0; k < deg; ++k) g.indices.push_back((int64_t)(lcg(s) % (uint64_t)n));
        g.indptr.push_back((int64_t)g.indices.size());
    }
    return g;
}

int64_t setsize(int64_t k) {   // CPython 3.10 random.sample (the Python caller computes this with the interpreter's math)
    int64_t ss = 21;
    if (k > 5) {
        int64_t p = 1;
        while (p < 3 * k) p *= 4;
        ss += p;
    }
    return ss;
}

struct BatchOut {
    std::vector<std::vector<int64_t>> src, dst, cnt;
    std::vector<int64_t> n_out;
};

int draw(const Graph& g, const std::vector<int64_t>& seeds, uint32_t seed, const std::vector<int64_t>& fan, int defer, BatchOut& o) {
    const int L = (int)fan.size();
    std::vector<int64_t> ss(L), cap(L);
    int64_t rows = (int64_t)seeds.size();
    o.src.assign(L, {}); o.dst.assign(L, {}); o.cnt.assign(L, {}); o.n_out.assign(L, 0);
    std::vector<int64_t*> ps(L), pd(L), pc(L);
    for (int h = 0; h < L; ++h) {
        ss[h] = setsize(fan[h]);
        o.cnt[h].assign(rows, 0);
        cap[h] = rows * fan[h];
        o.src[h].assign(cap[h], -1);
        o.dst[h].assign(cap[h], -1);
        ps[h] = o.src[h].data(); pd[h] = o.dst[h].data(); pc[h] = o.cnt[h].data();
        rows = cap[h];
    }
    const uint32_t key[2] = {seed, 7u};
    return dgll_host_sample_batch_seeded(key, 2, g.indptr.data(), g.indices.data(), seeds.data(), (int64_t)seeds.size(), fan.data(),
                                         ss.data(), L, ps.data(), pd.data(), pc.data(), cap.data(), o.n_out.data(), defer, 4);
}

bool same(const BatchOut& a, const BatchOut& b, bool with_dst_of_last) {
    if (a.n_out != b.n_out) return false;
    for (size_t h = 0; h < a.src.size(); ++h) {
        const int64_t n = a.n_out[h];
        if (std::memcmp(a.src[h].data(), b.src[h].data(), n * 8)) return false;
        if ((with_dst_of_last || h + 1 < a.src.size()) && std::memcmp(a.dst[h].data(), b.dst[h].data(), n * 8)) return false;
        if (a.cnt[h] != b.cnt[h]) return false;
    }
    return true;
}
}  // namespace

int main() {
    const int64_t n = 20000;
    const Graph g = make_graph(n);
    const std::vector<int64_t> fan = {10, 10, 25};          // sampling order (the reference's reversed(fanouts))
    const int n_batches = 24, n_threads = 8, batch = 128;
    std::vector<std::vector<int64_t>> seeds(n_batches);
    uint64_t s = 99;
    for (auto& b : seeds)
        for (int k = 0; k < batch; ++k) b.push_back((int64_t)(lcg(s) % (uint64_t)n));

    std::vector<BatchOut> want(n_batches), got(n_batches);
    for (int b = 0; b < n_batches; ++b)
        if (draw(g, seeds[b], 1000u + b, fan, b & 1, want[b]) != DGLL_OK) { std::fprintf(stderr, "draw failed: %s\n", dgll::g_err.c_str()); return 2; }
    std::vector<int> codes(n_threads, 0);
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; ++t)
        th.emplace_back([&, t] {
            for (int b = t; b < n_batches; b += n_threads) codes[t] |= draw(g, seeds[b], 1000u + b, fan, b & 1, got[b]);
        });
    for (auto& x : th) x.join();
    for (int t = 0; t < n_threads; ++t)
        if (codes[t]) { std::fprintf(stderr, "threaded draw failed\n"); return 2; }
    for (int b = 0; b < n_batches; ++b)
        if (!same(want[b], got[b], !(b & 1))) { std::fprintf(stderr, "batch %d differs between the threaded and the sequential draw\n", b); return 1; }

    // the single-stream entry with its helper threads (>= 65536 edges switches them on), then the stand-alone translation
    std::vector<uint32_t> st(624);
    int idx = 0;
    const uint32_t key[1] = {42u};
    if (dgll_host_mt_seed(key, 1, st.data(), &idx) != DGLL_OK) return 2;
    std::vector<int64_t> many;
    for (int64_t v = 0; v < n; ++v) many.push_back(v);
    const int64_t cap = n * 25;
    std::vector<int64_t> src(cap), dst(cap), cnt(n), src2(cap), dst2(cap), cnt2(n);
    int64_t n_out = 0, n_out2 = 0;
    std::vector<uint32_t> st2 = st;
    int idx2 = idx;
    if (dgll_host_sample_neighbors(st.data(), &idx, g.indptr.data(), g.indices.data(), many.data(), n, 25, setsize(25), src.data(),
                                   dst.data(), cnt.data(), cap, &n_out) != DGLL_OK) return 2;
    if (dgll_host_sample_neighbors(st2.data(), &idx2, g.indptr.data(), g.indices.data(), many.data(), n, 25, setsize(25), src2.data(),
                                   nullptr, cnt2.data(), cap, &n_out2) != DGLL_OK) return 2;
    if (dgll_host_translate_neighbors(g.indptr.data(), g.indices.data(), many.data(), n, cnt2.data(), src2.data(), dst2.data()) != DGLL_OK) return 2;
    if (n_out != n_out2 || idx != idx2 || st != st2 || std::memcmp(src.data(), src2.data(), n_out * 8) ||
        std::memcmp(dst.data(), dst2.data(), n_out * 8) || cnt != cnt2) {
        std::fprintf(stderr, "deferred translation differs from the in-call one\n");
        return 1;
    }
    // the native sampler pool (what MiniBatchPipeline runs now): 6 workers, 9 slots, 40 batches of 100 seeds (the last one ragged); a
    // consumer dequeues in order, checks every batch against the same draw made sequentially, and releases its slot two batches late
    {
        const int64_t bs = 100, n_train = 40 * bs - 33, L = (int64_t)fan.size();
        const int pool_threads = 6, n_slots = 9;
        std::vector<int64_t> train((size_t)n_train);
        uint64_t s2 = 7;
        for (auto& v : train) v = (int64_t)(lcg(s2) % (uint64_t)n);
        std::vector<int64_t> ss(L), rows_cap(L), capv(L), off_src(L, 0), off_ptr(L);
        int64_t rows = bs, o = bs;
        for (int h = 0; h < L; ++h) { ss[h] = setsize(fan[h]); rows_cap[h] = rows; capv[h] = rows * fan[h]; rows = capv[h]; }
        for (int h = 0; h + 1 < L; ++h) { off_src[h] = o; o += capv[h]; }
        for (int h = 0; h < L; ++h) { off_ptr[h] = o; o += rows_cap[h] + 1; }
        const int64_t entries = o;
        std::vector<std::vector<int64_t>> staged(n_slots, std::vector<int64_t>((size_t)entries, -7));
        std::vector<std::vector<uint32_t>> pos(n_slots, std::vector<uint32_t>((size_t)capv[L - 1], 0u));
        std::vector<int64_t*> sb(n_slots);
        std::vector<void*> pb(n_slots);
        for (int k = 0; k < n_slots; ++k) { sb[k] = staged[k].data(); pb[k] = pos[k].data(); }
        dgll_sampler_pool* pool = nullptr;
        if (dgll_host_sampler_pool_create(&pool, g.indptr.data(), g.indices.data(), train.data(), n_train, bs, fan.data(), ss.data(), (int)L, 3, 1,
                                          pool_threads, n_slots, sb.data(), entries, 0, off_src.data(), off_ptr.data(), pb.data(), 4) != DGLL_OK) {
            std::fprintf(stderr, "pool create failed: %s\n", dgll::g_err.c_str());
            return 2;
        }
        std::vector<int> held;
        int64_t desc[3 + 16];
        int64_t seen = 0;
        for (;;) {
            const int rc = dgll_host_sampler_pool_next(pool, desc, nullptr);
            if (rc == 1) break;
            if (rc != DGLL_OK) { std::fprintf(stderr, "pool next failed: %s\n", dgll::g_err.c_str()); return 2; }
            const int64_t i = desc[0];
            const int slot = (int)desc[1];
            if (i != seen || slot != i % n_slots) { std::fprintf(stderr, "pool delivered batch %lld out of order\n", (long long)i); return 1; }
            const int64_t first = i * bs, ns = std::min<int64_t>(bs, n_train - first);
            std::vector<int64_t> sd(train.begin() + first, train.begin() + first + ns);
            BatchOut want_b;
            const uint64_t seed = (3ull << 40) | (1ull << 20) | (uint64_t)i;
            {   // the same draw, sequentially, through the plain entry point (two-word key: the seed's low and high half)
                std::vector<int64_t> cap2(L);
                int64_t r = ns;
                want_b.src.assign(L, {}); want_b.dst.assign(L, {}); want_b.cnt.assign(L, {}); want_b.n_out.assign(L, 0);
                std::vector<int64_t*> ps(L), pd(L), pc(L);
                for (int h = 0; h < L; ++h) {
                    want_b.cnt[h].assign(r, 0); cap2[h] = r * fan[h]; want_b.src[h].assign(cap2[h], -1); want_b.dst[h].assign(cap2[h], -1);
                    ps[h] = want_b.src[h].data(); pd[h] = want_b.dst[h].data(); pc[h] = want_b.cnt[h].data(); r = cap2[h];
                }
                const uint32_t key2[2] = {(uint32_t)(seed & 0xffffffffu), (uint32_t)(seed >> 32)};
                if (dgll_host_sample_batch_seeded(key2, 2, g.indptr.data(), g.indices.data(), sd.data(), ns, fan.data(), ss.data(), (int)L, ps.data(),
                                                  pd.data(), pc.data(), cap2.data(), want_b.n_out.data(), 1, 1) != DGLL_OK) return 2;
            }
            bool ok = std::memcmp(staged[slot].data(), sd.data(), (size_t)ns * 8) == 0;
            int64_t r = ns;
            for (int h = 0; h < L && ok; ++h) {
                ok = desc[3 + h] == r && desc[3 + L + h] == want_b.n_out[h];
                const int64_t* ptr = staged[slot].data() + off_ptr[h];
                for (int64_t q = 0; q < r && ok; ++q) ok = ptr[q + 1] - ptr[q] == want_b.cnt[h][(size_t)q];
                if (h + 1 < L) ok = ok && std::memcmp(staged[slot].data() + off_src[h], want_b.src[h].data(), (size_t)want_b.n_out[h] * 8) == 0;
                else for (int64_t q = 0; q < want_b.n_out[h] && ok; ++q) ok = (int64_t)pos[slot][(size_t)q] == want_b.src[h][(size_t)q];
                r = want_b.n_out[h];
            }
            if (!ok) { std::fprintf(stderr, "pool batch %lld differs from the sequential draw\n", (long long)i); return 1; }
            held.push_back(slot);
            if (held.size() > 2) { if (dgll_host_sampler_pool_release(pool, held.front()) != DGLL_OK) return 2; held.erase(held.begin()); }
            ++seen;
        }
        dgll_host_sampler_pool_destroy(pool);
        if (seen != 40) { std::fprintf(stderr, "pool delivered %lld of 40 batches\n", (long long)seen); return 1; }
        // a pool that is destroyed while its workers still wait for slots (the consumer left early) must come down cleanly
        if (dgll_host_sampler_pool_create(&pool, g.indptr.data(), g.indices.data(), train.data(), n_train, bs, fan.data(), ss.data(), (int)L, 3, 2,
                                          pool_threads, n_slots, sb.data(), entries, 0, off_src.data(), off_ptr.data(), pb.data(), 4) != DGLL_OK) return 2;
        if (dgll_host_sampler_pool_next(pool, desc, nullptr) != DGLL_OK) return 2;
        dgll_host_sampler_pool_destroy(pool);
        std::printf("sampler_threads: native pool, 40 batches x %d workers over %d slots, in order, equal the sequential draw\n", pool_threads, n_slots);
    }
    std::printf("sampler_threads: %d batches x %d threads equal the sequential draw; %lld edges through the helper threads\n", n_batches,
                n_threads, (long long)n_out);
    return 0;
}
