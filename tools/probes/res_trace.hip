// Probe: where a wavefront of the resident-weights transform kernel spends a row block.  Includes the product kernel source
// with DGLL_RES_TRACE defined: lane 0 of every wave of the first 16 workgroups stamps the cycle counter at the start of a
// block, after the first wait, after each of the 8 chunk phases, after the next block's loads are issued, around the epilogue.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude tools/probes/res_trace.hip -o tools/probes/res_trace.bin
#define DGLL_RES_TRACE 1
#define DGLL_RES_PROBE 1
#include "../../dgll_amd/csrc/dense.hip"
#include <cstdio>
#include <vector>

namespace dgll {
void set_error(const std::string& msg) { fprintf(stderr, "error: %s\n", msg.c_str()); }
int hip_fail(hipError_t e, const char* what) { fprintf(stderr, "hip error %d at %s\n", (int)e, what); return -1; }
}

template <int NTW, int NC, int CS, int COLSPLIT>
static void run(const char* name, int pairs, int K) {
    const int64_t M = 2449029;
    bf16_t *x0, *x1, *w0, *w1, *out; unsigned long long* trace;
    (void)hipMalloc(&x0, M * K * 2); (void)hipMalloc(&x1, M * K * 2); (void)hipMalloc(&out, M * 256 * 2);
    (void)hipMalloc(&w0, 256 * 256 * 2); (void)hipMalloc(&w1, 256 * 256 * 2);
    (void)hipMemset(x0, 0, M * K * 2); (void)hipMemset(x1, 0, M * K * 2); (void)hipMemset(w0, 0, 256 * 256 * 2); (void)hipMemset(w1, 0, 256 * 256 * 2);
    const size_t tn = 16 * 8 * 4 * 16;
    (void)hipMalloc(&trace, tn * 8); (void)hipMemset(trace, 0, tn * 8);
    MfmaGemmArgs a{};
    a.A[0] = x0; a.lda[0] = K; a.K[0] = K; a.Wt[0] = w0; a.ldw[0] = 256;
    a.A[1] = x1; a.lda[1] = K; a.K[1] = K; a.Wt[1] = w1; a.ldw[1] = 256;
    a.pairs = pairs; a.out = out; a.ldo = 256; a.M = M; a.N = 256; a.relu = 1;
    a.trace = nullptr;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch_mfma_res<NTW, NC, CS, COLSPLIT>(a, 0);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) launch_mfma_res<NTW, NC, CS, COLSPLIT>(a, 0);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    a.trace = trace;
    launch_mfma_res<NTW, NC, CS, COLSPLIT>(a, 0);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> t(tn);
    (void)hipMemcpy(t.data(), trace, tn * 8, hipMemcpyDeviceToHost);
    printf("%s: %.3f ms (%.2f TB/s); cycle-counter deltas averaged over 16 workgroups x 8 waves, blocks 1..3 of each\n", name, ms,
           ((double)M * K * 2 * pairs + (double)M * 512) / 1e9 / ms);
    double d[16] = {0}; int n = 0;
    for (int wg = 0; wg < 16; ++wg) for (int w = 0; w < 8; ++w) for (int b = 1; b < 4; ++b) {
        const unsigned long long* e = &t[((wg * 8 + w) * 4 + b) * 16];
        if (!e[0] || !e[12]) continue;
        ++n;
        for (int c = 0; c < NC; ++c) d[1 + c] += (double)(e[2 + c] - (c ? e[1 + c] : e[0]));   // (phase 0 from the block's start)
        d[9] += (double)(e[11] - e[2 + NC - 1]);                         // issue next block's loads
        d[10] += (double)(e[12] - e[11]);                                // epilogue
        d[11] += (double)(e[12] - e[0]);                                 // whole block
    }
    if (!n) { printf("  no samples\n"); return; }
    printf(" ");
    for (int c = 0; c < NC; ++c) printf(" %sphase %d: %.0f", c ? "| " : "", c, d[1 + c] / n);
    printf("  | issue next: %.0f  | epilogue: %.0f  | block: %.0f  (%d samples)\n", d[9] / n, d[10] / n, d[11] / n, n);
    // one wave's raw timeline
    const unsigned long long* e = &t[((3 * 8 + 2) * 4 + 1) * 16];
    printf("  wg 3 wave 2 block 1 raw:");
    for (int i = 1; i <= 12; ++i) if (e[i]) printf(" %llu", e[i] - e[0]);
    printf("\n");
    (void)hipFree(x0); (void)hipFree(x1); (void)hipFree(out); (void)hipFree(w0); (void)hipFree(w1); (void)hipFree(trace);
}

int main() {
    run<4, 8, 1, 2>("fused 256+256 -> 256 (two workgroups per row block)", 2, 256);
    run<4, 4, 2, 1>("single 256 -> 256 (two waves per row group)", 1, 256);
    return 0;
}
