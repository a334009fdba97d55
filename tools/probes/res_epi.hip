// Probe: the epilogue forms of the resident-weights transform side by side on the two gated shapes of the bench step --
// plain, plain + sign bits written, gate read as a bf16 matrix (general epilogue), gate read as bits.  Includes the product
// kernel source with DGLL_RES_TRACE: launch time from HIP events, cycles a wave spends in a block and in its epilogue from stamps.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -pragma-unroll-threshold=131072 -Iinclude tools/probes/res_epi.hip -o tools/probes/res_epi.bin
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
static void run(const char* name, int K, int lda) {
    const int64_t M = 2449029;
    bf16_t *x0, *x1, *w0, *w1, *out, *gate; uint32_t *bits, *bits_o; unsigned long long* trace;
    (void)hipMalloc(&x0, M * lda * 2); (void)hipMalloc(&x1, M * lda * 2); (void)hipMalloc(&out, M * 256 * 2); (void)hipMalloc(&gate, M * 256 * 2);
    (void)hipMalloc(&w0, 256 * 256 * 2); (void)hipMalloc(&w1, 256 * 256 * 2); (void)hipMalloc(&bits, M * 32); (void)hipMalloc(&bits_o, M * 32);
    (void)hipMemset(x0, 0, M * lda * 2); (void)hipMemset(x1, 0, M * lda * 2); (void)hipMemset(w0, 0, 256 * 256 * 2); (void)hipMemset(w1, 0, 256 * 256 * 2);
    (void)hipMemset(gate, 0x3f, M * 256 * 2); (void)hipMemset(bits, 0x55, M * 32);
    const size_t tn = 16 * 8 * 4 * 16;
    (void)hipMalloc(&trace, tn * 8);
    printf("%s\n", name);
    for (int mode = 0; mode < 5; ++mode) {
        MfmaGemmArgs a{};
        a.A[0] = x0; a.lda[0] = lda; a.K[0] = K; a.Wt[0] = w0; a.ldw[0] = 256;
        a.A[1] = x1; a.lda[1] = lda; a.K[1] = K; a.Wt[1] = w1; a.ldw[1] = 256;
        a.pairs = 2; a.out = out; a.ldo = 256; a.M = M; a.N = 256; a.relu = mode <= 1;
        const char* what = "plain";
        if (mode == 1) { a.bits_out = bits_o; a.ld_bits_out = 8; what = "plain + sign bits written"; }
        if (mode == 2) { a.out_gate = gate; a.ldgate = 256; what = "gate as bf16 (general epilogue)"; }
        if (mode == 3) { a.gate_bits = bits; a.ld_gate_bits = 8; what = "gate as bits"; }
        if (mode == 4) { a.gate_bits = bits; a.ld_gate_bits = 8; a.bits_out = bits_o; a.ld_bits_out = 8; what = "gate as bits + sign bits written"; }
        a.trace = nullptr;
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        launch_mfma_res<NTW, NC, CS, COLSPLIT>(a, 0);
        (void)hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) launch_mfma_res<NTW, NC, CS, COLSPLIT>(a, 0);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 10;
        (void)hipMemset(trace, 0, tn * 8);
        a.trace = trace;
        launch_mfma_res<NTW, NC, CS, COLSPLIT>(a, 0);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> t(tn);
        (void)hipMemcpy(t.data(), trace, tn * 8, hipMemcpyDeviceToHost);
        double epi = 0, blk = 0, last = 0; int n = 0;
        for (int wg = 0; wg < 16; ++wg) for (int w = 0; w < 8; ++w) for (int b = 1; b < 4; ++b) {
            const unsigned long long* e = &t[((wg * 8 + w) * 4 + b) * 16];
            if (!e[0] || !e[12]) continue;
            ++n; epi += (double)(e[12] - e[11]); blk += (double)(e[12] - e[0]); last += (double)(e[11] - e[2 + NC - 2 + (NC == 1)]);
        }
        printf("  %-36s %.3f ms | per block: %.0f cycles, epilogue %.0f, last phase %.0f (%d samples)\n", what, ms, blk / (n ? n : 1), epi / (n ? n : 1),
               last / (n ? n : 1), n);
    }
    (void)hipFree(x0); (void)hipFree(x1); (void)hipFree(out); (void)hipFree(w0); (void)hipFree(w1); (void)hipFree(trace);
    (void)hipFree(gate); (void)hipFree(bits); (void)hipFree(bits_o);
}

int main() {
    run<4, 8, 1, 2>("256 + 256 -> 256 (two workgroups per row block)", 256, 256);
    run<4, 2, 2, 1>("47 + 47 -> 256 (two waves per row group)", 47, 64);
    return 0;
}
