// Developer micro-benchmark of the FP32 MFMA block engine on the three dominant products of
// the Deep-TICA step (not part of the product build):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I deep_cartograph_amd/csrc tools/gemm_bench.hip \
//         deep_cartograph_amd/csrc/common.hip -o tools/gemm_bench
#include "gemm_kernels.h"
#include <vector>
#include <cstdlib>
#include <cstring>
#include <chrono>

using namespace dcv;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

template <class F>
static double time_ms(F f, int iters) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) f();
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / iters;
}

// GPU-side duration of single launches: an event pair around each launch, launches separated by a host sync
template <class F>
static double time_ms_single(F f, int iters) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    double tot = 0;
    for (int i = 0; i < iters; ++i) {
        CK(hipEventRecord(a, 0));
        f();
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        tot += ms;
    }
    return tot / iters;
}

#ifdef DCV_STAMP
static void dump_stamps(const char* what, int nblk) {
    std::vector<unsigned long long> h(8 * 8192);
    CK(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(dcv::g_stamp), h.size() * 8));
    if (nblk > 8192) nblk = 8192;
    if (const char* dir = getenv("DCV_STAMP_DUMP")) {
        char path[512];
        snprintf(path, sizeof path, "%s/stamps_%s.bin", dir, what);
        for (char* c = path + strlen(dir); *c; ++c) if (*c == ' ') *c = '_';
        if (FILE* f = fopen(path, "wb")) { fwrite(h.data(), 8, (size_t)nblk * 8, f); fclose(f); }
    }
    double ph[4] = {0, 0, 0, 0};
    double clk = 0;
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int b = 0; b < nblk; ++b) {
        for (int p = 0; p < 4; ++p) ph[p] += (double)(h[b * 8 + p + 1] - h[b * 8 + p]);
        clk += (double)(h[b * 8 + 2] - h[b * 8 + 1]) / (double)(h[b * 8 + 6] - h[b * 8 + 5]) * 0.1;  // GHz (s_memrealtime = 100 MHz)
        if (h[b * 8] < t0) t0 = h[b * 8];
        if (h[b * 8 + 4] > t1) t1 = h[b * 8 + 4];
    }
    printf("   %s stamps (cycles, mean/WG over %d WGs): prologue %.0f  mainloop %.0f  lds-transpose %.0f  store-loop %.0f | in-kernel clock %.3f GHz\n",
           what, nblk, ph[0] / nblk, ph[1] / nblk, ph[2] / nblk, ph[3] / nblk, clk / nblk);
    // s_memtime is per XCD: spans per XCD (block b runs on XCD b % 8); s_memrealtime (10 ns) is chip wide
    double xs = 0;
    for (int x = 0; x < 8; ++x) {
        unsigned long long a = ~0ull, z = 0;
        for (int b = x; b < nblk; b += 8) { if (h[b * 8] < a) a = h[b * 8]; if (h[b * 8 + 4] > z) z = h[b * 8 + 4]; }
        xs += (double)(z - a) / 8;
    }
    unsigned long long r0 = ~0ull, r1 = 0;
    for (int b = 0; b < nblk; ++b) { if (h[b * 8 + 5] < r0) r0 = h[b * 8 + 5]; if (h[b * 8 + 6] > r1) r1 = h[b * 8 + 6]; }
    int hist[16] = {0};
    for (int b = 0; b < nblk; ++b) { int k = (int)((h[b * 8 + 5] - r0) * 16 / (r1 - r0 + 1)); hist[k]++; }
    printf("      per-XCD span first stamp0 -> last stamp4: %.0f cycles; mainloop realtime span %.2f us; mainloop-start histogram (16 bins):", xs, (double)(r1 - r0) * 0.01);
    for (int k = 0; k < 16; ++k) printf(" %d", hist[k]);
    printf("\n");
}
#else
static void dump_stamps(const char*, int) {}
#endif

int main(int argc, char** argv) {
    const int64_t R = argc > 1 ? atoll(argv[1]) : 131072;
    const int F = 512, H1 = 256, H2 = 128;
    float *X, *W1, *Hb, *dZ, *slab, *W2, *H2b, *bpart, *b1;
    CK(hipMalloc(&X, (size_t)R * (F + 64) * 4));
    CK(hipMalloc(&W1, (size_t)H1 * (F + 64) * 4));
    CK(hipMalloc(&b1, (size_t)H1 * 4));
    CK(hipMalloc(&Hb, (size_t)R * H1 * 4));
    CK(hipMalloc(&dZ, (size_t)R * H1 * 4));
    CK(hipMalloc(&W2, (size_t)H2 * H1 * 4));
    CK(hipMalloc(&H2b, (size_t)R * H2 * 4));
    const int64_t slab_cap = (R + 127) / 128 + 1;   // slabs of H1 x F floats: one per smallest k-chunk (256 rows); launch_gemm refuses more splits than this
    CK(hipMalloc(&slab, (size_t)slab_cap * H1 * F * 4));
    CK(hipMalloc(&bpart, (size_t)(R / 32 + 8) * H1 * 4));
    std::vector<float> h((size_t)R * F);
    const bool zeros = argc > 3 && !strcmp(argv[3], "zero");
    for (auto& v : h) v = zeros ? 0.f : (float)rand() / RAND_MAX - 0.5f;
    CK(hipMemcpy(X, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(W1, h.data(), (size_t)H1 * F * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(W2, h.data(), (size_t)H2 * H1 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(b1, h.data(), (size_t)H1 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dZ, h.data(), (size_t)R * H1 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(H2b, h.data(), (size_t)R * H2 * 4, hipMemcpyHostToDevice));
    CK(hipMemset(Hb, 0, (size_t)R * H1 * 4));
    hipStream_t s = 0;
    TailWs tw;   // workspace of the contraction-split tail tile, as the engine owns one (DCV_BENCH_NOTAIL=1: none)
    if (!alloc_tail_ws(&tw, 8)) { printf("tail workspace: allocation failed\n"); return 1; }
    const TailWs* twp = getenv("DCV_BENCH_NOTAIL") ? nullptr : &tw;
    const int it = argc > 2 ? atoi(argv[2]) : 20;
    if (argc > 3 && !strcmp(argv[3], "cov")) {   // lagged covariance: X[R,256]^T x (X, X shifted by 10 rows), two B operands
        const int Fc = 256;
        const int64_t P = R - 10;
#ifdef DCV_COVCFG
        using CovCfg = DCV_COVCFG;
#else
        using CovCfg = CfgCovT<false>;
#endif
        for (int64_t kc : {2048, 4096}) {
            int64_t kk = kc;
            for (int64_t c = kc; c >= kc / 2; c -= 32) if (((P + c - 1) / c) % 8 == 0) { kk = c; break; }
            const int64_t splits = (P + kk - 1) / kk;
            float* cslab;
            CK(hipMalloc(&cslab, (size_t)splits * 2 * Fc * Fc * 4));
            Operand op = make_operand(X, Fc, Fc, identity_rows(), getenv("DCV_COV_NOSHIFT") ? nullptr : b1);   // column shift as in the product (dcv_lagged_cov)
            EpiSlab epi{cslab, Fc, Fc, 2, 0, true, splits};
            double ms = time_ms([&] { launch_gemm_cfg<kTN, CovCfg, 2, EpiSlab>(op, op, 10, Fc, Fc, P, kk, epi, s); }, it);
            printf("cov TN 2B kc=%5lld splits=%lld %8.1f us  %6.1f TF\n", (long long)kk, (long long)splits, ms * 1e3, 4.0 * P * Fc * Fc / ms / 1e9);
            dump_stamps("cov", 4);   // stamps are taken by the blockIdx.z == 0 workgroups: the four output tiles of the first contraction chunk
            CK(hipFree(cslab));
        }
        return 0;
    }
    if (argc > 3 && !strcmp(argv[3], "wgrad")) {   // L0 weight gradient: tile shape x split count (chunk lengths need not be stage multiples)
        Operand A = make_operand(dZ, H1, H1), B = make_operand(X, F, F);
        EpiSlab epi{slab, H1, F, 1, 0, true, slab_cap};
        for (int64_t ns : {8, 16, 24, 29, 32, 40, 48, 64}) {
            const int64_t kc = (R + ns - 1) / ns;
            const int64_t kc32 = (kc + 31) / 32 * 32;
            for (int64_t c : {kc, kc32}) {
                if (c == kc32 && kc32 == kc) continue;
                double b = time_ms([&] { launch_gemm_cfg<kTN, CfgBigT<true>, 1, EpiSlab>(A, B, 0, H1, F, R, c, epi, s, nullptr, nullptr); }, it);
                double h = time_ms([&] { launch_gemm_cfg<kTN, CfgHalfMT<true>, 1, EpiSlab>(A, B, 0, H1, F, R, c, epi, s, nullptr, nullptr); }, it);
                double q = time_ms([&] { launch_gemm_cfg<kTN, CfgQuarterT<true>, 1, EpiSlab>(A, B, 0, H1, F, R, c, epi, s, nullptr, nullptr); }, it);
                printf("L0 wgrad want %2lld kc=%5lld splits=%3lld : 128x128 %6.1f us | 64x128 %6.1f us | 64x64 %6.1f us\n", (long long)ns, (long long)c,
                       (long long)((R + c - 1) / c), b * 1e3, h * 1e3, q * 1e3);
            }
        }
        return 0;
    }
    {   // L0 forward: [R,512] x [256,512]^T
        // DCV_LDPAD / DCV_LDPADW: floats added to the row pitch of X / W1 (<= 64): a 2 KiB pitch puts the 128-byte row
        // segments of a stage on very few L2 channels
        const int padx = getenv("DCV_LDPAD") ? atoi(getenv("DCV_LDPAD")) : 0, padw = getenv("DCV_LDPADW") ? atoi(getenv("DCV_LDPADW")) : 0;
        Operand A = make_operand(X, F + padx, F), B = make_operand(W1, F + padw, F);
        EpiBiasAct epi{Hb, H1, b1, DCV_ACT_LEAKY_RELU, true};
        double ms = time_ms([&] { launch_gemm<kNT, EpiBiasAct>(A, B, R, H1, F, 0, epi, s, nullptr, twp); }, it);
        printf("L0 fwd   NT %8.1f us  %6.1f TF\n", ms * 1e3, 2.0 * R * H1 * F / ms / 1e9);
        double ms1 = time_ms_single([&] { launch_gemm<kNT, EpiBiasAct>(A, B, R, H1, F, 0, epi, s, nullptr, twp); }, 20);
        auto t0 = std::chrono::high_resolution_clock::now();
        for (int i = 0; i < 200; ++i) launch_gemm<kNT, EpiBiasAct>(A, B, R, H1, F, 0, epi, s, nullptr, twp);
        auto t1 = std::chrono::high_resolution_clock::now();
        CK(hipDeviceSynchronize());
        printf("L0 fwd   single launch between events %8.1f us ; host time per enqueue (200 back-to-back) %6.1f us\n", ms1 * 1e3,
               std::chrono::duration<double, std::micro>(t1 - t0).count() / 200);
        dump_stamps("L0 fwd", (int)(R / 128 * 2 > 0 ? R / 128 * 2 : 1));
    }
    {   // L0 wgrad: dZ[R,256]^T x X[R,512], split
        Operand A = make_operand(dZ, H1, H1), B = make_operand(X, F, F);
        for (int64_t kc : (R < 32768 ? std::vector<int64_t>{128, 160, 192, 256, 288, 512, 1024} : std::vector<int64_t>{1024, 2048, 4096})) {
            EpiSlab epi{slab, H1, F, 1, 0, true, slab_cap};
            double ms = time_ms([&] { launch_gemm<kTN, EpiSlab>(A, B, H1, F, R, kc, epi, s); }, it);
            printf("L0 wgrad TN kc=%5lld %8.1f us  %6.1f TF\n", (long long)kc, ms * 1e3, 2.0 * R * H1 * F / ms / 1e9);
        }
    }
    {   // L1 forward: [R,256] x [128,256]^T
        Operand A = make_operand(Hb, H1, H1), B = make_operand(W2, H1, H1);
        EpiBiasAct epi{H2b, H2, b1, DCV_ACT_LEAKY_RELU, true};
        double ms = time_ms([&] { launch_gemm<kNT, EpiBiasAct>(A, B, R, H2, H1, 0, epi, s, nullptr, twp); }, it);
        printf("L1 fwd   NT %8.1f us  %6.1f TF\n", ms * 1e3, 2.0 * R * H2 * H1 / ms / 1e9);
        dump_stamps("L1 fwd", (int)(R / 128));
    }
    {   // L1 dgrad: dZ2[R,128] x W2[128,256]
        Operand A = make_operand(H2b, H2, H2), B = make_operand(W2, H1, H1);
        EpiActGrad epi{dZ, H1, Hb, H1, DCV_ACT_LEAKY_RELU, bpart, H1, true};
        double ms = time_ms([&] { launch_gemm<kNN, EpiActGrad>(A, B, R, H1, H2, 0, epi, s, nullptr, twp); }, it);
        printf("L1 dgrad NN %8.1f us  %6.1f TF\n", ms * 1e3, 2.0 * R * H2 * H1 / ms / 1e9);
        dump_stamps("L1 dgrad", (int)(R / 128 * 2));
    }
    {   // L1 wgrad: dZ2[R,128]^T x H1[R,256]
        Operand A = make_operand(H2b, H2, H2), B = make_operand(Hb, H1, H1);
        for (int64_t kc : (R < 32768 ? std::vector<int64_t>{256, 512, 1024} : std::vector<int64_t>{512, 1024, 2048})) {
            EpiSlab epi{slab, H2, H1, 1, 0, true, slab_cap};
            double ms = time_ms([&] { launch_gemm<kTN, EpiSlab>(A, B, H2, H1, R, kc, epi, s); }, it);
            printf("L1 wgrad TN kc=%5lld %8.1f us  %6.1f TF\n", (long long)kc, ms * 1e3, 2.0 * R * H2 * H1 / ms / 1e9);
        }
    }
    return 0;
}
