// Developer check + micro-benchmark of the plane-operand NT kernels (TileCfg::PL, gemm.h) against the in-register
// split kernel they must reproduce bit for bit (not part of the product build):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I deep_cartograph_amd/csrc -I include tools/planes_bench.hip \
//         deep_cartograph_amd/csrc/common.hip -o tools/planes_bench
//   ./planes_bench [rows=131072] [iters=20] [N=256] [K=512] [activation id = leaky_relu]
#include "gemm_kernels.h"
#include <vector>
#include <cstdlib>
#include <cstring>

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

static size_t mismatches(const std::vector<float>& a, const std::vector<float>& b, double* maxd) {
    size_t n = 0;
    *maxd = 0;
    for (size_t i = 0; i < a.size(); ++i) {
        if (memcmp(&a[i], &b[i], 4) != 0) {
            ++n;
            const double d = fabs((double)a[i] - (double)b[i]);
            if (!(d <= *maxd)) *maxd = d;
        }
    }
    return n;
}

int main(int argc, char** argv) {
    const int64_t M = argc > 1 ? atoll(argv[1]) : 131072;
    const int it = argc > 2 ? atoi(argv[2]) : 20;
    const int64_t N = argc > 3 ? atoll(argv[3]) : 256;
    const int64_t K = argc > 4 ? atoll(argv[4]) : 512;
    const int act = argc > 5 ? atoi(argv[5]) : DCV_ACT_LEAKY_RELU;
    set_gemm_split(true);
    float *A, *B, *C0, *C1, *bias, *Ap, *Bp;
    CK(hipMalloc(&A, (size_t)M * K * 4));
    CK(hipMalloc(&B, (size_t)N * K * 4));
    CK(hipMalloc(&C0, (size_t)M * N * 4));
    CK(hipMalloc(&C1, (size_t)M * N * 4));
    CK(hipMalloc(&bias, (size_t)N * 4));
    CK(hipMalloc(&Ap, planes_bytes(M, K)));
    CK(hipMalloc(&Bp, planes_bytes(N, K)));
    std::vector<float> h((size_t)M * K), hb((size_t)N * K), hc0((size_t)M * N), hc1((size_t)M * N);
    srand(7);
    for (auto& v : h) v = ((float)rand() / RAND_MAX - 0.5f) * 3.f;
    for (auto& v : hb) v = ((float)rand() / RAND_MAX - 0.5f) * 0.2f;
    CK(hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(bias, hb.data(), (size_t)N * 4, hipMemcpyHostToDevice));
    hipStream_t s = 0;
    if (launch_split_planes<false>(A, K, M, K, Ap, M, s) || launch_split_planes<false>(B, K, N, K, Bp, N, s)) { printf("split failed: %s\n", dcv_last_error()); return 1; }
    CK(hipDeviceSynchronize());
    const double ms_split = time_ms([&] { launch_split_planes<false>(A, K, M, K, Ap, M, s); }, 3);
    printf("split_planes %lld x %lld: %.1f us (%.2f TB/s)\n", (long long)M, (long long)K, ms_split * 1e3, (double)M * K * 10.0 / ms_split / 1e9);

    const Operand a = make_operand(A, K, K), b = make_operand(B, K, K);
    const Operand ap = make_plane_operand(Ap, K), bp = make_plane_operand(Bp, K);
    EpiBiasAct e0{C0, N, bias, act, true};
    EpiBiasAct e1{C1, N, bias, act, true};
    const double flop = 2.0 * M * N * K;
    auto report = [&](const char* what, double ms) { printf("%-34s %9.1f us %7.1f TF\n", what, ms * 1e3, flop / ms / 1e9); };
    auto check = [&](const char* what) {
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(hc1.data(), C1, hc1.size() * 4, hipMemcpyDeviceToHost));
        double md;
        const size_t n = mismatches(hc0, hc1, &md);
        printf("  %s vs in-register split: %zu of %zu elements differ (max |d| %.3g)\n", what, n, hc0.size(), md);
    };
    int rc = launch_gemm<kNT, EpiBiasAct>(a, b, M, N, K, 0, e0, s);
    if (rc) { printf("reference launch failed: %s\n", dcv_last_error()); return 1; }
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(hc0.data(), C0, hc0.size() * 4, hipMemcpyDeviceToHost));
    report("in-register split (PL=0)", time_ms([&] { launch_gemm<kNT, EpiBiasAct>(a, b, M, N, K, 0, e0, s); }, it));

    if (getenv("DCV_LS_BIG")) {   // split-from-LDS (TileCfg::PL bit 4) on 128 x 128 tiles with 16-deep stages: 80 KB of LDS, two workgroups per CU
        CK(hipMemset(C1, 0xff, (size_t)M * N * 4));
        rc = launch_gemm_ls<TileCfg<2, 2, 2, 2, 16, 2, true, 28>, EpiBiasAct>(a, b, M, N, K, e1, s, nullptr, nullptr);
        if (rc) { printf("LS big launch failed (%d): %s\n", rc, dcv_last_error()); return 1; }
        check("split-from-LDS 128 x 128");
        report("split-from-LDS 128 x 128, KB 16", time_ms([&] { launch_gemm_ls<TileCfg<2, 2, 2, 2, 16, 2, true, 28>, EpiBiasAct>(a, b, M, N, K, e1, s, nullptr, nullptr); }, it));
        report("in-register 128 x 128, KB 32", time_ms([&] { launch_gemm_cfg<kNT, CfgBigT<true>, 1, EpiBiasAct>(a, b, 0, M, N, K, 0, e0, s, nullptr, nullptr); }, it));
        report("in-register 128 x 128, KB 16", time_ms([&] { launch_gemm_cfg<kNT, TileCfg<2, 2, 2, 2, 16, 2, true>, 1, EpiBiasAct>(a, b, 0, M, N, K, 0, e0, s, nullptr, nullptr); }, it));
    }
    if (getenv("DCV_OCC_PROBE")) {   // the same flops as 64 x 64 tiles at 2, 4 and 8 workgroups per CU: rows doubled, contraction halved
        for (int f = 1; f <= 4; f *= 2) {
            const int64_t Mf = 8192 * f, Kf = 512 / f;
            float *Af, *Cf;
            CK(hipMalloc(&Af, (size_t)Mf * Kf * 4));
            CK(hipMalloc(&Cf, (size_t)Mf * N * 4));
            CK(hipMemcpy(Af, h.data(), (size_t)8192 * 512 * 4, hipMemcpyHostToDevice));
            const Operand af = make_operand(Af, Kf, Kf), bf = make_operand(B, Kf, Kf);
            EpiBiasAct ef{Cf, N, bias, act, true};
            const double ms = time_ms([&] { launch_gemm_cfg<kNT, CfgQuarterT<true>, 1, EpiBiasAct>(af, bf, 0, Mf, N, Kf, 0, ef, s, nullptr, nullptr); }, it);
            printf("occupancy probe: %lld x %lld x %lld, 64 x 64 tiles, %lld workgroups (%d per CU), %d stages each: %.1f us\n", (long long)Mf, (long long)N,
                   (long long)Kf, (long long)(Mf / 64 * (N / 64)), (int)(Mf / 64 * (N / 64) / 256), (int)(Kf / 32), ms * 1e3);
            const double msh = time_ms([&] { launch_gemm_cfg<kNT, CfgHalfMT<true>, 1, EpiBiasAct>(af, bf, 0, Mf, N, Kf, 0, ef, s, nullptr, nullptr); }, it);
            printf("occupancy probe: %lld x %lld x %lld, 64 x 128 tiles, %lld workgroups, %d stages each: %.1f us\n", (long long)Mf, (long long)N, (long long)Kf,
                   (long long)(Mf / 64 * (N / 128)), (int)(Kf / 32), msh * 1e3);
            const double msb = time_ms([&] { launch_gemm_cfg<kNT, CfgBigT<true>, 1, EpiBiasAct>(af, bf, 0, Mf, N, Kf, 0, ef, s, nullptr, nullptr); }, it);
            printf("occupancy probe: %lld x %lld x %lld, 128 x 128 tiles, %lld workgroups, %d stages each: %.1f us\n", (long long)Mf, (long long)N, (long long)Kf,
                   (long long)(Mf / 128 * (N / 128)), (int)(Kf / 32), msb * 1e3);
            CK(hipFree(Af));
            CK(hipFree(Cf));
        }
    }
    {   // contraction-split tail tile (GemmDims::tail_split): only the rows of the ragged last tile may differ, by rounding
        TailWs tw;
        if (!alloc_tail_ws(&tw, 8)) { printf("tail workspace: allocation failed\n"); return 1; }
        CK(hipMemset(C1, 0xff, (size_t)M * N * 4));
        rc = launch_gemm<kNT, EpiBiasAct>(a, b, M, N, K, 0, e1, s, nullptr, &tw);
        if (rc) { printf("tail launch failed: %s\n", dcv_last_error()); return 1; }
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(hc1.data(), C1, hc1.size() * 4, hipMemcpyDeviceToHost));
        size_t nd = 0, first = hc0.size();
        double md = 0;
        for (size_t i = 0; i < hc0.size(); ++i)
            if (memcmp(&hc0[i], &hc1[i], 4) != 0) {
                if (first == hc0.size()) first = i;
                ++nd;
                const double dd = fabs((double)hc0[i] - (double)hc1[i]);
                if (!(dd <= md)) md = dd;
            }
        printf("  tail k-split vs plain: %zu elements differ (first in row %zu of %lld, max |d| %.3g)\n", nd, first / (size_t)N, (long long)M, md);
        report("in-register split + tail k-split", time_ms([&] { launch_gemm<kNT, EpiBiasAct>(a, b, M, N, K, 0, e1, s, nullptr, &tw); }, it));
        for (int r = 0; r < 3; ++r) {   // repeated launches reuse the tickets
            launch_gemm<kNT, EpiBiasAct>(a, b, M, N, K, 0, e1, s, nullptr, &tw);
        }
        CK(hipDeviceSynchronize());
        std::vector<float> hc2(hc1.size());
        CK(hipMemcpy(hc2.data(), C1, hc2.size() * 4, hipMemcpyDeviceToHost));
        printf("  tail k-split run-to-run: %s\n", memcmp(hc1.data(), hc2.data(), hc1.size() * 4) == 0 ? "bit-identical" : "DIFFERENT");
        if (getenv("DCV_TAIL_STRESS")) {   // alternate two different A operands: a stale partial of the other one would change bits
            float* A2;
            CK(hipMalloc(&A2, (size_t)M * K * 4));
            std::vector<float> h2(h.size());
            for (size_t i = 0; i < h.size(); ++i) h2[i] = h[i] * 1.7f + 0.3f;
            CK(hipMemcpy(A2, h2.data(), h2.size() * 4, hipMemcpyHostToDevice));
            const Operand a2 = make_operand(A2, K, K);
            const size_t tail0 = (size_t)(M / 64 * 64) * N, ntail = hc1.size() - tail0;
            std::vector<float> refA(hc1.begin() + tail0, hc1.end()), refB(ntail), got(ntail);
            launch_gemm<kNT, EpiBiasAct>(a2, b, M, N, K, 0, e1, s, nullptr, &tw);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(refB.data(), C1 + tail0, ntail * 4, hipMemcpyDeviceToHost));
            const int iters = atoi(getenv("DCV_TAIL_STRESS"));
            int bad = 0;
            for (int r = 0; r < iters; ++r) {
                const bool useA = (r * 7 % 3) != 0;
                launch_gemm<kNT, EpiBiasAct>(useA ? a : a2, b, M, N, K, 0, e1, s, nullptr, &tw);
                CK(hipMemcpyAsync(got.data(), C1 + tail0, ntail * 4, hipMemcpyDeviceToHost, s));
                CK(hipStreamSynchronize(s));
                if (memcmp(got.data(), (useA ? refA : refB).data(), ntail * 4) != 0) ++bad;
            }
            printf("  tail stress: %d of %d launches with alternating operands differ from their reference\n", bad, iters);
        }
    }
    {   // the FP32-input MFMA flavour of the same product: agreement to fp32 rounding
        set_gemm_split(false);
        CK(hipMemset(C1, 0xff, (size_t)M * N * 4));
        rc = launch_gemm<kNT, EpiBiasAct>(a, b, M, N, K, 0, e1, s);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(hc1.data(), C1, hc1.size() * 4, hipMemcpyDeviceToHost));
        double md = 0, mx = 0;
        size_t worst = 0;
        for (size_t i = 0; i < hc0.size(); ++i) {
            const double dd = fabs((double)hc0[i] - (double)hc1[i]);
            if (!(dd <= md)) { md = dd; worst = i; }
            if (fabs((double)hc0[i]) > mx) mx = fabs((double)hc0[i]);
        }
        printf("  fp32-input MFMA vs split: max |d| %.3g (largest |value| %.3g) at row %zu col %zu\n", md, mx, worst / (size_t)N, worst % (size_t)N);
        // both against float64 on the first rows
        double e_split = 0, e_native = 0, s_split = 0, s_native = 0;
        const int64_t rows = M < 192 ? M : 192;
        for (int64_t r = 0; r < rows; ++r)
            for (int64_t c = 0; c < N; ++c) {
                double acc = 0;
                for (int64_t k = 0; k < K; ++k) acc += (double)h[r * K + k] * (double)hb[c * K + k];
                acc += (double)hb[c];
                double ref = act == DCV_ACT_LEAKY_RELU ? (acc > 0 ? acc : 0.01 * acc) : acc;
                const double ds = (double)hc0[r * N + c] - ref, dn = (double)hc1[r * N + c] - ref;
                if (fabs(ds) > e_split) e_split = fabs(ds);
                if (fabs(dn) > e_native) e_native = fabs(dn);
                s_split += ds; s_native += dn;
            }
        printf("  vs float64 (first %lld rows): split max |err| %.3g mean err %.3g ; fp32-input MFMA max |err| %.3g mean err %.3g\n", (long long)rows, e_split,
               s_split / (rows * N), e_native, s_native / (rows * N));
        set_gemm_split(true);
    }
    if (getenv("DCV_CHECK_BACKWARD")) {   // dgrad (NN + activation-gradient epilogue) and wgrad (TN, split-K slabs): fp32-input MFMA vs split flavour
        const int64_t K2 = 128;   // dZ[M, K2] . W[K2, N] -> dX[M, N] ;  dZ[M, K2]^T . Hin[M, N] -> dW[K2, N]
        float *dZ, *W, *dX0, *dX1, *bp0, *bp1, *slab0, *slab1;
        const int64_t kc = 288, splits = (M + kc - 1) / kc;
        CK(hipMalloc(&dZ, (size_t)M * K2 * 4));
        CK(hipMalloc(&W, (size_t)K2 * N * 4));
        CK(hipMalloc(&dX0, (size_t)M * N * 4));
        CK(hipMalloc(&dX1, (size_t)M * N * 4));
        CK(hipMalloc(&bp0, (size_t)(M / 32 + 8) * N * 4));
        CK(hipMalloc(&bp1, (size_t)(M / 32 + 8) * N * 4));
        CK(hipMalloc(&slab0, (size_t)splits * K2 * N * 4));
        CK(hipMalloc(&slab1, (size_t)splits * K2 * N * 4));
        CK(hipMemcpy(dZ, h.data(), (size_t)M * K2 * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(W, hb.data(), (size_t)K2 * N * 4, hipMemcpyHostToDevice));
        const Operand ad = make_operand(dZ, K2, K2), bd = make_operand(W, N, N), hin = make_operand(C0, N, N);
        int tm0 = 0, tm1 = 0;
        EpiActGrad g0{dX0, N, C0, N, DCV_ACT_LEAKY_RELU, bp0, N, true}, g1{dX1, N, C0, N, DCV_ACT_LEAKY_RELU, bp1, N, true};
        set_gemm_split(true);
        rc = launch_gemm<kNN, EpiActGrad>(ad, bd, M, N, K2, 0, g0, s, &tm0);
        EpiSlab s0{slab0, K2, N, 1, 0, true, splits};
        rc |= launch_gemm<kTN, EpiSlab>(ad, hin, K2, N, M, kc, s0, s);
        set_gemm_split(false);
        rc |= launch_gemm<kNN, EpiActGrad>(ad, bd, M, N, K2, 0, g1, s, &tm1);
        EpiSlab s1{slab1, K2, N, 1, 0, true, splits};
        rc |= launch_gemm<kTN, EpiSlab>(ad, hin, K2, N, M, kc, s1, s);
        set_gemm_split(true);
        if (rc) { printf("backward check launch failed: %s\n", dcv_last_error()); return 1; }
        CK(hipDeviceSynchronize());
        auto cmp = [&](const char* what, const float* d0, const float* d1, size_t n) {
            std::vector<float> a0(n), a1(n);
            CK(hipMemcpy(a0.data(), d0, n * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(a1.data(), d1, n * 4, hipMemcpyDeviceToHost));
            double md = 0, mx = 0;
            size_t w = 0;
            for (size_t i = 0; i < n; ++i) {
                const double dd = fabs((double)a0[i] - (double)a1[i]);
                if (!(dd <= md)) { md = dd; w = i; }
                if (fabs((double)a0[i]) > mx) mx = fabs((double)a0[i]);
            }
            printf("  %s: fp32-input MFMA vs split max |d| %.3g of %.3g (rel %.2e) at element %zu\n", what, md, mx, md / mx, w);
        };
        cmp("dgrad dX", dX0, dX1, (size_t)M * N);
        printf("  (row tiles %d / %d)\n", tm0, tm1);
        cmp("dgrad bias partials", bp0, bp1, (size_t)(tm0 < tm1 ? tm0 : tm1) * N);
        // slabs summed on the host
        std::vector<float> sl0((size_t)splits * K2 * N), sl1(sl0.size());
        CK(hipMemcpy(sl0.data(), slab0, sl0.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(sl1.data(), slab1, sl1.size() * 4, hipMemcpyDeviceToHost));
        double md = 0, mx = 0;
        for (size_t i = 0; i < (size_t)K2 * N; ++i) {
            double a0 = 0, a1 = 0;
            for (int64_t z = 0; z < splits; ++z) { a0 += sl0[z * K2 * N + i]; a1 += sl1[z * K2 * N + i]; }
            if (fabs(a0 - a1) > md) md = fabs(a0 - a1);
            if (fabs(a0) > mx) mx = fabs(a0);
        }
        printf("  wgrad (kc %lld, %lld splits): fp32-input MFMA vs split max |d| %.3g of %.3g (rel %.2e)\n", (long long)kc, (long long)splits, md, mx, md / mx);
    }
    CK(hipMemset(C1, 0xff, (size_t)M * N * 4));
    rc = launch_gemm_planes<3, EpiBiasAct>(ap, bp, M, N, K, e1, s);
    if (rc) { printf("PL=3 launch: rc=%d %s\n", rc, rc < 0 ? dcv_last_error() : "(not applicable)"); }
    else {
        check("PL=3");
        report("A, B planes (PL=3)", time_ms([&] { launch_gemm_planes<3, EpiBiasAct>(ap, bp, M, N, K, e1, s); }, it));
    }
    CK(hipMemset(C1, 0xff, (size_t)M * N * 4));
    rc = launch_gemm_planes<2, EpiBiasAct>(a, bp, M, N, K, e1, s);
    if (rc) { printf("PL=2 launch: rc=%d %s\n", rc, rc < 0 ? dcv_last_error() : "(not applicable)"); }
    else {
        check("PL=2");
        report("B planes (PL=2)", time_ms([&] { launch_gemm_planes<2, EpiBiasAct>(a, bp, M, N, K, e1, s); }, it));
    }
#ifdef DCV_PL1
    CK(hipMemset(C1, 0xff, (size_t)M * N * 4));
    rc = launch_gemm_planes<1, EpiBiasAct>(ap, b, M, N, K, e1, s);
    if (rc) { printf("PL=1 launch: rc=%d %s\n", rc, rc < 0 ? dcv_last_error() : "(not applicable)"); }
    else {
        check("PL=1");
        report("A planes (PL=1)", time_ms([&] { launch_gemm_planes<1, EpiBiasAct>(ap, b, M, N, K, e1, s); }, it));
    }
#endif
    return 0;
}
