// Time-lagged covariance sums  A = sum z_t z_t^T,  B = sum z_t z_lag^T  (z = x - shift) with the
// FP32 MFMA engine: the frames axis is the contraction axis, cut into chunks of rows; every
// (output tile, chunk) pair is one workgroup, both products share the x_t fragments, chunk
// partials are summed in float64 in chunk order (deterministic).  The x_lag operand is the
// same matrix `lag` rows further down -- pairing costs no bytes (the reference materialises two
// copies, cv_calculator.py:2247).
#include "gemm_kernels.h"
#include <stdlib.h>

namespace dcv {

constexpr int64_t kCovChunkRows = 2048;   // fp32 accumulation inside a chunk, float64 across chunks: shorter chunks, smaller rounding error (hTICA holds 1e-5)
constexpr size_t kCovMaxSlabBytes = (size_t)2 << 30;

struct CovPlan {
    int nb;           // 1 (lag 0) or 2
    int64_t k_chunk;  // rows per split
    int64_t splits;
    size_t stats_ws;  // bytes of the column-statistics workspace
    size_t sums;      // bytes of the three [4][F] float64 blocks
    size_t slab;      // bytes of the split-K slabs
    size_t asum;      // bytes of the per-chunk column sums of z_t ([splits][F] floats; used when a shift is given)
};

static CovPlan cov_plan(int64_t n_pairs, int F, int lag) {
    CovPlan p;
    p.nb = lag > 0 ? 2 : 1;
    static const int64_t chunk_env = [] { const char* e = getenv("DCV_COV_CHUNK"); return e ? (int64_t)atoll(e) : (int64_t)0; }();   // diagnostic override
    p.k_chunk = chunk_env >= 64 ? chunk_env / 32 * 32 : kCovChunkRows;
    const size_t per_split = (size_t)p.nb * F * F * sizeof(float);
    while (cdiv(n_pairs, p.k_chunk) * per_split > kCovMaxSlabBytes) p.k_chunk *= 2;
    // A split count that is a multiple of 8 lets the kernel's XCD-aware map put the output tiles of one chunk on one XCD,
    // where they share its L2 (every tile of a chunk reads the same rows): measured without it FETCH_SIZE = 3.0 x the
    // matrix, L2 hit rate 4 %, waves waiting 68 % of their cycles (profiles/r02_c3_*).  Shorten the chunk (in steps of 32
    // rows: whole stages) until the count fits.
    for (int64_t kc = p.k_chunk; kc >= p.k_chunk / 2 && kc >= 64; kc -= 32)
        if (cdiv(n_pairs, kc) % 8 == 0) {
            p.k_chunk = kc;
            break;
        }
    // the shortened chunk may push the slabs past the cap again: lengthen in whole stages until they fit (the XCD-aware map
    // then falls back to the plain one when the count is no multiple of 8 -- slower, never wrong)
    while (cdiv(n_pairs, p.k_chunk) * per_split > kCovMaxSlabBytes) p.k_chunk += 32;
    p.splits = cdiv(n_pairs, p.k_chunk);
    p.stats_ws = align_up(dcv_col_stats_workspace(n_pairs, F), 256);
    p.sums = align_up((size_t)3 * 4 * F * sizeof(double), 256);
    p.slab = align_up((size_t)p.splits * per_split, 256);
    p.asum = align_up((size_t)p.splits * F * sizeof(float), 256);
    return p;
}

// out[which][m][n] = sum_z slab[z][which][m][n]   (float64, fixed order)
__global__ void cov_reduce_kernel(const float* __restrict__ slab, int64_t splits, int nb, int64_t FF,
                                  double* __restrict__ outA, double* __restrict__ outB) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= FF) return;
    for (int w = 0; w < nb; ++w) {
        double acc = 0.0;
        int64_t z = 0;
        for (; z + 8 <= splits; z += 8) {   // eight independent loads in flight, added in chunk order (same sum as a serial walk)
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = slab[((z + u) * nb + w) * FF + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += (double)v[u];
        }
        for (; z < splits; ++z) acc += (double)slab[(z * nb + w) * FF + i];
        (w == 0 ? outA : outB)[i] = acc;
    }
    if (nb == 1) outB[i] = 0.0;
}

// a = S_t - P*shift ; b = S_t - head + tail - P*shift   (S_t: sum over rows [0,P))
__global__ void cov_sums_kernel(const double* __restrict__ s_all, const double* __restrict__ s_head,
                                const double* __restrict__ s_tail, const float* __restrict__ shift, int64_t P, int F,
                                int lag, double* __restrict__ a, double* __restrict__ b) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    const double sh = shift ? (double)shift[f] * (double)P : 0.0;
    const double st = s_all[f];
    a[f] = st - sh;
    b[f] = lag > 0 ? (st - s_head[f] + s_tail[f] - sh) : 0.0;
}

// The same two sums from the covariance kernel's own per-chunk column sums of z_t = x_t - shift (float32 inside a chunk
// of <= 2048 centred values, float64 across chunks, chunk order): a = sum_z asum[z] ; b = a - head + tail, where head /
// tail are the sums of the raw `lag` first / last rows (the shift cancels).  Saves the pass over the matrix.
__global__ void cov_sums_fused_kernel(const float* __restrict__ asum, int64_t splits, const double* __restrict__ s_head,
                                      const double* __restrict__ s_tail, int F, int lag, double* __restrict__ a,
                                      double* __restrict__ b) {
    const int f = blockIdx.x;   // one wave per column: lanes over the chunks, fixed-order combine
    const int lane = threadIdx.x;
    double acc = 0.0;
    for (int64_t z = lane; z < splits; z += 64) acc += (double)asum[z * F + f];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) {
        a[f] = acc;
        b[f] = lag > 0 ? (acc - s_head[f] + s_tail[f]) : 0.0;
    }
}

}  // namespace dcv

using namespace dcv;

extern "C" size_t dcv_lagged_cov_workspace(int64_t n_pairs, int32_t F, int32_t lag) {
    if (n_pairs <= 0 || F <= 0 || lag < 0) return 0;
    const CovPlan p = cov_plan(n_pairs, F, lag);
    return p.stats_ws + p.sums + p.slab + p.asum;
}

extern "C" int dcv_lagged_cov(const float* X_d, int64_t n_pairs, int32_t F, int64_t ld, int32_t lag,
                              const float* shift_d, double* out_d, void* ws_d, size_t ws_bytes, void* stream) {
    DCV_REQUIRE(X_d && out_d && n_pairs > 0 && F > 0 && ld >= F && lag >= 0, "dcv_lagged_cov: bad arguments");
    DCV_REQUIRE(ws_d && ws_bytes >= dcv_lagged_cov_workspace(n_pairs, F, lag), "dcv_lagged_cov: workspace too small");
    hipStream_t s = as_stream(stream);
    const CovPlan p = cov_plan(n_pairs, F, lag);
    char* ws = static_cast<char*>(ws_d);
    void* stats_ws = ws;
    double* sums = reinterpret_cast<double*>(ws + p.stats_ws);
    float* slab = reinterpret_cast<float*>(ws + p.stats_ws + p.sums);
    double* a = out_d;
    double* b = out_d + F;
    double* A = out_d + 2 * (int64_t)F;
    double* B = A + (int64_t)F * F;

    float* asum = reinterpret_cast<float*>(ws + p.stats_ws + p.sums + p.slab);
    // With a shift (the standardised / centred case) the column sums of z_t fall out of the covariance kernel itself;
    // without one the values are not centred and a float32 chunk sum would cost digits: the float64 statistics pass stays.
    static const bool fuse_off = [] { const char* e = getenv("DCV_COV_FUSED_SUMS"); return e && e[0] == '0'; }();
    const bool fused_sums = shift_d != nullptr && !fuse_off;
    int rc = DCV_OK;
    if (!fused_sums) {
        // column sums of x_t over all pairs
        rc = dcv_col_stats(X_d, n_pairs, F, ld, sums, stats_ws, p.stats_ws, stream);
        if (rc) return rc;
    }
    if (lag > 0) {   // the `lag` head / tail rows that turn the x_t sums into the x_lag sums
        rc = dcv_col_stats(X_d, lag, F, ld, sums + 4 * F, stats_ws, p.stats_ws, stream);
        if (rc) return rc;
        rc = dcv_col_stats(X_d + n_pairs * ld, lag, F, ld, sums + 8 * F, stats_ws, p.stats_ws, stream);
        if (rc) return rc;
    }
    if (!fused_sums) {
        hipLaunchKernelGGL(cov_sums_kernel, dim3((F + 255) / 256), dim3(256), 0, s, sums, sums + 4 * F, sums + 8 * F, shift_d,
                           n_pairs, F, lag, a, b);
        DCV_CHECK_LAUNCH();
    }

    // Always the FP32-input MFMA, whatever dcv_set_gemm_mode says: a covariance is a sum of thousands of same-sign
    // products per accumulator and the BF16 matrix pipe adds into its accumulator by truncation -- measured bias
    // -1e-5 relative at 4096 rows (tools/split_bench.hip) against an unbiased +-3e-6 here, and TICA eigenvectors are
    // held to 1e-5.
    const Operand op = make_operand(X_d, ld, F, identity_rows(), shift_d);
    EpiSlab epi{slab, F, F, p.nb, 0, quad_ok(slab, F), p.splits};   // p.splits slabs are what cov_plan sized the workspace for
    if (fused_sums) {
        EpiSlabSum es{epi, asum};
        if (p.nb == 2)
            rc = launch_gemm_cfg<kTN, CfgCovT<false>, 2, EpiSlabSum>(op, op, lag, F, F, n_pairs, p.k_chunk, es, s);
        else
            rc = launch_gemm_cfg<kTN, CfgBigT<false>, 1, EpiSlabSum>(op, op, 0, F, F, n_pairs, p.k_chunk, es, s);
        if (rc) return rc;
        hipLaunchKernelGGL(cov_sums_fused_kernel, dim3(F), dim3(64), 0, s, (const float*)asum, p.splits, (const double*)(sums + 4 * F),
                           (const double*)(sums + 8 * F), F, lag, a, b);
        DCV_CHECK_LAUNCH();
    } else {
        if (p.nb == 2)
            rc = launch_gemm_cfg<kTN, CfgCovT<false>, 2, EpiSlab>(op, op, lag, F, F, n_pairs, p.k_chunk, epi, s);
        else
            rc = launch_gemm_cfg<kTN, CfgBigT<false>, 1, EpiSlab>(op, op, 0, F, F, n_pairs, p.k_chunk, epi, s);
        if (rc) return rc;
    }
    const int64_t FF = (int64_t)F * F;
    hipLaunchKernelGGL(cov_reduce_kernel, dim3((unsigned)cdiv(FF, 256)), dim3(256), 0, s, slab, p.splits, p.nb, FF, A, B);
    DCV_CHECK_LAUNCH();
    return DCV_OK;
}
