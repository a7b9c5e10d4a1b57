// k-selection scores of statistics.optimize_clustering (reference statistics.py:59-93) on the GPU
// (SURVEY section 8 f4): Calinski-Harabasz and Davies-Bouldin are two streaming passes over the
// points (per-label sums / counts, then dispersions about the label means); the silhouette is the
// exact O(n^2) all-pairs form -- every point's summed Euclidean distance to every cluster -- so it is
// compute-bound on the float64 vector pipe and practical up to a few million points per GPU.
// float64 throughout (the points are the CSV values the reference clusters).
//
// Determinism: per-wave LDS accumulators combined in wave then block order, as in kmeans.hip.
#include "common.h"

namespace dcv {

constexpr int kScThreads = 256;
constexpr int kScMaxD = 16;
constexpr int kScMaxK = 64;
constexpr int kScMaxBlocks = 1024;

static int sc_blocks(int64_t n) {
    int64_t b = cdiv(n, (int64_t)kScThreads * 4);
    const int64_t cap = (int64_t)num_cus() * 4;
    if (b > cap) b = cap;
    if (b > kScMaxBlocks) b = kScMaxBlocks;
    if (b < 1) b = 1;
    return (int)b;
}

// part[block] = [sums k*d | counts k | sum ||x - c_label||^2 k | sum ||x - c_label|| k]; without centres
// only the first two groups are meaningful
__global__ __launch_bounds__(kScThreads) void label_stats_kernel(const double* __restrict__ P, int64_t n, int d,
                                                                 const int32_t* __restrict__ labels,
                                                                 const double* __restrict__ centers, int k,
                                                                 double* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int W = k * d + 3 * k;
    double* s_c = reinterpret_cast<double*>(smem);   // [k][d]
    double* s_acc = s_c + k * d;                      // [4 waves][W]
    const int t = threadIdx.x, wave = t >> 6;
    for (int i = t; i < k * d; i += kScThreads) s_c[i] = centers ? centers[i] : 0.0;
    for (int i = t; i < 4 * W; i += kScThreads) s_acc[i] = 0.0;
    __syncthreads();
    double* my = s_acc + wave * W;
    const int64_t per_block = (n + gridDim.x - 1) / gridDim.x;
    const int64_t begin = (int64_t)blockIdx.x * per_block;
    const int64_t end = begin + per_block < n ? begin + per_block : n;
    for (int64_t i = begin + t; i < end; i += kScThreads) {
        const int lab = labels[i];
        if (lab < 0 || lab >= k) continue;   // noise labels (-1) of other algorithms are not scored
        double ss = 0.0;
        for (int c = 0; c < d; ++c) {
            const double x = P[i * d + c];
            atomicAdd(&my[lab * d + c], x);
            const double df = x - s_c[lab * d + c];
            ss += df * df;
        }
        atomicAdd(&my[k * d + lab], 1.0);
        if (centers) {
            atomicAdd(&my[k * d + k + lab], ss);
            atomicAdd(&my[k * d + 2 * k + lab], sqrt(ss));
        }
    }
    __syncthreads();
    double* out = part + (int64_t)blockIdx.x * W;
    for (int i = t; i < W; i += kScThreads) out[i] = ((s_acc[i] + s_acc[W + i]) + s_acc[2 * W + i]) + s_acc[3 * W + i];
}

// acc[i] = sum over the blocks of part[b][i]: one workgroup per output, threads take b = t, t + 256, ... (independent
// loads, all in flight), fixed LDS tree -- a single thread walking the partials pays one memory latency per partial.
__global__ __launch_bounds__(256) void sum_blocks_kernel(const double* __restrict__ part, int nblocks, int width, double* __restrict__ acc) {
    __shared__ double s_red[256];
    const int i = blockIdx.x, t = threadIdx.x;
    double s = 0.0;
    for (int b = t; b < nblocks; b += 256) s += part[(int64_t)b * width + i];
    s_red[t] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) s_red[t] += s_red[t + o];
        __syncthreads();
    }
    if (t == 0) acc[i] = s_red[0];
}

// S[i][c] = sum over the points j of cluster c of ||Q_i - P_j||.  P is sorted by cluster (cluster c
// occupies rows [start[c], start[c+1])); Q are the query points (this rank's block, any order).
// grid = (query tiles, clusters); the cluster's points stream through LDS in tiles.
constexpr int kSilTile = 1024;
template <int D>
__global__ __launch_bounds__(kScThreads) void cluster_dist_sums_kernel(const double* __restrict__ Q, int64_t nq,
                                                                       const double* __restrict__ P,
                                                                       const int64_t* __restrict__ start, int k, int d_rt,
                                                                       double* __restrict__ S) {
    __shared__ double s_p[kSilTile * (D > 0 ? D : kScMaxD)];
    const int d = D > 0 ? D : d_rt;
    const int c = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * kScThreads + threadIdx.x;
    double x[D > 0 ? D : kScMaxD];
#pragma unroll
    for (int q = 0; q < (D > 0 ? D : kScMaxD); ++q) x[q] = (q < d && i < nq) ? Q[i * d + q] : 0.0;
    const int64_t j0 = start[c], j1 = start[c + 1];
    double acc = 0.0;
    for (int64_t jb = j0; jb < j1; jb += kSilTile) {
        const int m = (int)(jb + kSilTile < j1 ? kSilTile : j1 - jb);
        __syncthreads();
        for (int q = threadIdx.x; q < m * d; q += kScThreads) s_p[q] = P[jb * d + q];
        __syncthreads();
        for (int j = 0; j < m; ++j) {
            double ss = 0.0;
#pragma unroll
            for (int q = 0; q < (D > 0 ? D : kScMaxD); ++q)
                if (q < d) {
                    const double df = x[q] - s_p[j * d + q];
                    ss += df * df;
                }
            acc += sqrt(ss);
        }
    }
    if (i < nq) S[i * k + c] = acc;
}

// silhouette sample values from S (sklearn silhouette_samples): a = S[i][own] / (n_own - 1),
// b = min over the other clusters of S[i][c] / n_c, s = (b - a) / max(a, b), 0 for singletons;
// part[block] = sum of s over the block's queries (thread order, then lane / wave order)
__global__ __launch_bounds__(kScThreads) void silhouette_sum_kernel(const double* __restrict__ S, int64_t nq, int k,
                                                                    const int32_t* __restrict__ qlabels,
                                                                    const int64_t* __restrict__ start,
                                                                    double* __restrict__ part) {
    __shared__ double red[kScThreads];
    const int t = threadIdx.x;
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kScThreads + t; i < nq; i += (int64_t)gridDim.x * kScThreads) {
        const int own = qlabels[i];
        if (own < 0 || own >= k) continue;
        const double n_own = (double)(start[own + 1] - start[own]);
        double b = INFINITY;
        for (int c = 0; c < k; ++c) {
            const double nc = (double)(start[c + 1] - start[c]);
            if (c != own && nc > 0.0) {
                const double v = S[i * k + c] / nc;
                if (v < b) b = v;
            }
        }
        if (n_own > 1.0 && b < INFINITY) {
            const double a = S[i * k + own] / (n_own - 1.0);
            const double mx = a > b ? a : b;
            if (mx > 0.0) s += (b - a) / mx;
        }
    }
    red[t] = s;
    __syncthreads();
    if (t == 0) {
        double tot = 0.0;
        for (int q = 0; q < kScThreads; ++q) tot += red[q];
        part[blockIdx.x] = tot;
    }
}

// ------------------------------------------------------------------ linear binning (first stage of the binned KDE behind the FES)
// Every point spreads unit weight over the 2^d grid nodes around it, proportionally to proximity (KDEpy's linear binning).
// Weights are accumulated as 64-bit fixed point (2^-36 resolution): integer atomics commute, so the grid is the same
// whatever the order of the threads.  1-D grids that fit LDS are built per block first.
constexpr double kBinScale = 68719476736.0;   // 2^36
template <int D>
__global__ __launch_bounds__(kScThreads) void linear_binning_kernel(const double* __restrict__ P, int64_t n, int64_t ldp, int c0, int c1,
                                                                    double lo0, double inv0, double lo1, double inv1, int bins,
                                                                    unsigned long long* __restrict__ grid, unsigned long long* __restrict__ outside) {
    extern __shared__ unsigned long long s_grid[];
    const bool local = D == 1 && bins <= 4096;
    if (local) {
        for (int i = threadIdx.x; i < bins; i += kScThreads) s_grid[i] = 0ull;
        __syncthreads();
    }
    unsigned long long miss = 0ull;
    for (int64_t i = (int64_t)blockIdx.x * kScThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kScThreads) {
        const double t0 = (P[i * ldp + c0] - lo0) * inv0;   // position in grid-spacing units
        if (!(t0 >= 0.0 && t0 <= (double)(bins - 1))) { ++miss; continue; }
        int i0 = (int)t0;
        if (i0 > bins - 2) i0 = bins - 2;
        const double f0 = t0 - (double)i0;
        if constexpr (D == 1) {
            const unsigned long long w1 = (unsigned long long)(f0 * kBinScale + 0.5), w0 = (unsigned long long)kBinScale - w1;
            if (local) {
                atomicAdd(&s_grid[i0], w0);
                atomicAdd(&s_grid[i0 + 1], w1);
            } else {
                atomicAdd(&grid[i0], w0);
                atomicAdd(&grid[i0 + 1], w1);
            }
        } else {
            const double t1 = (P[i * ldp + c1] - lo1) * inv1;
            if (!(t1 >= 0.0 && t1 <= (double)(bins - 1))) { ++miss; continue; }
            int j0 = (int)t1;
            if (j0 > bins - 2) j0 = bins - 2;
            const double f1 = t1 - (double)j0;
            const unsigned long long w11 = (unsigned long long)(f0 * f1 * kBinScale + 0.5);
            const unsigned long long w10 = (unsigned long long)(f0 * (1.0 - f1) * kBinScale + 0.5);
            const unsigned long long w01 = (unsigned long long)((1.0 - f0) * f1 * kBinScale + 0.5);
            const unsigned long long w00 = (unsigned long long)kBinScale - w11 - w10 - w01;
            unsigned long long* g = grid + (int64_t)i0 * bins + j0;   // grid[i][j]: first coordinate = row
            atomicAdd(g, w00);
            atomicAdd(g + 1, w01);
            atomicAdd(g + bins, w10);
            atomicAdd(g + bins + 1, w11);
        }
    }
    if (miss) atomicAdd(outside, miss);
    if (local) {
        __syncthreads();
        for (int i = threadIdx.x; i < bins; i += kScThreads)
            if (s_grid[i]) atomicAdd(&grid[i], s_grid[i]);
    }
}

__global__ void binning_to_double_kernel(const unsigned long long* __restrict__ fixed, int64_t cells, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cells) out[i] = (double)fixed[i] / kBinScale;
}

}  // namespace dcv

using namespace dcv;

extern "C" size_t dcv_label_stats_workspace(int64_t n, int32_t d, int32_t k) {
    return (size_t)sc_blocks(n) * (size_t)(k * d + 3 * k) * sizeof(double);
}

extern "C" int dcv_label_stats(const double* P_d, int64_t n, int32_t d, const int32_t* labels_d, const double* centers_d,
                               int32_t k, double* acc_d, void* ws_d, size_t ws_bytes, void* stream) {
    DCV_REQUIRE(P_d && labels_d && acc_d && ws_d, "dcv_label_stats: null argument");
    DCV_REQUIRE(n >= 1 && d >= 1 && d <= kScMaxD && k >= 1 && k <= kScMaxK, "dcv_label_stats: n=%lld d=%d k=%d out of range (d <= 16, k <= 64)",
                (long long)n, d, k);
    DCV_REQUIRE(ws_bytes >= dcv_label_stats_workspace(n, d, k), "dcv_label_stats: workspace too small");
    hipStream_t s = as_stream(stream);
    const int nb = sc_blocks(n);
    const int W = k * d + 3 * k;
    const size_t lds = ((size_t)k * d + 4 * (size_t)W) * sizeof(double);
    hipLaunchKernelGGL(label_stats_kernel, dim3(nb), dim3(kScThreads), lds, s, P_d, n, d, labels_d, centers_d, k, static_cast<double*>(ws_d));
    DCV_CHECK_LAUNCH();
    hipLaunchKernelGGL(sum_blocks_kernel, dim3((unsigned)W), dim3(256), 0, s, static_cast<const double*>(ws_d), nb, W, acc_d);
    DCV_CHECK_LAUNCH();
    return DCV_OK;
}

extern "C" int dcv_cluster_dist_sums(const double* Q_d, int64_t nq, const double* Psorted_d, const int64_t* start_d, int32_t k,
                                     int32_t d, double* S_d, void* stream) {
    DCV_REQUIRE(Q_d && Psorted_d && start_d && S_d, "dcv_cluster_dist_sums: null argument");
    DCV_REQUIRE(nq >= 1 && d >= 1 && d <= kScMaxD && k >= 1 && k <= kScMaxK, "dcv_cluster_dist_sums: nq=%lld d=%d k=%d out of range",
                (long long)nq, d, k);
    hipStream_t s = as_stream(stream);
    const dim3 grid((unsigned)cdiv(nq, kScThreads), (unsigned)k);
    switch (d) {
        case 1: hipLaunchKernelGGL(cluster_dist_sums_kernel<1>, grid, dim3(kScThreads), 0, s, Q_d, nq, Psorted_d, start_d, k, d, S_d); break;
        case 2: hipLaunchKernelGGL(cluster_dist_sums_kernel<2>, grid, dim3(kScThreads), 0, s, Q_d, nq, Psorted_d, start_d, k, d, S_d); break;
        case 3: hipLaunchKernelGGL(cluster_dist_sums_kernel<3>, grid, dim3(kScThreads), 0, s, Q_d, nq, Psorted_d, start_d, k, d, S_d); break;
        case 4: hipLaunchKernelGGL(cluster_dist_sums_kernel<4>, grid, dim3(kScThreads), 0, s, Q_d, nq, Psorted_d, start_d, k, d, S_d); break;
        default: hipLaunchKernelGGL(cluster_dist_sums_kernel<0>, grid, dim3(kScThreads), 0, s, Q_d, nq, Psorted_d, start_d, k, d, S_d); break;
    }
    DCV_CHECK_LAUNCH();
    return DCV_OK;
}

extern "C" size_t dcv_silhouette_sum_workspace(int64_t nq) { return (size_t)sc_blocks(nq) * sizeof(double); }

extern "C" int dcv_silhouette_sum(const double* S_d, int64_t nq, int32_t k, const int32_t* qlabels_d, const int64_t* start_d,
                                  double* sum_d, void* ws_d, size_t ws_bytes, void* stream) {
    DCV_REQUIRE(S_d && qlabels_d && start_d && sum_d && ws_d, "dcv_silhouette_sum: null argument");
    const int nb = sc_blocks(nq);
    DCV_REQUIRE(ws_bytes >= (size_t)nb * sizeof(double), "dcv_silhouette_sum: workspace too small");
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(silhouette_sum_kernel, dim3(nb), dim3(kScThreads), 0, s, S_d, nq, k, qlabels_d, start_d, static_cast<double*>(ws_d));
    DCV_CHECK_LAUNCH();
    hipLaunchKernelGGL(sum_blocks_kernel, dim3(1), dim3(256), 0, s, static_cast<const double*>(ws_d), nb, 1, sum_d);
    DCV_CHECK_LAUNCH();
    return DCV_OK;
}

extern "C" size_t dcv_linear_binning_workspace(int32_t d, int32_t bins) {
    if (d < 1 || d > 2 || bins < 2) return 0;
    const size_t cells = d == 1 ? (size_t)bins : (size_t)bins * bins;
    return (cells + 1) * sizeof(unsigned long long);
}

extern "C" int dcv_linear_binning(const double* P_d, int64_t n, int64_t ldp, int32_t d, const int32_t* cols_h, const double* lo_h,
                                  const double* hi_h, int32_t bins, double* grid_d, int64_t* outside_h, void* ws_d, size_t ws_bytes,
                                  void* stream) {
    DCV_REQUIRE(P_d && cols_h && lo_h && hi_h && grid_d && n > 0, "dcv_linear_binning: bad arguments");
    DCV_REQUIRE((d == 1 || d == 2) && bins >= 2 && bins <= 8192, "dcv_linear_binning: d=%d (1 or 2) bins=%d (2..8192) unsupported", d, bins);
    DCV_REQUIRE(ws_d && ws_bytes >= dcv_linear_binning_workspace(d, bins), "dcv_linear_binning: workspace too small");
    for (int c = 0; c < d; ++c) DCV_REQUIRE(hi_h[c] > lo_h[c] && cols_h[c] >= 0 && cols_h[c] < ldp, "dcv_linear_binning: bad bounds / column");
    hipStream_t s = as_stream(stream);
    const int64_t cells = d == 1 ? bins : (int64_t)bins * bins;
    unsigned long long* fixed = static_cast<unsigned long long*>(ws_d);
    DCV_CHECK_HIP(hipMemsetAsync(fixed, 0, (cells + 1) * sizeof(unsigned long long), s));
    int64_t nb = cdiv(n, (int64_t)kScThreads * 8);
    if (nb > (int64_t)num_cus() * 8) nb = (int64_t)num_cus() * 8;
    if (nb < 1) nb = 1;
    const double inv0 = (double)(bins - 1) / (hi_h[0] - lo_h[0]);
    if (d == 1) {
        const size_t lds = bins <= 4096 ? (size_t)bins * sizeof(unsigned long long) : 0;
        hipLaunchKernelGGL(linear_binning_kernel<1>, dim3((unsigned)nb), dim3(kScThreads), lds, s, P_d, n, ldp, (int)cols_h[0], 0, lo_h[0], inv0, 0.0,
                           0.0, (int)bins, fixed, fixed + cells);
    } else {
        const double inv1 = (double)(bins - 1) / (hi_h[1] - lo_h[1]);
        hipLaunchKernelGGL(linear_binning_kernel<2>, dim3((unsigned)nb), dim3(kScThreads), 0, s, P_d, n, ldp, (int)cols_h[0], (int)cols_h[1], lo_h[0],
                           inv0, lo_h[1], inv1, (int)bins, fixed, fixed + cells);
    }
    DCV_CHECK_LAUNCH();
    hipLaunchKernelGGL(binning_to_double_kernel, dim3((unsigned)cdiv(cells, 256)), dim3(256), 0, s, fixed, cells, grid_d);
    DCV_CHECK_LAUNCH();
    if (outside_h) {
        unsigned long long miss = 0;
        DCV_CHECK_HIP(hipMemcpyAsync(&miss, fixed + cells, sizeof(miss), hipMemcpyDeviceToHost, s));
        DCV_CHECK_HIP(hipStreamSynchronize(s));
        *outside_h = (int64_t)miss;
    }
    return DCV_OK;
}
