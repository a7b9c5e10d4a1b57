// k-means on the projected CVs (float64, d <= 16, k <= 64): one HBM-bound pass per Lloyd
// iteration computing labels, per-cluster sums / counts, inertia and the number of changed
// labels; centroid-nearest-sample search; 1-NN label transfer.
//
// Determinism: every wave owns a private LDS accumulator, waves are combined in wave order,
// blocks in block order, and each thread walks its points in increasing index order.
#include "common.h"

namespace dcv {

constexpr int kKmThreads = 256;
constexpr int kKmMaxD = 16;
constexpr int kKmMaxK = 64;
constexpr int kKmMaxBlocks = 1024;

static int km_blocks(int64_t n) {
    int64_t b = cdiv(n, (int64_t)kKmThreads * 4);
    const int64_t cap = (int64_t)num_cus() * 4;
    if (b > cap) b = cap;
    if (b > kKmMaxBlocks) b = kKmMaxBlocks;
    if (b < 1) b = 1;
    return (int)b;
}

__device__ __forceinline__ void load_point(const double* __restrict__ P, int64_t i, int d, double* x) {
    const double* p = P + i * d;
    if ((d & 1) == 0) {
#pragma unroll
        for (int c = 0; c < kKmMaxD; c += 2) {
            if (c < d) {
                const double2 v = *reinterpret_cast<const double2*>(p + c);
                x[c] = v.x;
                x[c + 1] = v.y;
            }
        }
    } else {
#pragma unroll
        for (int c = 0; c < kKmMaxD; ++c)
            if (c < d) x[c] = p[c];
    }
}

// acc layout per block: [sums k*d | counts k | inertia | changed]
__global__ __launch_bounds__(kKmThreads) void kmeans_step_kernel(const double* __restrict__ P, int64_t n, int d,
                                                                 const double* __restrict__ offset,
                                                                 const double* __restrict__ centers, int k,
                                                                 int32_t* __restrict__ labels,
                                                                 double* __restrict__ mindist,
                                                                 double* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int W = k * d + k;                       // per-wave accumulator width
    double* s_c = reinterpret_cast<double*>(smem);  // [k][d]
    double* s_cn = s_c + k * d;                     // [k]
    double* s_acc = s_cn + k;                       // [4 waves][W]
    double* s_misc = s_acc + 4 * W;                 // [256] inertia, then [256] changed
    double off[kKmMaxD];
#pragma unroll
    for (int c = 0; c < kKmMaxD; ++c) off[c] = (offset && c < d) ? offset[c] : 0.0;
    const int t = threadIdx.x;
    const int wave = t >> 6;
    for (int i = t; i < k * d; i += kKmThreads) s_c[i] = centers[i];
    for (int i = t; i < 4 * W; i += kKmThreads) s_acc[i] = 0.0;
    __syncthreads();
    for (int j = t; j < k; j += kKmThreads) {
        double s = 0.0;
        for (int c = 0; c < d; ++c) s += s_c[j * d + c] * s_c[j * d + c];
        s_cn[j] = s;
    }
    __syncthreads();

    double* my_acc = s_acc + wave * W;
    double inertia = 0.0;
    double changed = 0.0;
    // contiguous block of points per workgroup, thread-strided inside
    const int64_t per_block = (n + gridDim.x - 1) / gridDim.x;
    const int64_t begin = (int64_t)blockIdx.x * per_block;
    const int64_t end = begin + per_block < n ? begin + per_block : n;
    for (int64_t i = begin + t; i < end; i += kKmThreads) {
        double x[kKmMaxD];
        load_point(P, i, d, x);
        if (offset) {
#pragma unroll
            for (int c = 0; c < kKmMaxD; ++c)
                if (c < d) x[c] -= off[c];  // X -= X_mean, as KMeans.fit does
        }
        // label = argmin_j (|c_j|^2 - 2 x.c_j), first minimum wins (sklearn lloyd_iter_chunked_dense)
        int best = 0;
        double bestv = INFINITY;
        for (int j = 0; j < k; ++j) {
            double dot = 0.0;
#pragma unroll
            for (int c = 0; c < kKmMaxD; ++c)
                if (c < d) dot += x[c] * s_c[j * d + c];
            const double v = s_cn[j] - 2.0 * dot;
            if (v < bestv) {
                bestv = v;
                best = j;
            }
        }
        double dist = 0.0;
#pragma unroll
        for (int c = 0; c < kKmMaxD; ++c)
            if (c < d) {
                const double df = x[c] - s_c[best * d + c];
                dist += df * df;
            }
        inertia += dist;
        if (mindist) mindist[i] = dist;
        if (labels[i] != best) changed += 1.0;
        labels[i] = best;
#pragma unroll
        for (int c = 0; c < kKmMaxD; ++c)
            if (c < d) atomicAdd(&my_acc[best * d + c], x[c]);
        atomicAdd(&my_acc[k * d + best], 1.0);
    }
    s_misc[t] = inertia;
    s_misc[kKmThreads + t] = changed;
    __syncthreads();
    double* my_part = part + (int64_t)blockIdx.x * (W + 2);
    for (int i = t; i < W; i += kKmThreads) my_part[i] = ((s_acc[i] + s_acc[W + i]) + s_acc[2 * W + i]) + s_acc[3 * W + i];
    if (t < 2) {
        double s = 0.0;
        for (int q = 0; q < kKmThreads; ++q) s += s_misc[t * kKmThreads + q];
        my_part[W + t] = s;
    }
}

__global__ void kmeans_final_kernel(const double* __restrict__ part, int nblocks, int width, double* __restrict__ acc) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= width) return;
    double s = 0.0;
    for (int b = 0; b < nblocks; ++b) s += part[(int64_t)b * width + i];
    acc[i] = s;
}

// ------------------------------------------------------------------ nearest sample per centroid
// numpy: sqrt(add.reduce((x - c)**2, axis=1)); pairwise summation degenerates to a sequential
// sum for d < 8 and to 8 interleaved partial sums for 8 <= d <= 128.
__device__ __forceinline__ double np_norm(const double* x, const double* c, int d) {
    // separately rounded multiply / add (no FMA contraction), as NumPy's ufunc loops do
    double sq[kKmMaxD];
#pragma unroll
    for (int q = 0; q < kKmMaxD; ++q)
        if (q < d) {
            const double df = x[q] - c[q];
            sq[q] = __dmul_rn(df, df);
        }
    double res;
    if (d < 8) {
        res = 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (q < d) res = __dadd_rn(res, sq[q]);
    } else {
        double r[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) r[q] = sq[q];
        const int full = d - (d % 8);
#pragma unroll
        for (int q = 8; q < kKmMaxD; q += 8)
            if (q < full) {
#pragma unroll
                for (int u = 0; u < 8; ++u) r[u] = __dadd_rn(r[u], sq[q + u]);
            }
        res = __dadd_rn(__dadd_rn(__dadd_rn(r[0], r[1]), __dadd_rn(r[2], r[3])),
                        __dadd_rn(__dadd_rn(r[4], r[5]), __dadd_rn(r[6], r[7])));
#pragma unroll
        for (int q = 8; q < kKmMaxD; ++q)
            if (q >= full && q < d) res = __dadd_rn(res, sq[q]);
    }
    return sqrt(res);
}

// grid.x = point blocks, grid.y = centroid.  part[(j*nblocks + b)] = (dist, row)
__global__ __launch_bounds__(kKmThreads) void nearest_rows_kernel(const double* __restrict__ P, int64_t n, int d,
                                                                  const double* __restrict__ centers,
                                                                  double* __restrict__ pdist, int64_t* __restrict__ prow) {
    __shared__ double s_d[kKmThreads];
    __shared__ int64_t s_i[kKmThreads];
    const int j = blockIdx.y;
    const int t = threadIdx.x;
    double c[kKmMaxD];
#pragma unroll
    for (int q = 0; q < kKmMaxD; ++q) c[q] = q < d ? centers[j * d + q] : 0.0;
    const int64_t per_block = (n + gridDim.x - 1) / gridDim.x;
    const int64_t begin = (int64_t)blockIdx.x * per_block;
    const int64_t end = begin + per_block < n ? begin + per_block : n;
    double best = INFINITY;
    int64_t besti = INT64_MAX;
    for (int64_t i = begin + t; i < end; i += kKmThreads) {
        double x[kKmMaxD];
        load_point(P, i, d, x);
        const double v = np_norm(x, c, d);
        if (v < best) {  // increasing i: strict '<' keeps the first index
            best = v;
            besti = i;
        }
    }
    s_d[t] = best;
    s_i[t] = besti;
    __syncthreads();
    for (int off = kKmThreads / 2; off > 0; off >>= 1) {
        if (t < off) {
            const double od = s_d[t + off];
            const int64_t oi = s_i[t + off];
            if (od < s_d[t] || (od == s_d[t] && oi < s_i[t])) {
                s_d[t] = od;
                s_i[t] = oi;
            }
        }
        __syncthreads();
    }
    if (t == 0) {
        pdist[(int64_t)j * gridDim.x + blockIdx.x] = s_d[0];
        prow[(int64_t)j * gridDim.x + blockIdx.x] = s_i[0];
    }
}

__global__ void nearest_rows_final(const double* __restrict__ pdist, const int64_t* __restrict__ prow, int nblocks, int k,
                                   int64_t row_offset, double* __restrict__ dist, int64_t* __restrict__ rows) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= k) return;
    double best = INFINITY;
    int64_t besti = INT64_MAX;
    for (int b = 0; b < nblocks; ++b) {
        const double v = pdist[(int64_t)j * nblocks + b];
        const int64_t i = prow[(int64_t)j * nblocks + b];
        if (v < best || (v == best && i < besti)) {
            best = v;
            besti = i;
        }
    }
    dist[j] = best;
    rows[j] = besti == INT64_MAX ? -1 : besti + row_offset;
}

// ------------------------------------------------------------------ 1-NN of supplementary points
__global__ __launch_bounds__(kKmThreads) void nearest_point_kernel(const double* __restrict__ train, int64_t n_train,
                                                                   const double* __restrict__ sup, int64_t n_sup, int d,
                                                                   int64_t* __restrict__ nn) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* tile = reinterpret_cast<double*>(smem);  // [256][d]
    const int t = threadIdx.x;
    const int64_t i = (int64_t)blockIdx.x * kKmThreads + t;
    double x[kKmMaxD];
    if (i < n_sup) load_point(sup, i, d, x);
    double best = INFINITY;
    int64_t besti = -1;
    for (int64_t base = 0; base < n_train; base += kKmThreads) {
        const int64_t cnt = n_train - base < kKmThreads ? n_train - base : kKmThreads;
        __syncthreads();
        for (int64_t q = t; q < cnt * d; q += kKmThreads) tile[q] = train[base * d + q];
        __syncthreads();
        if (i < n_sup) {
            for (int q = 0; q < (int)cnt; ++q) {
                double s = 0.0;
#pragma unroll
                for (int c = 0; c < kKmMaxD; ++c)
                    if (c < d) {
                        const double df = x[c] - tile[q * d + c];
                        s += df * df;
                    }
                if (s < best) {
                    best = s;
                    besti = base + q;
                }
            }
        }
    }
    if (i < n_sup) nn[i] = besti;
}

}  // namespace dcv

using namespace dcv;

extern "C" size_t dcv_kmeans_workspace(int64_t n, int32_t d, int32_t k) {
    if (n <= 0 || d <= 0 || k <= 0) return 0;
    return (size_t)kKmMaxBlocks * ((size_t)k * d + k + 2) * sizeof(double);
}

extern "C" int dcv_kmeans_step(const double* P_d, int64_t n, int32_t d, const double* offset_d,
                               const double* centers_d, int32_t k, int32_t* labels_d, double* acc_d, double* mindist_d, void* ws_d, size_t ws_bytes,
                               void* stream) {
    DCV_REQUIRE(P_d && centers_d && labels_d && acc_d && n > 0, "dcv_kmeans_step: bad arguments");
    DCV_REQUIRE(d >= 1 && d <= kKmMaxD && k >= 1 && k <= kKmMaxK, "dcv_kmeans_step: d=%d (1..16) k=%d (1..64) unsupported", d, k);
    DCV_REQUIRE(ws_d && ws_bytes >= dcv_kmeans_workspace(n, d, k), "dcv_kmeans_step: workspace too small");
    hipStream_t s = as_stream(stream);
    const int nb = km_blocks(n);
    const int W = k * d + k;
    const size_t lds = ((size_t)k * d + k + 4 * W + 2 * kKmThreads) * sizeof(double);
    double* part = static_cast<double*>(ws_d);
    hipLaunchKernelGGL(kmeans_step_kernel, dim3(nb), dim3(kKmThreads), lds, s, P_d, n, d, offset_d, centers_d, k, labels_d, mindist_d, part);
    DCV_CHECK_LAUNCH();
    hipLaunchKernelGGL(kmeans_final_kernel, dim3((W + 2 + 255) / 256), dim3(256), 0, s, part, nb, W + 2, acc_d);
    DCV_CHECK_LAUNCH();
    return DCV_OK;
}

extern "C" size_t dcv_nearest_rows_workspace(int64_t n, int32_t d, int32_t k) {
    (void)d;
    if (n <= 0 || k <= 0) return 0;
    return (size_t)kKmMaxBlocks * k * (sizeof(double) + sizeof(int64_t));
}

extern "C" int dcv_nearest_rows(const double* P_d, int64_t n, int32_t d, const double* centers_d, int32_t k,
                                int64_t row_offset, double* dist_d, int64_t* rows_d, void* ws_d, size_t ws_bytes,
                                void* stream) {
    DCV_REQUIRE(P_d && centers_d && dist_d && rows_d && n > 0, "dcv_nearest_rows: bad arguments");
    DCV_REQUIRE(d >= 1 && d <= kKmMaxD && k >= 1 && k <= 1024, "dcv_nearest_rows: d=%d k=%d unsupported", d, k);
    DCV_REQUIRE(ws_d && ws_bytes >= dcv_nearest_rows_workspace(n, d, k), "dcv_nearest_rows: workspace too small");
    hipStream_t s = as_stream(stream);
    const int nb = km_blocks(n);
    double* pdist = static_cast<double*>(ws_d);
    int64_t* prow = reinterpret_cast<int64_t*>(pdist + (size_t)kKmMaxBlocks * k);
    hipLaunchKernelGGL(nearest_rows_kernel, dim3(nb, k), dim3(kKmThreads), 0, s, P_d, n, d, centers_d, pdist, prow);
    DCV_CHECK_LAUNCH();
    hipLaunchKernelGGL(nearest_rows_final, dim3((k + 63) / 64), dim3(64), 0, s, pdist, prow, nb, k, row_offset, dist_d, rows_d);
    DCV_CHECK_LAUNCH();
    return DCV_OK;
}

extern "C" int dcv_nearest_point(const double* train_d, int64_t n_train, const double* sup_d, int64_t n_sup, int32_t d,
                                 int64_t* nn_d, void* stream) {
    DCV_REQUIRE(train_d && sup_d && nn_d && n_train > 0 && n_sup > 0, "dcv_nearest_point: bad arguments");
    DCV_REQUIRE(d >= 1 && d <= kKmMaxD, "dcv_nearest_point: d=%d unsupported", d);
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(nearest_point_kernel, dim3((unsigned)cdiv(n_sup, kKmThreads)), dim3(kKmThreads),
                       (size_t)kKmThreads * d * sizeof(double), s, train_d, n_train, sup_d, n_sup, d, nn_d);
    DCV_CHECK_LAUNCH();
    return DCV_OK;
}
