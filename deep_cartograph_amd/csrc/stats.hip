// Column statistics and normalisation: HBM-bound streaming passes over the frames x features
// matrix.  Lanes run along the feature axis (16-byte loads, whole rows per wave), rows are
// strided over the block; partial results per block are combined in a fixed order, so the
// result is deterministic.
#include "common.h"
#include <stdlib.h>

namespace dcv {

// 16-byte non-temporal load: the feature matrix is streamed once per pass (round 4: the same policy took the k-means point
// stream from 4.6 to 5.7 TB/s, the normalisation +2.7 %); DCV_STREAM_NT=0 at run time restores plain loads where a kernel
// offers the switch
__device__ __forceinline__ float4 nt_load4(const float* p) {
    typedef float nv4 __attribute__((ext_vector_type(4)));
    const nv4 v = __builtin_nontemporal_load(reinterpret_cast<const nv4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}


constexpr int kStatsThreads = 256;
constexpr int kStatsMaxBlocks = 2048;

struct StatsGeom {
    int cpb;      // column groups (of VEC columns) handled per pass by one block
    int rpp;      // rows per pass
    int ncb;      // column-block passes
};

template <int VEC>
__host__ __device__ inline StatsGeom stats_geom(int F) {
    int ng = (F + VEC - 1) / VEC;
    StatsGeom g;
    g.cpb = ng < kStatsThreads ? ng : kStatsThreads;
    g.rpp = kStatsThreads / g.cpb;
    g.ncb = (ng + g.cpb - 1) / g.cpb;
    return g;
}

// partial layout: part[block][stat][F], stat = 0 sum, 1 sumsq, 2 min, 3 max (float64)
template <int VEC>
__global__ __launch_bounds__(kStatsThreads) void col_stats_kernel(const float* __restrict__ X, int64_t n, int F,
                                                                  int64_t ld, double* __restrict__ part) {
    const StatsGeom g = stats_geom<VEC>(F);
    const int t = threadIdx.x;
    const int cg = t % g.cpb;
    const int rl = t / g.cpb;
    const bool active = rl < g.rpp;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* s_sum = reinterpret_cast<double*>(smem);               // [256][VEC]
    double* s_sq = s_sum + kStatsThreads * VEC;                     // [256][VEC]
    float* s_mn = reinterpret_cast<float*>(s_sq + kStatsThreads * VEC);
    float* s_mx = s_mn + kStatsThreads * VEC;

    // contiguous block of rows per workgroup
    const int64_t rows_per_block = (n + gridDim.x - 1) / gridDim.x;
    const int64_t r_begin = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r_end = r_begin + rows_per_block < n ? r_begin + rows_per_block : n;
    double* my_part = part + (int64_t)blockIdx.x * 4 * F;

    for (int cb = 0; cb < g.ncb; ++cb) {
        const int col = (cb * g.cpb + cg) * VEC;
        const bool col_ok = active && col < F;
        double sum[VEC], sq[VEC];
        float mn[VEC], mx[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            sum[v] = 0.0;
            sq[v] = 0.0;
            mn[v] = INFINITY;
            mx[v] = -INFINITY;
        }
        if (col_ok) {
            const float* base = X + col;
            int64_t r = r_begin + rl;
            // 4 independent row loads in flight
            for (; r + 3 * (int64_t)g.rpp < r_end; r += 4 * (int64_t)g.rpp) {
                float x[4][VEC];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float* p = base + (r + (int64_t)u * g.rpp) * ld;
                    if constexpr (VEC == 4) {
                        const float4 q = nt_load4(p);
                        x[u][0] = q.x; x[u][1] = q.y; x[u][2] = q.z; x[u][3] = q.w;
                    } else {
                        x[u][0] = *p;
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        const double d = (double)x[u][v];
                        sum[v] += d;
                        sq[v] = fma(d, d, sq[v]);
                        mn[v] = fminf(mn[v], x[u][v]);
                        mx[v] = fmaxf(mx[v], x[u][v]);
                    }
            }
            for (; r < r_end; r += g.rpp) {
                const float* p = base + r * ld;
                float x[VEC];
                if constexpr (VEC == 4) {
                    const float4 q = nt_load4(p);
                    x[0] = q.x; x[1] = q.y; x[2] = q.z; x[3] = q.w;
                } else {
                    x[0] = *p;
                }
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    const double d = (double)x[v];
                    sum[v] += d;
                    sq[v] = fma(d, d, sq[v]);
                    mn[v] = fminf(mn[v], x[v]);
                    mx[v] = fmaxf(mx[v], x[v]);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            s_sum[t * VEC + v] = sum[v];
            s_sq[t * VEC + v] = sq[v];
            s_mn[t * VEC + v] = mn[v];
            s_mx[t * VEC + v] = mx[v];
        }
        __syncthreads();
        if (rl == 0 && col < F) {
            // fixed-order combine over the row lanes of this block
            for (int v = 0; v < VEC && col + v < F; ++v) {
                double a = 0.0, b = 0.0;
                float lo = INFINITY, hi = -INFINITY;
                for (int q = 0; q < g.rpp; ++q) {
                    const int tt = q * g.cpb + cg;
                    a += s_sum[tt * VEC + v];
                    b += s_sq[tt * VEC + v];
                    lo = fminf(lo, s_mn[tt * VEC + v]);
                    hi = fmaxf(hi, s_mx[tt * VEC + v]);
                }
                my_part[0 * F + col + v] = a;
                my_part[1 * F + col + v] = b;
                my_part[2 * F + col + v] = (double)lo;
                my_part[3 * F + col + v] = (double)hi;
            }
        }
    }
}

// out[stat][f] = combination over the blocks' partials, fixed order.  A block covers 64 consecutive outputs; its waves
// take the partials b = wave, wave + 4, ... with eight loads in flight each (a single thread walking all 2048 partials
// of an output one dependent load at a time took 855 us at F = 512 -- longer than the statistics pass itself) and are
// combined in wave order.
__global__ __launch_bounds__(256) void col_stats_final_kernel(const double* __restrict__ part, int nblocks, int F, double* __restrict__ out) {
    __shared__ double s_red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;  // index into [4][F]
    const bool ok = i < 4 * F;
    const int stat = ok ? i / F : 0;
    double acc = stat == 2 ? INFINITY : (stat == 3 ? -INFINITY : 0.0);
    if (ok) {
        const double* p = part + i;
        const int64_t stride = (int64_t)4 * F;
        if (stat < 2) {
#pragma unroll 8
            for (int b = wave; b < nblocks; b += 4) acc += p[(int64_t)b * stride];
        } else if (stat == 2) {
#pragma unroll 8
            for (int b = wave; b < nblocks; b += 4) acc = fmin(acc, p[(int64_t)b * stride]);
        } else {
#pragma unroll 8
            for (int b = wave; b < nblocks; b += 4) acc = fmax(acc, p[(int64_t)b * stride]);
        }
    }
    s_red[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && ok) {
        double r;
        if (stat < 2) r = ((s_red[0][lane] + s_red[1][lane]) + s_red[2][lane]) + s_red[3][lane];
        else if (stat == 2) r = fmin(fmin(s_red[0][lane], s_red[1][lane]), fmin(s_red[2][lane], s_red[3][lane]));
        else r = fmax(fmax(s_red[0][lane], s_red[1][lane]), fmax(s_red[2][lane], s_red[3][lane]));
        out[i] = r;
    }
}

static int stats_blocks(int64_t n, int F) {
    // enough rows per block to amortise the combine; at most kStatsMaxBlocks
    int64_t b = cdiv(n, 64);
    if (b > kStatsMaxBlocks) b = kStatsMaxBlocks;
    if (b < 1) b = 1;
    return (int)b;
}

// ------------------------------------------------------------------------------ normalise
template <int VEC>
__global__ __launch_bounds__(256) void normalize_kernel(const float* __restrict__ X, float* __restrict__ Y, int64_t n,
                                                        int F, int64_t ldx, int64_t ldy,
                                                        const float* __restrict__ mean,
                                                        const float* __restrict__ range) {
    const int ng = (F + VEC - 1) / VEC;
    const int64_t total = n * ng;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / ng;
        const int c = (int)(i - r * ng) * VEC;
        if constexpr (VEC == 4) {
            const float4 x = *reinterpret_cast<const float4*>(X + r * ldx + c);
            const float4 m = *reinterpret_cast<const float4*>(mean + c);
            const float4 s = *reinterpret_cast<const float4*>(range + c);
            float4 y;
            // true IEEE subtraction and division, as torch's sub_ / div_ (cv_calculator.py:833-835)
            y.x = __fdiv_rn(__fsub_rn(x.x, m.x), s.x);
            y.y = __fdiv_rn(__fsub_rn(x.y, m.y), s.y);
            y.z = __fdiv_rn(__fsub_rn(x.z, m.z), s.z);
            y.w = __fdiv_rn(__fsub_rn(x.w, m.w), s.w);
            *reinterpret_cast<float4*>(Y + r * ldy + c) = y;
        } else {
            Y[r * ldy + c] = __fdiv_rn(__fsub_rn(X[r * ldx + c], mean[c]), range[c]);
        }
    }
}

// The same for rows of F = 4 * ng floats with 256 % ng == 0: a thread keeps ONE column group -- mean and range are read
// once -- and walks the block's rows with U independent 16-byte loads in flight (the flat grid-stride form above has one
// load in flight per thread, a 64-bit division per element and re-reads mean / range every iteration: 4.5 TB/s at 5M x 256).
// NT (DCV_NORMALIZE_NT=1, experiment): the matrix is streamed once -- non-temporal loads / stores keep it out of the L2's way
template <int U, bool NT = false>
__global__ __launch_bounds__(256) void normalize_rows_kernel(const float* __restrict__ X, float* __restrict__ Y, int64_t n, int ng,
                                                             int64_t ldx, int64_t ldy, const float* __restrict__ mean,
                                                             const float* __restrict__ range, int rows_per_block) {
    const int t = threadIdx.x, tx = t % ng, ty = t / ng, lanes = 256 / ng;
    const float4 m = *reinterpret_cast<const float4*>(mean + 4 * tx);
    const float4 s = *reinterpret_cast<const float4*>(range + 4 * tx);
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < n ? r0 + rows_per_block : n;
    // Two alternating batches of U rows per thread: the loads of the next batch are issued before this one is divided and
    // stored (a batch at a time left every wave with nothing in flight during its 50 division instructions per float4), and
    // they are UNCONDITIONAL loads from clamped rows -- a load inside a branch makes hipcc put s_waitcnt vmcnt(0) behind
    // it (round 4, kmeans.hip).  In place (Y == X) is safe: a row is read once, before it is written, by the same thread.
    auto ld = [&](float4 (&x)[U], int64_t r) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int64_t rr = r + (int64_t)u * lanes;
            rr = rr < r1 ? rr : r1 - 1;
            const float* p = X + rr * ldx + 4 * tx;
            if constexpr (NT) x[u] = nt_load4(p);
            else x[u] = *reinterpret_cast<const float4*>(p);
        }
    };
    auto st = [&](const float4 (&x)[U], int64_t r) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t rr = r + (int64_t)u * lanes;
            if (rr < r1) {
                float4 y;   // true IEEE subtraction and division, as torch's sub_ / div_ (cv_calculator.py:833-835)
                y.x = __fdiv_rn(__fsub_rn(x[u].x, m.x), s.x);
                y.y = __fdiv_rn(__fsub_rn(x[u].y, m.y), s.y);
                y.z = __fdiv_rn(__fsub_rn(x[u].z, m.z), s.z);
                y.w = __fdiv_rn(__fsub_rn(x[u].w, m.w), s.w);
                if constexpr (NT) {
                    typedef float nv4 __attribute__((ext_vector_type(4)));
                    __builtin_nontemporal_store(nv4{y.x, y.y, y.z, y.w}, reinterpret_cast<nv4*>(Y + rr * ldy + 4 * tx));
                } else {
                    *reinterpret_cast<float4*>(Y + rr * ldy + 4 * tx) = y;
                }
            }
        }
    };
    if (r0 >= r1) return;
    const int64_t step = (int64_t)lanes * U;
    float4 xa[U], xb[U];
    int64_t r = r0 + ty;
    ld(xa, r);
    for (; r < r1; r += 2 * step) {
        ld(xb, r + step);
        st(xa, r);
        ld(xa, r + 2 * step);
        st(xb, r + step);
    }
}

}  // namespace dcv

using namespace dcv;

static bool vec4_ok(const void* p, int F, int64_t ld) {
    return (F % 4 == 0) && (ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(p) & 15) == 0);
}

extern "C" size_t dcv_col_stats_workspace(int64_t n, int32_t F) {
    if (n <= 0 || F <= 0) return 0;
    return (size_t)stats_blocks(n, F) * 4 * (size_t)F * sizeof(double);
}

extern "C" int dcv_col_stats(const float* X_d, int64_t n, int32_t F, int64_t ld, double* out_d, void* ws_d,
                             size_t ws_bytes, void* stream) {
    DCV_REQUIRE(X_d && out_d && n > 0 && F > 0 && ld >= F, "dcv_col_stats: bad arguments (n=%lld F=%d ld=%lld)",
                (long long)n, F, (long long)ld);
    DCV_REQUIRE(ws_d && ws_bytes >= dcv_col_stats_workspace(n, F), "dcv_col_stats: workspace too small");
    hipStream_t s = as_stream(stream);
    const int nb = stats_blocks(n, F);
    double* part = static_cast<double*>(ws_d);
    if (vec4_ok(X_d, F, ld)) {
        const size_t lds = kStatsThreads * 4 * (2 * sizeof(double) + 2 * sizeof(float));
        hipLaunchKernelGGL(col_stats_kernel<4>, dim3(nb), dim3(kStatsThreads), lds, s, X_d, n, F, ld, part);
    } else {
        const size_t lds = kStatsThreads * 1 * (2 * sizeof(double) + 2 * sizeof(float));
        hipLaunchKernelGGL(col_stats_kernel<1>, dim3(nb), dim3(kStatsThreads), lds, s, X_d, n, F, ld, part);
    }
    DCV_CHECK_LAUNCH();
    hipLaunchKernelGGL(col_stats_final_kernel, dim3((4 * F + 63) / 64), dim3(256), 0, s, part, nb, F, out_d);
    DCV_CHECK_LAUNCH();
    return DCV_OK;
}

extern "C" int dcv_normalize(const float* X_d, float* Y_d, int64_t n, int32_t F, int64_t ldx, int64_t ldy,
                             const float* mean_d, const float* range_d, void* stream) {
    DCV_REQUIRE(X_d && Y_d && mean_d && range_d && n > 0 && F > 0 && ldx >= F && ldy >= F,
                "dcv_normalize: bad arguments");
    hipStream_t s = as_stream(stream);
    const bool v4 = vec4_ok(X_d, F, ldx) && vec4_ok(Y_d, F, ldy) && ((reinterpret_cast<uintptr_t>(mean_d) & 15) == 0) &&
                    ((reinterpret_cast<uintptr_t>(range_d) & 15) == 0);
    const int64_t total = n * (v4 ? F / 4 : F);
    int64_t blocks = cdiv(total, 256);
    const int64_t cap = (int64_t)num_cus() * 16;
    if (blocks > cap) blocks = cap;
    const int ng = F / 4;
    static const bool rows_off = [] { const char* e = getenv("DCV_NORMALIZE_FLAT"); return e && e[0] == '1'; }();
    if (v4 && !rows_off && ng <= 256 && 256 % ng == 0 && (reinterpret_cast<uintptr_t>(mean_d) & 15) == 0 && (reinterpret_cast<uintptr_t>(range_d) & 15) == 0) {
        const int lanes = 256 / ng;
        int rpb = lanes * 8 * 4;   // four rounds of eight loads per thread
        while ((int64_t)cdiv(n, rpb) > (int64_t)num_cus() * 64) rpb *= 2;
        // non-temporal loads / stores: the matrix is streamed exactly once.  A/B on one box, 5M x 256, three runs each (round 4):
        // 5.11 / 5.12 / 5.14 TB/s plain, 5.31 / 5.21 / 5.25 TB/s non-temporal (+2.7 %); DCV_NORMALIZE_NT=0 restores plain accesses
        static const int nt_env = [] { const char* e = getenv("DCV_NORMALIZE_NT"); return e ? atoi(e) : 1; }();
        static const int rpb_env = [] { const char* e = getenv("DCV_NORMALIZE_RPB"); return e ? atoi(e) : 0; }();
        if (rpb_env > 0) rpb = rpb_env;
        if (nt_env) hipLaunchKernelGGL((normalize_rows_kernel<8, true>), dim3((unsigned)cdiv(n, rpb)), dim3(256), 0, s, X_d, Y_d, n, ng, ldx, ldy, mean_d, range_d, rpb);
        else hipLaunchKernelGGL((normalize_rows_kernel<8, false>), dim3((unsigned)cdiv(n, rpb)), dim3(256), 0, s, X_d, Y_d, n, ng, ldx, ldy, mean_d, range_d, rpb);
    } else if (v4)
        hipLaunchKernelGGL(normalize_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, s, X_d, Y_d, n, F, ldx, ldy, mean_d,
                           range_d);
    else
        hipLaunchKernelGGL(normalize_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, s, X_d, Y_d, n, F, ldx, ldy, mean_d,
                           range_d);
    DCV_CHECK_LAUNCH();
    return DCV_OK;
}
