// Linear projection of the frames x features matrix onto d <= 16 collective variables:
//   out = (((x - fmean)/frange) @ W + bias - cvmean) / cvrange          (one HBM-bound pass)
// 16 lanes share a row (16-byte loads, 256 contiguous bytes per row segment), each lane keeps
// 4 rows in flight so that the LDS-resident weights are read once per 4 rows; the 16 partial
// dot products are combined with wave shuffles.  Per-column extrema of the output are reduced
// per block and combined in a fixed order.
#include "common.h"

namespace dcv {

constexpr int kProjThreads = 256;
constexpr int kLanesPerRow = 16;
constexpr int kRowsInFlight = 4;
constexpr int kRowsPerPass = kProjThreads / kLanesPerRow;           // 16
constexpr int kRowsPerIter = kRowsPerPass * kRowsInFlight;          // 64
constexpr int kProjMaxBlocks = 4096;

struct ProjArgs {
    const float* X;
    int64_t n;
    int F;
    int64_t ld;
    const float* fmean;   // may be null
    const float* frange;  // may be null
    const float* W;       // F x d row-major
    int d;
    const float* bias;     // may be null
    const float* cvmean;   // may be null
    const float* cvrange;  // may be null
    float* out;            // may be null
    float* part;           // [blocks][2][D] or null
};

// LDS: Wt [D][Fp] | mean [Fp] | range [Fp]   (Fp = F rounded up to 4)
template <int D, int VEC>
__global__ __launch_bounds__(kProjThreads) void project_kernel(ProjArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int Fp = (a.F + 3) & ~3;
    float* sW = reinterpret_cast<float*>(smem);
    float* sM = sW + D * Fp;
    float* sR = sM + Fp;
    const int t = threadIdx.x;
    for (int i = t; i < D * Fp; i += kProjThreads) {
        const int c = i / Fp, f = i - c * Fp;
        sW[i] = (c < a.d && f < a.F) ? a.W[(int64_t)f * a.d + c] : 0.f;
    }
    for (int f = t; f < Fp; f += kProjThreads) {
        sM[f] = (a.fmean && f < a.F) ? a.fmean[f] : 0.f;
        sR[f] = (a.frange && f < a.F) ? a.frange[f] : 1.f;
    }
    __syncthreads();
    const bool do_norm = a.fmean != nullptr;
    const int lr = t % kLanesPerRow;  // lane within the row group
    const int rg = t / kLanesPerRow;  // row group within the block
    float vmin[D], vmax[D];
#pragma unroll
    for (int c = 0; c < D; ++c) {
        vmin[c] = INFINITY;
        vmax[c] = -INFINITY;
    }
    float cvm[D], cvr[D], bs[D];
#pragma unroll
    for (int c = 0; c < D; ++c) {
        cvm[c] = (a.cvmean && c < a.d) ? a.cvmean[c] : 0.f;
        cvr[c] = (a.cvrange && c < a.d) ? a.cvrange[c] : 1.f;
        bs[c] = (a.bias && c < a.d) ? a.bias[c] : 0.f;
    }
    const bool do_cvnorm = a.cvmean != nullptr;

    const int64_t n_iter = (a.n + kRowsPerIter - 1) / kRowsPerIter;
    for (int64_t it = blockIdx.x; it < n_iter; it += gridDim.x) {
        const int64_t r0 = it * kRowsPerIter + rg;
        float acc[kRowsInFlight][D];
#pragma unroll
        for (int u = 0; u < kRowsInFlight; ++u)
#pragma unroll
            for (int c = 0; c < D; ++c) acc[u][c] = 0.f;
        const float* rowp[kRowsInFlight];
        bool rok[kRowsInFlight];
#pragma unroll
        for (int u = 0; u < kRowsInFlight; ++u) {
            const int64_t r = r0 + (int64_t)u * kRowsPerPass;
            rok[u] = r < a.n;
            rowp[u] = a.X + (rok[u] ? r : 0) * a.ld;
        }
        for (int col = lr * VEC; col < a.F; col += kLanesPerRow * VEC) {
            float x[kRowsInFlight][VEC];
#pragma unroll
            for (int u = 0; u < kRowsInFlight; ++u) {
                if constexpr (VEC == 4) {
                    // unconditional (rows past the end re-read row 0 and are never stored) and non-temporal: the matrix is streamed once
                    typedef float nv4 __attribute__((ext_vector_type(4)));
                    const nv4 qv = __builtin_nontemporal_load(reinterpret_cast<const nv4*>(rowp[u] + col));
                    const float4 q = make_float4(qv.x, qv.y, qv.z, qv.w);
                    x[u][0] = q.x; x[u][1] = q.y; x[u][2] = q.z; x[u][3] = q.w;
                } else {
                    x[u][0] = rok[u] ? rowp[u][col] : 0.f;
                }
            }
            if (do_norm) {
                float m[VEC], s[VEC];
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    m[v] = sM[col + v];
                    s[v] = sR[col + v];
                }
#pragma unroll
                for (int u = 0; u < kRowsInFlight; ++u)
#pragma unroll
                    for (int v = 0; v < VEC; ++v) x[u][v] = __fdiv_rn(__fsub_rn(x[u][v], m[v]), s[v]);
            }
#pragma unroll
            for (int c = 0; c < D; ++c) {
                float w[VEC];
                if constexpr (VEC == 4) {
                    const float4 q = *reinterpret_cast<const float4*>(sW + c * Fp + col);
                    w[0] = q.x; w[1] = q.y; w[2] = q.z; w[3] = q.w;
                } else {
                    w[0] = sW[c * Fp + col];
                }
#pragma unroll
                for (int u = 0; u < kRowsInFlight; ++u)
#pragma unroll
                    for (int v = 0; v < VEC; ++v) acc[u][c] = fmaf(x[u][v], w[v], acc[u][c]);
            }
        }
        // combine the 16 lanes of each row group
#pragma unroll
        for (int u = 0; u < kRowsInFlight; ++u)
#pragma unroll
            for (int c = 0; c < D; ++c) {
                float v = acc[u][c];
                v += __shfl_xor(v, 8, kWave);
                v += __shfl_xor(v, 4, kWave);
                v += __shfl_xor(v, 2, kWave);
                v += __shfl_xor(v, 1, kWave);
                acc[u][c] = v;
            }
        if (lr == 0) {
#pragma unroll
            for (int u = 0; u < kRowsInFlight; ++u) {
                if (!rok[u]) continue;
                const int64_t r = r0 + (int64_t)u * kRowsPerPass;
#pragma unroll
                for (int c = 0; c < D; ++c) {
                    if (c >= a.d) break;
                    float y = acc[u][c] + bs[c];
                    if (do_cvnorm) y = __fdiv_rn(__fsub_rn(y, cvm[c]), cvr[c]);
                    if (a.out) a.out[r * a.d + c] = y;
                    vmin[c] = fminf(vmin[c], y);
                    vmax[c] = fmaxf(vmax[c], y);
                }
            }
        }
    }
    if (a.part) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);  // reuse: [2][D][kRowsPerPass]
        if (lr == 0) {
#pragma unroll
            for (int c = 0; c < D; ++c) {
                red[(0 * D + c) * kRowsPerPass + rg] = vmin[c];
                red[(1 * D + c) * kRowsPerPass + rg] = vmax[c];
            }
        }
        __syncthreads();
        if (t < 2 * D) {
            const int which = t / D, c = t % D;
            float v = which == 0 ? INFINITY : -INFINITY;
            for (int q = 0; q < kRowsPerPass; ++q) {
                const float w = red[(which * D + c) * kRowsPerPass + q];
                v = which == 0 ? fminf(v, w) : fmaxf(v, w);
            }
            a.part[((int64_t)blockIdx.x * 2 + which) * D + c] = v;
        }
    }
}

// one workgroup per (min | max, component): threads take the blocks b = t, t + 256, ..., LDS tree (min / max are exact
// in any order)
template <int D>
__global__ __launch_bounds__(256) void project_minmax_final(const float* __restrict__ part, int nblocks, int d, float* __restrict__ minmax) {
    __shared__ float s_red[256];
    const int t = threadIdx.x;
    const int which = blockIdx.x / d, c = blockIdx.x % d;
    float v = which == 0 ? INFINITY : -INFINITY;
    for (int b = t; b < nblocks; b += 256) {
        const float w = part[((int64_t)b * 2 + which) * D + c];
        v = which == 0 ? fminf(v, w) : fmaxf(v, w);
    }
    s_red[t] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) s_red[t] = which == 0 ? fminf(s_red[t], s_red[t + o]) : fmaxf(s_red[t], s_red[t + o]);
        __syncthreads();
    }
    if (t == 0) minmax[which * d + c] = s_red[0];
}

static int proj_blocks(int64_t n) {
    int64_t b = cdiv(n, kRowsPerIter);
    const int64_t cap = (int64_t)num_cus() * 8;
    if (b > cap) b = cap;
    if (b > kProjMaxBlocks) b = kProjMaxBlocks;
    if (b < 1) b = 1;
    return (int)b;
}

template <int D>
static int launch_project(const ProjArgs& a, int nb, bool vec4, float* minmax, hipStream_t s) {
    const int Fp = (a.F + 3) & ~3;
    size_t lds = (size_t)(D * Fp + 2 * Fp) * sizeof(float);
    const size_t red = (size_t)2 * D * kRowsPerPass * sizeof(float);
    if (lds < red) lds = red;
    DCV_REQUIRE(lds <= 160 * 1024, "dcv_project_linear: F=%d d=%d needs %zu bytes of LDS (max 160 KiB)", a.F, a.d, lds);
    if (vec4) {
        if (lds > 64 * 1024)
            DCV_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&project_kernel<D, 4>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((project_kernel<D, 4>), dim3(nb), dim3(kProjThreads), lds, s, a);
    } else {
        if (lds > 64 * 1024)
            DCV_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&project_kernel<D, 1>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((project_kernel<D, 1>), dim3(nb), dim3(kProjThreads), lds, s, a);
    }
    DCV_CHECK_LAUNCH();
    if (minmax) {
        hipLaunchKernelGGL(project_minmax_final<D>, dim3(2 * a.d), dim3(256), 0, s, a.part, nb, a.d, minmax);
        DCV_CHECK_LAUNCH();
    }
    return DCV_OK;
}

}  // namespace dcv

using namespace dcv;

extern "C" size_t dcv_project_linear_workspace(int64_t n, int32_t F, int32_t d) {
    (void)F;
    if (n <= 0 || d <= 0) return 0;
    return (size_t)kProjMaxBlocks * 2 * 16 * sizeof(float);
}

extern "C" int dcv_project_linear(const float* X_d, int64_t n, int32_t F, int64_t ld, const float* fmean_d,
                                  const float* frange_d, const float* W_d, int32_t d, const float* bias_d,
                                  const float* cvmean_d, const float* cvrange_d, float* out_d, float* minmax_d,
                                  void* ws_d, size_t ws_bytes, void* stream) {
    DCV_REQUIRE(X_d && W_d && n > 0 && F > 0 && ld >= F, "dcv_project_linear: bad arguments");
    DCV_REQUIRE(d >= 1 && d <= 16, "dcv_project_linear: d=%d unsupported (1..16)", d);
    DCV_REQUIRE((fmean_d == nullptr) == (frange_d == nullptr), "dcv_project_linear: fmean/frange must come together");
    DCV_REQUIRE((cvmean_d == nullptr) == (cvrange_d == nullptr), "dcv_project_linear: cvmean/cvrange must come together");
    DCV_REQUIRE(out_d || minmax_d, "dcv_project_linear: nothing to compute");
    if (minmax_d) DCV_REQUIRE(ws_d && ws_bytes >= dcv_project_linear_workspace(n, F, d), "dcv_project_linear: workspace too small");
    hipStream_t s = as_stream(stream);
    ProjArgs a{X_d, n, F, ld, fmean_d, frange_d, W_d, d, bias_d, cvmean_d, cvrange_d, out_d,
               minmax_d ? static_cast<float*>(ws_d) : nullptr};
    const int nb = proj_blocks(n);
    const bool vec4 = (F % 4 == 0) && (ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(X_d) & 15) == 0);
    if (d <= 2) return launch_project<2>(a, nb, vec4, minmax_d, s);
    if (d <= 4) return launch_project<4>(a, nb, vec4, minmax_d, s);
    if (d <= 8) return launch_project<8>(a, nb, vec4, minmax_d, s);
    return launch_project<16>(a, nb, vec4, minmax_d, s);
}
