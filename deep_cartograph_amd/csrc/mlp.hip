// MLP engine: Deep-TICA and autoencoder training / inference on the FP32 MFMA block engine.
//
// A step never materialises the batch: the first layer gathers its rows straight from the
// resident, pre-normalised feature matrix through a RowMap (x_t rows, then the same samples
// `lag` rows later for Deep-TICA).  Every Linear layer is three products on the same engine
//   forward  H_l  = act(In_l W_l^T + b_l)                      NT, fused bias + activation
//   dgrad    dZ_{l-1} = (dZ_l W_l) * act'(H_{l-1})             NN, fused activation gradient
//   wgrad    dW_l = dZ_l^T In_l                                TN, split over the batch rows
// wgrad split partials and the bias column sums are folded in a fixed order by one reduction
// kernel (deterministic), Adam is one more launch.  The d x d TICA algebra of the Deep-TICA
// loss runs in float64 on the device from batch statistics that a data-parallel caller
// all-reduces, so nothing returns to the host inside an epoch.
#include "mlp_state.h"
#include "tica_head.h"
#include <new>
#include <math.h>

using namespace dcv;

namespace dcv {

constexpr int kColsumRows = 32;    // rows per block of colsum_kernel (a block walks its rows serially: short blocks, many of them)
constexpr int kSseRows = 16;       // rows per block of ae_sse_kernel
constexpr int kStatBlockRows = 128;

// ------------------------------------------------------------------ small kernels
// partial column sums of dZ (rows x n): part[block][n], one block per kColsumRows rows.  Threads are
// laid out (row group, column): wide matrices give every thread one column, narrow ones (the d
// outputs of the last layer) put many row groups on one column and combine them through LDS.
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ Z, int64_t rows, int n, int64_t ld,
                                                     float* __restrict__ part) {
    __shared__ float red[256];
    const int64_t r0 = (int64_t)blockIdx.x * kColsumRows;
    const int64_t r1 = r0 + kColsumRows < rows ? r0 + kColsumRows : rows;
    const int t = threadIdx.x;
    const int cols = n < 256 ? n : 256;   // columns handled per pass
    const int groups = 256 / cols;        // row groups per column
    const int c_in = t % cols, g = t / cols;
    for (int c0 = 0; c0 < n; c0 += cols) {
        const int c = c0 + c_in;
        float s = 0.f;
        if (g < groups && c < n) {
            // eight loads in flight, added in row order (the same sum as a serial walk, without its dependent round trips)
            for (int64_t rb = r0 + g; rb < r1; rb += 8 * (int64_t)groups) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int64_t r = rb + (int64_t)u * groups;
                    v[u] = r < r1 ? Z[r * ld + c] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) s += v[u];
            }
        }
        red[t] = s;
        __syncthreads();
        if (g == 0 && c < n) {
            float tot = 0.f;
            for (int q = 0; q < groups; ++q) tot += red[q * cols + c_in];
            part[(int64_t)blockIdx.x * n + c] = tot;
        }
        __syncthreads();
    }
}

struct ReduceDesc {
    const float* slab;   // [splits][count]
    const float* bpart;  // [bblocks][out]
    int64_t w_off, b_off;
    int64_t w_count;     // out*in
    int out;
    int splits, bblocks;
    int64_t w_stride, b_stride;   // floats between consecutive partials (0: dense = w_count / out).  The fused small-network
                                  // kernels pad them to multiples of 4 so that 16-byte loads work for any layer size (15-wide
                                  // layers: 810 = 54 * 15 weights per partial took the scalar walk, 15 us per reduction)
};
static inline int64_t rd_wstride(const ReduceDesc& d) { return d.w_stride > 0 ? d.w_stride : d.w_count; }
static inline int64_t rd_bstride(const ReduceDesc& d) { return d.b_stride > 0 ? d.b_stride : (int64_t)d.out; }
struct ReduceArgs {
    ReduceDesc l[2 * DCV_MAX_LAYERS];   // [0, L): the Linear layers; [L, 2 L): weight / bias of the batch normalisation behind layer l - L (empty without one)
    int L;
};

// torch.optim single-tensor updates (CPU code path of torch 2.x: _single_tensor_adam / _adamw / _sgd / _rmsprop /
// _adagrad), fp32 state.  One thread per element; `s1`, `s2`, `s3` are the optimiser's state tensors.
struct OptArgs {
    int kind, flag;   // DCV_OPT_*; flag: amsgrad (Adam family), nesterov (SGD), centered (RMSprop)
    int first;        // SGD: first step (momentum buffer := gradient)
    float lr, b1, b2, eps, wd;
    float c1, c2;     // Adam family: lr / (1 - b1^t), sqrt(1 - b2^t); Adagrad: c1 = lr / (1 + (t - 1) lr_decay)
    // scalars torch forms in Python doubles and then hands to a float32 kernel: computed on the host in double and
    // rounded once, exactly as there ((float)(1 - 0.999) is not 1.f - 0.999f)
    float w1, w2;     // 1 - beta1 (Adam) / 1 - dampening (SGD) ; 1 - beta2 (Adam) / 1 - alpha (RMSprop)
    float decay;      // AdamW: 1 - lr * weight_decay
    float p0, p1, p2, p3;   // further per-step scalars of Adamax / NAdam / RAdam / Adadelta / ASGD / Rprop (next_opt_args)
    int maximize;           // torch.optim's maximize: the update runs on the negated gradient
    // LDS image of the fused small-network kernels (snet.h: snet_image_build): every updated parameter is mirrored into the
    // zero-padded weight image those kernels stage with one contiguous copy, at img[img_idx[i]] (img_idx[i] < 0: not in it)
    float* img;
    const int* img_idx;
};
// pi = p[i], loaded by the caller (the reduction kernels issue that load before they wait for the partial sums)
// WT: write-through stores (the launch then ends without dirty lines to write back: reduce_grads_quad_kernel)
template <bool WT>
__device__ __forceinline__ void opt_st(float* p, float v) {
    if constexpr (WT) handoff_store(p, v);
    else *p = v;
}
template <bool WT>
__device__ __forceinline__ void opt_stp(const OptArgs& a, float* p, int64_t i, float v) {
    opt_st<WT>(p + i, v);
    if (a.img != nullptr) {
        const int j = a.img_idx[i];
        if (j >= 0) opt_st<WT>(a.img + j, v);
    }
}
template <bool WT = false>
__device__ __forceinline__ void opt_update_p(int64_t i, float gi, float pi, float* __restrict__ p, float* __restrict__ s1, float* __restrict__ s2,
                                             float* __restrict__ s3, const OptArgs& a) {
    if (a.maximize) gi = -gi;   // `grad = grads[i] if not maximize else -grads[i]`: the first line of every _single_tensor_* update
    switch (a.kind) {
        case DCV_OPT_ADAM:
        case DCV_OPT_ADAMW: {
            if (a.kind == DCV_OPT_ADAMW) pi = pi * a.decay;                        // param.mul_(1 - lr * weight_decay)
            else if (a.wd != 0.f) gi = fmaf(a.wd, pi, gi);                        // grad.add(param, alpha=weight_decay)
            float mi = s1[i], vi = s2[i];
            mi = mi + (gi - mi) * a.w1;                                           // exp_avg.lerp_(grad, 1 - beta1)
            vi = vi * a.b2 + a.w2 * gi * gi;                                      // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
            float vden = vi;
            if (a.flag) {                                                          // amsgrad: max_exp_avg_sq = max(., exp_avg_sq)
                vden = fmaxf(s3[i], vi);
                opt_st<WT>(s3 + i, vden);
            }
            const float denom = sqrtf(vden) / a.c2 + a.eps;
            opt_st<WT>(s1 + i, mi);
            opt_st<WT>(s2 + i, vi);
            opt_stp<WT>(a, p, i, pi - a.c1 * (mi / denom));                                      // param.addcdiv_(exp_avg, denom, value=-step_size)
            break;
        }
        case DCV_OPT_SGD: {
            if (a.wd != 0.f) gi = fmaf(a.wd, pi, gi);
            if (a.b1 != 0.f) {                                                     // b1 = momentum, w1 = 1 - dampening
                float bi = a.first ? gi : s1[i] * a.b1 + a.w1 * gi;               // buf.mul_(momentum).add_(grad, alpha=1 - dampening)
                opt_st<WT>(s1 + i, bi);
                gi = a.flag ? fmaf(a.b1, bi, gi) : bi;                            // nesterov: grad.add(buf, alpha=momentum)
            }
            opt_stp<WT>(a, p, i, pi - a.lr * gi);
            break;
        }
        case DCV_OPT_RMSPROP: {
            if (a.wd != 0.f) gi = fmaf(a.wd, pi, gi);
            float sq = s2[i] * a.b2 + a.w2 * gi * gi;                             // square_avg.mul_(alpha).addcmul_(grad, grad, 1 - alpha)
            opt_st<WT>(s2 + i, sq);
            float avg;
            if (a.flag) {                                                          // centered
                float ga = s3[i];
                ga = ga + (gi - ga) * a.w2;                                        // grad_avg.lerp_(grad, 1 - alpha)
                opt_st<WT>(s3 + i, ga);
                avg = sqrtf(sq - ga * ga) + a.eps;                                 // addcmul(grad_avg, grad_avg, -1).sqrt_().add_(eps)
            } else {
                avg = sqrtf(sq) + a.eps;
            }
            if (a.b1 > 0.f) {                                                      // b1 = momentum
                const float bi = s1[i] * a.b1 + gi / avg;                          // buf.mul_(momentum).addcdiv_(grad, avg)
                opt_st<WT>(s1 + i, bi);
                opt_stp<WT>(a, p, i, pi - a.lr * bi);
            } else {
                opt_stp<WT>(a, p, i, pi - a.lr * (gi / avg));
            }
            break;
        }
        case DCV_OPT_ADAMAX: {   // _single_tensor_adamax: s1 = exp_avg, s2 = exp_inf
            if (a.wd != 0.f) gi = fmaf(a.wd, pi, gi);
            float mi = s1[i];
            mi = mi + (gi - mi) * a.w1;                                           // exp_avg.lerp_(grad, 1 - beta1)
            const float ui = fmaxf(s2[i] * a.b2, fabsf(gi) + a.eps);              // maximum(exp_inf * beta2, |grad| + eps)
            opt_st<WT>(s1 + i, mi);
            opt_st<WT>(s2 + i, ui);
            opt_stp<WT>(a, p, i, pi - a.c1 * (mi / ui));                                         // addcdiv_(exp_avg, exp_inf, value=-lr / bias_correction)
            break;
        }
        case DCV_OPT_NADAM: {    // _single_tensor_nadam: p0 = -lr (1 - mu) / (1 - mu_product), p1 = -lr mu_next / (1 - mu_product_next), c2 = 1 - beta2^t
            if (a.wd != 0.f) {
                if (a.flag) pi = pi * a.decay;                                     // decoupled: param.mul_(1 - lr * weight_decay)
                else gi = fmaf(a.wd, pi, gi);
            }
            float mi = s1[i], vi = s2[i];
            mi = mi + (gi - mi) * a.w1;
            vi = vi * a.b2 + a.w2 * gi * gi;
            const float denom = sqrtf(vi / a.c2) + a.eps;                          // exp_avg_sq.div(bias_correction2).sqrt().add(eps)
            opt_st<WT>(s1 + i, mi);
            opt_st<WT>(s2 + i, vi);
            pi = pi + a.p0 * (gi / denom);
            opt_stp<WT>(a, p, i, pi + a.p1 * (mi / denom));
            break;
        }
        case DCV_OPT_RADAM: {    // _single_tensor_radam: c1 = 1 - beta1^t, c2 = sqrt(1 - beta2^t), p0 = rect (0: rho_t <= 5)
            if (a.wd != 0.f) {
                if (a.flag) pi = pi * a.decay;
                else gi = fmaf(a.wd, pi, gi);
            }
            float mi = s1[i], vi = s2[i];
            mi = mi + (gi - mi) * a.w1;
            vi = vi * a.b2 + a.w2 * gi * gi;
            opt_st<WT>(s1 + i, mi);
            opt_st<WT>(s2 + i, vi);
            const float mhat = mi / a.c1;
            if (a.p0 > 0.f) opt_stp<WT>(a, p, i, pi - ((mhat * a.lr) * (a.c2 / (sqrtf(vi) + a.eps))) * a.p0);
            else opt_stp<WT>(a, p, i, pi - mhat * a.lr);
            break;
        }
        case DCV_OPT_ADADELTA: { // _single_tensor_adadelta: s1 = square_avg, s2 = acc_delta, b2 = rho, w2 = 1 - rho
            if (a.wd != 0.f) gi = fmaf(a.wd, pi, gi);
            const float sq = s1[i] * a.b2 + a.w2 * gi * gi;
            const float acc = s2[i];
            const float delta = sqrtf(acc + a.eps) / sqrtf(sq + a.eps) * gi;
            opt_st<WT>(s1 + i, sq);
            opt_st<WT>(s2 + i, acc * a.b2 + a.w2 * delta * delta);
            opt_stp<WT>(a, p, i, pi - a.lr * delta);
            break;
        }
        case DCV_OPT_ASGD: {     // _single_tensor_asgd: p0 = 1 - lambd * eta, p1 = eta (the averaged copy ax is not kept)
            if (a.wd != 0.f) gi = fmaf(a.wd, pi, gi);
            pi = pi * a.p0;
            opt_stp<WT>(a, p, i, pi - a.p1 * gi);
            break;
        }
        case DCV_OPT_RPROP: {    // _single_tensor_rprop: s1 = prev, s2 = step_size; p0 / p1 = eta minus / plus, p2 / p3 = step bounds
            const float sg = gi * s1[i];
            const float f = sg > 0.f ? a.p1 : (sg < 0.f ? a.p0 : 1.f);
            const float st = fminf(fmaxf(s2[i] * f, a.p2), a.p3);
            opt_st<WT>(s2 + i, st);
            if (sg < 0.f) gi = 0.f;
            const float sgn = gi > 0.f ? 1.f : (gi < 0.f ? -1.f : 0.f);
            opt_stp<WT>(a, p, i, pi - sgn * st);
            opt_st<WT>(s1 + i, gi);
            break;
        }
        default: {   // DCV_OPT_ADAGRAD
            if (a.wd != 0.f) gi = fmaf(a.wd, pi, gi);
            const float su = s2[i] + gi * gi;                                      // state_sum.addcmul_(grad, grad, value=1)
            opt_st<WT>(s2 + i, su);
            opt_stp<WT>(a, p, i, pi - a.c1 * (gi / (sqrtf(su) + a.eps)));                         // param.addcdiv_(grad, std, value=-clr)
            break;
        }
    }
}
__device__ __forceinline__ void opt_update(int64_t i, float gi, float* __restrict__ p, float* __restrict__ s1, float* __restrict__ s2,
                                           float* __restrict__ s3, const OptArgs& a) {
    opt_update_p(i, gi, p[i], p, s1, s2, s3, a);
}
__global__ __launch_bounds__(256) void optimizer_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ s1,
                                                        float* __restrict__ s2, float* __restrict__ s3, int64_t n, OptArgs a) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) opt_update(i, g[i], p, s1, s2, s3, a);
}

// grads[w] = sum_s slab[s][w] ; grads[b] = sum_blk bpart[blk][b].  A block covers 64 consecutive
// elements; its 16 waves take the partials q = wave, wave + 16, ... (coalesced 256-byte reads)
// and are combined in wave order: float64 accumulation, fixed order, deterministic.
constexpr int kRedWaves = 16;
// fuse != 0 (one-GPU training step, nothing to all-reduce in between): the thread that finishes a gradient element
// applies the optimiser update to its parameter at once -- one launch less per step.
__global__ __launch_bounds__(64 * kRedWaves) void reduce_grads_kernel(ReduceArgs a, float* __restrict__ grads, float scale, int fuse,
                                                                      float* __restrict__ params, float* __restrict__ s1,
                                                                      float* __restrict__ s2, float* __restrict__ s3, OptArgs oa) {
    __shared__ double s_red[kRedWaves][64];
    const int l = blockIdx.y;
    const ReduceDesc& d = a.l[l];
    const int64_t total = d.w_count + d.out;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t base = (int64_t)blockIdx.x * 64; base < total; base += (int64_t)gridDim.x * 64) {
        const int64_t i = base + lane;
        double s = 0.0;
        if (i < d.w_count) {
            const float* p = d.slab + i;
#pragma unroll 8
            for (int q = wave; q < d.splits; q += kRedWaves) s += (double)p[(int64_t)q * (d.w_stride > 0 ? d.w_stride : d.w_count)];
        } else if (i < total) {
            const float* p = d.bpart + (i - d.w_count);
#pragma unroll 8
            for (int q = wave; q < d.bblocks; q += kRedWaves) s += (double)p[(int64_t)q * (d.b_stride > 0 ? d.b_stride : (int64_t)d.out)];
        }
        s_red[wave][lane] = s;
        __syncthreads();
        if (wave == 0 && i < total) {
            double tot = 0.0;
#pragma unroll
            for (int w = 0; w < kRedWaves; ++w) tot += s_red[w][lane];
            const float g = (float)(tot * (double)scale);
            const int64_t pidx = i < d.w_count ? d.w_off + i : d.b_off + (i - d.w_count);
            grads[pidx] = g;
            if (fuse) opt_update(pidx, g, params, s1, s2, s3, oa);
        }
        __syncthreads();
    }
}

// The same reduction for small split counts (small batches: the 16-wave form above spends a 1024-thread block on 64
// outputs and a handful of partials): a block covers 64 consecutive elements, its four waves take the partials
// q = wave, wave + 4, ... (independent loads, eight in flight per thread: a serial walk over the 257 partials of the
// fused last-layer pass cost 64 us of load latency) and are combined in wave order; float64, fixed order, optional
// fused optimiser update.
__global__ __launch_bounds__(256) void reduce_grads_small_kernel(ReduceArgs a, float* __restrict__ grads, float scale, int fuse,
                                                                 float* __restrict__ params, float* __restrict__ s1,
                                                                 float* __restrict__ s2, float* __restrict__ s3, OptArgs oa) {
    __shared__ double s_red[4][64];
    const ReduceDesc& d = a.l[blockIdx.y];
    const int64_t total = d.w_count + d.out;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t base = (int64_t)blockIdx.x * 64; base < total; base += (int64_t)gridDim.x * 64) {
        const int64_t i = base + lane;
        double s = 0.0;
        if (i < d.w_count) {
            const float* p = d.slab + i;
#pragma unroll 8
            for (int q = wave; q < d.splits; q += 4) s += (double)p[(int64_t)q * (d.w_stride > 0 ? d.w_stride : d.w_count)];
        } else if (i < total) {
            const float* p = d.bpart + (i - d.w_count);
#pragma unroll 8
            for (int q = wave; q < d.bblocks; q += 4) s += (double)p[(int64_t)q * (d.b_stride > 0 ? d.b_stride : (int64_t)d.out)];
        }
        s_red[wave][lane] = s;
        __syncthreads();
        if (wave == 0 && i < total) {
            const double tot = ((s_red[0][lane] + s_red[1][lane]) + s_red[2][lane]) + s_red[3][lane];
            const float g = (float)(tot * (double)scale);
            const int64_t pidx = i < d.w_count ? d.w_off + i : d.b_off + (i - d.w_count);
            grads[pidx] = g;
            if (fuse) opt_update(pidx, g, params, s1, s2, s3, oa);
        }
        __syncthreads();
    }
}

// The small-split reduction on a flat grid with 16-byte loads.  The weights and the biases of every entry are separate
// items {partials, count, number of partials}; the grid is the concatenation of the items' blocks (no empty workgroups
// for the narrow layers).  A block of 256 threads covers 1024 / G consecutive elements with G groups of threads, group g
// taking the partials q = g, g + G, ... (four consecutive elements per thread: one global_load_dwordx4 per partial
// where the item allows); G = 4 for few partials, 16 when an item has more than 32 (the 129 bias partials of the
// 64-row tiles, the 257 of the fused last-layer pass: walked by 4 groups they are a chain of 64 dependent-latency
// loads, the longest path of the launch).  After the exchange every thread of the first 1024 / G finishes ONE element,
// groups combined in order, float64: deterministic -- and its parameter load was issued before the partials were
// waited for.
struct QuadItem {
    const float* src;   // [parts][stride], count <= stride values used
    int64_t dst;        // offset of element 0 in grads / params
    int64_t stride;     // floats between partials
    int count, parts;
    int blk0;           // first block of this item in the grid
    int groups;         // G
};
struct QuadArgs {
    QuadItem it[4 * DCV_MAX_LAYERS];
    int n;
};
inline int quad_groups(int parts) { return parts > 32 ? 16 : 4; }
template <int G>
__device__ __forceinline__ void reduce_quad_block(const QuadItem& d, int blk, float* __restrict__ grads, float scale, int fuse,
                                                  float* __restrict__ params, float* __restrict__ s1, float* __restrict__ s2,
                                                  float* __restrict__ s3, const OptArgs& oa, double* s_red) {
    constexpr int EPB = 1024 / G, TPG = EPB / 4;   // elements per block, threads per group
    const int t = threadIdx.x, g = t / TPG, sub = t % TPG;
    const int base = blk * EPB;
    const int mine = base + t;
    const bool fin = t < EPB && mine < d.count;
    float pi = 0.f;
    if (fuse && fin) pi = params[d.dst + mine];
    const int e0 = base + 4 * sub;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    if (e0 < d.count && (d.stride & 3) == 0 && e0 + 4 <= d.stride && (reinterpret_cast<uintptr_t>(d.src) & 15) == 0) {
        // (the last quad of a partial may reach into its padding: those elements are summed and never finished)
        const float* p = d.src + e0;
#pragma unroll 8
        for (int q = g; q < d.parts; q += G) {
            const float4 v = *reinterpret_cast<const float4*>(p + (int64_t)q * d.stride);
            acc[0] += (double)v.x;
            acc[1] += (double)v.y;
            acc[2] += (double)v.z;
            acc[3] += (double)v.w;
        }
    } else if (e0 < d.count) {
        const float* p = d.src + e0;
        const int nv = d.count - e0 < 4 ? d.count - e0 : 4;
#pragma unroll 4
        for (int q = g; q < d.parts; q += G) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (j < nv) acc[j] += (double)p[(int64_t)q * d.stride + j];
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) s_red[g * EPB + 4 * sub + j] = acc[j];
    __syncthreads();
    if (fin) {
        double tot = s_red[t];
#pragma unroll
        for (int k = 1; k < G; ++k) tot += s_red[k * EPB + t];
        const float gr = (float)(tot * (double)scale);
        handoff_store(grads + d.dst + mine, gr);
        if (fuse) opt_update_p<true>(d.dst + mine, gr, pi, params, s1, s2, s3, oa);
    }
}
__global__ __launch_bounds__(256) void reduce_grads_quad_kernel(QuadArgs a, float* __restrict__ grads, float scale, int fuse,
                                                                float* __restrict__ params, float* __restrict__ s1,
                                                                float* __restrict__ s2, float* __restrict__ s3, OptArgs oa) {
    __shared__ double s_red[1024];
    int l = 0;
    while (l + 1 < a.n && (int)blockIdx.x >= a.it[l + 1].blk0) ++l;   // uniform
    const QuadItem& d = a.it[l];
    const int blk = (int)blockIdx.x - d.blk0;
    if (d.groups == 16) reduce_quad_block<16>(d, blk, grads, scale, fuse, params, s1, s2, s3, oa, s_red);
    else reduce_quad_block<4>(d, blk, grads, scale, fuse, params, s1, s2, s3, oa, s_red);
}

// ------------------------------------------------------------------ Deep-TICA batch statistics
// F: f_t of sample r in row r, f_lag in row r + lag_off, d columns (lag_off = B when the two halves
// of the batch are separate rows, = lag when a contiguous batch shares its rows: see dcv_mlp_forward).
// Each block stages kStatBlockRows pairs in
// LDS (float64) and every thread owns whole outputs of [sum f_t | sum f_lag | sum f_t f_t^T |
// sum f_t f_lag^T]; part[block][2d + 2d^2] float64, combined in block order afterwards.
__global__ __launch_bounds__(256) void tica_stats_kernel(const float* __restrict__ F, int64_t ld, int B, int d, int lag_off,
                                                         double* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* s_t = reinterpret_cast<double*>(smem);   // [rows][d]
    double* s_l = s_t + kStatBlockRows * d;           // [rows][d]
    const int W = 2 * d + 2 * d * d;
    const int t = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * kStatBlockRows;
    const int nr = (int)(r0 + kStatBlockRows < B ? kStatBlockRows : B - r0);
    for (int i = t; i < nr * d; i += 256) {
        const int r = i / d, c = i - r * d;
        s_t[i] = (double)F[(r0 + r) * ld + c];
        s_l[i] = (double)F[(r0 + r + lag_off) * ld + c];   // lag_off = B (two halves) or lag (shared rows)
    }
    __syncthreads();
    double* my = part + (int64_t)blockIdx.x * W;
    for (int o = t; o < W; o += 256) {
        double s = 0.0;
        if (o < d) {
            for (int r = 0; r < nr; ++r) s += s_t[r * d + o];
        } else if (o < 2 * d) {
            for (int r = 0; r < nr; ++r) s += s_l[r * d + o - d];
        } else if (o < 2 * d + d * d) {
            const int q = o - 2 * d, i = q / d, j = q % d;
            for (int r = 0; r < nr; ++r) s += s_t[r * d + i] * s_t[r * d + j];
        } else {
            const int q = o - 2 * d - d * d, i = q / d, j = q % d;
            for (int r = 0; r < nr; ++r) s += s_t[r * d + i] * s_l[r * d + j];
        }
        my[o] = s;
    }
}

// dZ_last[r][c] = g[c] * act'(H_last[r][c]): the gradient of s = sum_j cv_j w.r.t. the network output is the
// same vector for every frame (the layers after the network are affine)
__global__ __launch_bounds__(256) void seed_grad_kernel(const float* __restrict__ H, int64_t ldh, int64_t rows, int d, int act,
                                                        const float* __restrict__ g, float* __restrict__ dZ, int64_t ld_dz) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * d) return;
    const int64_t r = i / d;
    const int c = (int)(i - r * d);
    dZ[r * ld_dz + c] = g[c] * act_grad_from_out(act, H[r * ldh + c]);
}

// part[block][c] = sum over the block's rows of |G[r][c]| * scale[c] (float64, rows in index order)
constexpr int kAbsRows = 256;
__global__ __launch_bounds__(256) void abs_colsum_kernel(const float* __restrict__ G, int64_t ldg, int64_t rows, int F,
                                                         const float* __restrict__ scale, double* __restrict__ part) {
    const int64_t r0 = (int64_t)blockIdx.x * kAbsRows;
    const int64_t r1 = r0 + kAbsRows < rows ? r0 + kAbsRows : rows;
    for (int c = threadIdx.x; c < F; c += 256) {
        double acc = 0.0;
        const double sc = (double)scale[c];
        for (int64_t r = r0; r < r1; ++r) acc += fabs((double)G[r * ldg + c]) * sc;
        part[(int64_t)blockIdx.x * F + c] = acc;
    }
}

// out[i] = sum_b part[b][i], one wave per output, fixed combination tree
__global__ __launch_bounds__(64) void sum_partials_kernel(const double* __restrict__ part, int nblocks, int width,
                                                          double* __restrict__ out) {
    const int i = blockIdx.x;
    if (i >= width) return;
    const int lane = threadIdx.x;
    double s = 0.0;
    for (int b = lane; b < nblocks; b += 64) s += part[(int64_t)b * width + i];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) out[i] = s;
}

// One thread: C0, Ctau, loss = -tr((A Ctau)^2) with A = (C0 + reg I)^-1, and the matrices that
// turn (f_t - mu, f_lag - mu) into dL/df (see DESIGN.md "Deep-TICA gradient").  DT > 0 fixes the
// dimension at compile time (everything in registers); DT == 0 is the generic d <= 16 form.
template <int DT>
__device__ __forceinline__ void tica_grad_body(const double* __restrict__ stats, int d_rt, double Bg, double reg, double* __restrict__ gradp,
                                               double* __restrict__ log, int* __restrict__ log_count, int log_cap, int log_width) {
    constexpr int DM = DT > 0 ? DT : kMaxTicaDim;
    const int d = DT > 0 ? DT : d_rt;
    // one thread, a dependent chain: float64 divisions (a ~30-instruction sequence each) are replaced by multiplications
    // with 1 / B and the reciprocals of the Cholesky diagonal -- 1 + d divisions instead of ~8 d^2
    double mu[DM], ml[DM], invL[DM];
    double C0[DM * DM], Ct[DM * DM], A[DM * DM], K[DM * DM], T[DM * DM], Lc[DM * DM];
    const double invB = 1.0 / Bg;
    const double* sft = stats;
    const double* sfl = stats + d;
    const double* Stt = stats + 2 * d;
    const double* Stl = stats + 2 * d + d * d;
#pragma unroll
    for (int i = 0; i < DM; ++i)
        if (i < d) {
            mu[i] = sft[i] * invB;
            ml[i] = sfl[i] * invB;
        }
#pragma unroll
    for (int i = 0; i < DM; ++i)
#pragma unroll
        for (int j = 0; j < DM; ++j)
            if (i < d && j < d) {
                C0[i * DM + j] = 0.5 * (Stt[i * d + j] + Stt[j * d + i]) * invB - mu[i] * mu[j];
                const double cij = Stl[i * d + j] * invB - mu[i] * ml[j];
                const double cji = Stl[j * d + i] * invB - mu[j] * ml[i];
                Ct[i * DM + j] = 0.5 * (cij + cji);
            }
    // Cholesky of C0 + reg I
    bool ok = true;
#pragma unroll
    for (int i = 0; i < DM; ++i)
#pragma unroll
        for (int j = 0; j < DM; ++j)
            if (i < d && j <= i) {
                double s = C0[i * DM + j] + (i == j ? reg : 0.0);
#pragma unroll
                for (int k = 0; k < DM; ++k)
                    if (k < j) s -= Lc[i * DM + k] * Lc[j * DM + k];
                if (i == j) {
                    if (!(s > 0.0)) ok = false;
                    Lc[i * DM + i] = sqrt(s);
                    invL[i] = 1.0 / Lc[i * DM + i];
                } else {
                    Lc[i * DM + j] = s * invL[j];
                }
            }
    // A = (L L^T)^-1 : solve L Y = I, then L^T A = Y
#pragma unroll
    for (int c = 0; c < DM; ++c)
        if (c < d) {
            double y[DM];
#pragma unroll
            for (int i = 0; i < DM; ++i)
                if (i < d) {
                    double s = (i == c) ? 1.0 : 0.0;
#pragma unroll
                    for (int k = 0; k < DM; ++k)
                        if (k < i) s -= Lc[i * DM + k] * y[k];
                    y[i] = s * invL[i];
                }
#pragma unroll
            for (int ii = 0; ii < DM; ++ii) {
                const int i = DM - 1 - ii;
                if (i < d) {
                    double s = y[i];
#pragma unroll
                    for (int k = 0; k < DM; ++k)
                        if (k > i && k < d) s -= Lc[k * DM + i] * A[k * DM + c];
                    A[i * DM + c] = s * invL[i];
                }
            }
        }
    // K = A Ct ; loss = -tr(K K)
#pragma unroll
    for (int i = 0; i < DM; ++i)
#pragma unroll
        for (int j = 0; j < DM; ++j)
            if (i < d && j < d) {
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < DM; ++k)
                    if (k < d) s += A[i * DM + k] * Ct[k * DM + j];
                K[i * DM + j] = s;
            }
    double loss = 0.0;
#pragma unroll
    for (int i = 0; i < DM; ++i)
#pragma unroll
        for (int j = 0; j < DM; ++j)
            if (i < d && j < d) loss -= K[i * DM + j] * K[j * DM + i];
    if (!ok) loss = NAN;
    if (gradp) {
        // T = K A  (= A Ct A, symmetric) ; Gtau = -2 T ; G0 = 2 K T
#pragma unroll
        for (int i = 0; i < DM; ++i)
#pragma unroll
            for (int j = 0; j < DM; ++j)
                if (i < d && j < d) {
                    double s = 0.0;
#pragma unroll
                    for (int k = 0; k < DM; ++k)
                        if (k < d) s += K[i * DM + k] * A[k * DM + j];
                    T[i * DM + j] = s;
                }
        double* g_mu = gradp;
        double* g_u = gradp + d;
        double* g_v = g_u + d * d;
        double* g_c = g_v + d * d;
#pragma unroll
        for (int i = 0; i < DM; ++i)
            if (i < d) {
                g_mu[i] = mu[i];
                double cs = 0.0;
#pragma unroll
                for (int j = 0; j < DM; ++j)
                    if (j < d) {
                        double g0 = 0.0;
#pragma unroll
                        for (int k = 0; k < DM; ++k)
                            if (k < d) g0 += K[i * DM + k] * T[k * DM + j];
                        const double Gt = -(T[i * DM + j] + T[j * DM + i]);  // -2 * sym(T)
                        g_u[i * d + j] = 4.0 * g0 * invB;                     // (2/B) G0, G0 = 2 K T
                        g_v[i * d + j] = Gt * invB;                           // (1/B) Gtau
                        cs += Gt * (ml[j] - mu[j]);
                    }
                g_c[i] = -cs * invB;
            }
    }
    const int slot = *log_count;
    if (slot < log_cap) {
        double* rec = log + (int64_t)slot * log_width;
        rec[0] = loss;
        rec[1] = Bg;
#pragma unroll
        for (int i = 0; i < DM; ++i)
#pragma unroll
            for (int j = 0; j < DM; ++j)
                if (i < d && j < d) {
                    rec[2 + i * d + j] = C0[i * DM + j];
                    rec[2 + d * d + i * d + j] = Ct[i * DM + j];
                }
#pragma unroll
        for (int i = 0; i < DM; ++i)
            if (i < d) rec[2 + 2 * d * d + i] = mu[i];
    }
    *log_count = slot + 1;
}
template <int DT>
__global__ void tica_grad_kernel(const double* __restrict__ stats, int d_rt, double Bg, double reg, double* __restrict__ gradp,
                                 double* __restrict__ log, int* __restrict__ log_count, int log_cap, int log_width) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    tica_grad_body<DT>(stats, d_rt, Bg, reg, gradp, log, log_count, log_cap, log_width);
}

typedef void (*TicaGradFn)(const double*, int, double, double, double*, double*, int*, int, int);
static TicaGradFn tica_grad_fn(int d) {
    switch (d) {
        case 1: return tica_grad_kernel<1>;
        case 2: return tica_grad_kernel<2>;
        case 3: return tica_grad_kernel<3>;
        case 4: return tica_grad_kernel<4>;
        case 5: return tica_grad_kernel<5>;
        case 6: return tica_grad_kernel<6>;
        default: return tica_grad_kernel<0>;
    }
}

// the wave-parallel loss head as a launch of its own: the data-parallel path, where the batch statistics are all-reduced
// between the statistics kernel and the head (the single-thread form above is a chain of ~2000 dependent float64
// instructions, 8 us; this one ~3 us)
template <int D>
__global__ __launch_bounds__(64) void tica_grad_wave_kernel(const double* __restrict__ stats, double Bg, double reg, double* __restrict__ gradp,
                                                            double* __restrict__ log, int* __restrict__ log_count, int log_cap, int log_width) {
    __shared__ TicaWaveLds<D> s_head;
    __shared__ double s_stats[2 * D + 2 * D * D];
    if (threadIdx.x < 2 * D + 2 * D * D) s_stats[threadIdx.x] = stats[threadIdx.x];
    wave_sync_lds();
    tica_grad_wave<D>(s_head, s_stats, Bg, reg, gradp, log, log_count, log_cap, log_width, (int)threadIdx.x);
}
typedef void (*TicaGradWaveFn)(const double*, double, double, double*, double*, int*, int, int);
static TicaGradWaveFn tica_grad_wave_fn(int d) {
    switch (d) {
        case 1: return tica_grad_wave_kernel<1>;
        case 2: return tica_grad_wave_kernel<2>;
        case 3: return tica_grad_wave_kernel<3>;
        case 4: return tica_grad_wave_kernel<4>;
        default: return nullptr;
    }
}

// The same statistics for D <= 4 outputs with every thread at work: a thread walks whole rows (its pair's
// 2 D values, 2 D + 2 D^2 float64 accumulators in registers), waves combine by shuffles, the block through
// LDS.  rows_per_block pairs per block (a multiple of 256; stats_plan): enough blocks to spread a small batch over
// the chip, few enough partials for the last block's ordered sum.  One launch: the block that finishes last adds the
// partials up in block order and -- on one GPU, where nothing is all-reduced in between (fused.on) -- goes straight
// on to the d x d loss head (tica_grad_body), saving the launch of tica_grad_kernel.
template <int D>
__global__ __launch_bounds__(256) void tica_stats_rows_kernel(const float* __restrict__ F, int64_t ld, int B, int lag_off,
                                                              int rows_per_block, double* __restrict__ part, unsigned* __restrict__ ticket,
                                                              double* __restrict__ out, FusedHead fused) {
    constexpr int W = 2 * D + 2 * D * D;
    __shared__ double red[4][W];
    __shared__ TicaWaveLds<D> s_head;
    __shared__ unsigned s_last;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    double acc[W];
#pragma unroll
    for (int o = 0; o < W; ++o) acc[o] = 0.0;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < B ? r0 + rows_per_block : B;
    for (int64_t r = r0 + t; r < r1; r += 256) {
        double a[D], b[D];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            a[i] = (double)F[r * ld + i];
            b[i] = (double)F[(r + lag_off) * ld + i];
        }
#pragma unroll
        for (int i = 0; i < D; ++i) {
            acc[i] += a[i];
            acc[D + i] += b[i];
#pragma unroll
            for (int j = 0; j < D; ++j) {
                acc[2 * D + i * D + j] += a[i] * a[j];
                acc[2 * D + D * D + i * D + j] += a[i] * b[j];
            }
        }
    }
    // wave reduction as a butterfly reduce-scatter: every step halves the values a lane carries (W = 40 -> 20 -> 10 -> 5,
    // then three all-reduce steps): ~50 float64 shuffles per lane in independent chains instead of 6 W = 240 dependent
    // ones (a float64 shuffle is two ds_bpermute round trips: the plain form spent 14 us of latency here)
    {
        int base = 0, dup = 0;
        const int cnt = butterfly_sum<W, 32, W, double>(acc, lane, base, dup);
        if ((lane & dup) == 0) {
#pragma unroll
            for (int i = 0; i < W; ++i)
                if (i < cnt) red[wave][base + i] = acc[i];
        }
    }
    __syncthreads();
    if (t < W) handoff_store(part + (int64_t)blockIdx.x * W + t, ((red[0][t] + red[1][t]) + red[2][t]) + red[3][t]);
    // the block that finishes last adds the partials up in block order (no second launch; same sums whichever
    // block it is).  Hand-off between the blocks: handoff.h (write-through partials, drained by every wave before the
    // barrier and the agent-scope ticket; the last arriver acquires at agent scope and reads with sc1 loads).
    if (handoff_arrive_last(ticket, gridDim.x, &s_last)) {
        // partials of the other blocks come from memory (1-2 us each): G thread groups take the blocks b = g, g + G, ...
        // with several loads in flight, then W threads add the G group sums in group order (fixed order: deterministic)
        constexpr int G = 256 / W;
        __shared__ double s_grp[G][W];
        const int g = t / W, o = t - g * W;
        if (g < G) {
            double s = 0.0;
            for (unsigned b0 = g; b0 < gridDim.x; b0 += 8 * G) {   // eight loads in flight, added in block order
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const unsigned b = b0 + (unsigned)u * G;
                    v[u] = handoff_load(part + (int64_t)(b < gridDim.x ? b : b0) * W + o);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (b0 + (unsigned)u * G < gridDim.x) s += v[u];
            }
            s_grp[g][o] = s;
        }
        __syncthreads();
        if (t < W) {
            double s = 0.0;
#pragma unroll
            for (int q = 0; q < G; ++q) s += s_grp[q][t];
            out[t] = s;
            red[0][t] = s;
        }
        if (fused.on) {
            __syncthreads();
            if (wave == 0)
                tica_grad_wave<D>(s_head, &red[0][0], fused.Bg, fused.reg, fused.gradp, fused.log, fused.log_count, fused.log_cap, fused.log_width, lane);
        }
    }
}
typedef void (*tica_stats_fn_t)(const float*, int64_t, int, int, int, double*, unsigned*, double*, FusedHead);
static tica_stats_fn_t tica_stats_rows_fn(int d) {
    switch (d) {
        case 1: return tica_stats_rows_kernel<1>;
        case 2: return tica_stats_rows_kernel<2>;
        case 3: return tica_stats_rows_kernel<3>;
        case 4: return tica_stats_rows_kernel<4>;
        default: return nullptr;
    }
}
// rows per block of the kernel above: at most 512 blocks, whole multiples of 256 rows
static int stats_rows_per_block(int64_t batch) { return (int)(cdiv(cdiv(batch, 256), 256) * 256); }   // <= 256 blocks: each ends on one ticket (~70 ns apiece, serialised)

// Gradient of the loss w.r.t. the network outputs.  Sample i (0 <= i < B) has f_t in row i and f_lag in
// row i + lag_off:  dL/df_t[i] = Gu u_i + Gv v_i + c,  dL/df_lag[i] = Gv u_i  (u = f_t - mu, v = f_lag - mu).
// Row j of dZ collects whatever lands on it: its own t-gradient (j < B) plus the lag-gradient of sample
// j - lag_off (j >= lag_off).  With lag_off = B the halves are disjoint; with lag_off = lag (contiguous
// batch, shared rows) an interior row receives both.  Multiplied by act'(F) of the last layer.
// Evaluated in float64 from the float64 batch statistics, rounded once: the loss does not change when a constant is
// added to the outputs, so the exact gradient rows sum to zero over the batch, and every parameter whose gradient is a
// multiple of that sum (the last bias; the bias of any hidden unit that stays on one side of its ReLU kink over the
// batch) has an exactly zero gradient.  Adam divides by |g| + 1e-8: a common-mode residue of 1e-5 -- what mu and the
// matrices rounded to float32 leave -- moves those parameters by a full +-lr per step, where autograd's
// (g - mean g) leaves 1e-10.  In float64 the rows sum to zero up to their own final rounding, as there.
__global__ __launch_bounds__(256) void tica_dF_kernel(const float* __restrict__ F, int64_t ldf, int B, int d, int lag_off,
                                                      const double* __restrict__ gradp, int act, float* __restrict__ dZ,
                                                      int64_t ldz, DropCfg drop, float hscale) {
    __shared__ double s_g[kMaxTicaDim * (2 * kMaxTicaDim + 2)];
    const int np = d + 2 * d * d + d;
    for (int i = threadIdx.x; i < np; i += 256) s_g[i] = gradp[i];
    __syncthreads();
    const double* mu = s_g;
    const double* Gu = s_g + d;
    const double* Gv = Gu + d * d;
    const double* cv = Gv + d * d;
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t rows = (int64_t)B + lag_off;
    if (j >= rows) return;
    const bool has_t = j < B;            // row j is the f_t row of sample j
    const bool has_l = j >= lag_off;     // row j is the f_lag row of sample j - lag_off
    double u[kMaxTicaDim], v[kMaxTicaDim], w[kMaxTicaDim];
    float fj[kMaxTicaDim];
    const float* frow = F + j * ldf;
    for (int i = 0; i < d; ++i) {
        fj[i] = frow[i];
        u[i] = (double)fj[i] - mu[i];                                               // u_j
        v[i] = has_t ? (double)F[(j + lag_off) * ldf + i] - mu[i] : 0.0;           // v_j
        w[i] = has_l ? (double)F[(j - lag_off) * ldf + i] - mu[i] : 0.0;           // u_{j - lag_off}
    }
    for (int i = 0; i < d; ++i) {
        double g = 0.0;
        if (has_t) {
            g = cv[i];
            for (int q = 0; q < d; ++q) {
                g = fma(Gu[i * d + q], u[q], g);
                g = fma(Gv[i * d + q], v[q], g);
            }
        }
        if (has_l) {
            double gl = 0.0;
            for (int q = 0; q < d; ++q) gl = fma(Gv[i * d + q], w[q], gl);
            g += gl;
        }
        float gf = (float)g;
        if (drop.thr != 0u) gf *= f4c(drop.mult(j, i & ~3), i & 3);   // F holds act(z) * keep / (1 - p)
        dZ[j * ldz + i] = gf * act_grad_from_out(act, fj[i] * hscale);
    }
}

// ------------------------------------------------------------------ fused backward of a narrow last layer
// Deep-TICA's last Linear maps K hidden units to D <= 8 outputs: as separate products its wgrad, dgrad and the
// two bias-gradient passes each stream the K-wide activations H (or write the K-wide dZ) for a handful of
// flops per byte.  One pass does all of it: per row r
//   g      = dL/dz_last[r]  (D values; the tica_dF formula above, evaluated in place)
//   dW    += g (x) H[r]          -> slab[block][D][K]          (wgrad partial of the last layer)
//   db    += g                   -> bpart_last[block][D]
//   dZ[r]  = (g W) * act'(H[r])  -> written once, 16-byte stores (dgrad of the last layer)
//   db'   += dZ[r]               -> bpart_prev[block][K]       (bias gradient of the layer before)
// HBM: K floats read + K floats written per row.  Thread (row group rl, 4 columns c4): D x 4 weights and
// D x 4 + 4 accumulators in registers; the 256 / (K/4) row groups of a block are combined through LDS in
// fixed order, blocks by reduce_grads_kernel in float64.
template <int D>
__global__ __launch_bounds__(256) void head_backward_kernel(const float* __restrict__ F, int64_t ldf, int B, int lag_off,
                                                            const double* __restrict__ gradp, int act_last,
                                                            const float* __restrict__ H, int64_t ldh, int K, int act_prev,
                                                            const float* __restrict__ W, int64_t rows_per_block,
                                                            float* __restrict__ dZ, int64_t ldz, float* __restrict__ slab,
                                                            float* __restrict__ bpart_last, float* __restrict__ bpart_prev,
                                                            DropCfg drop_prev, float hscale_prev) {
    constexpr int U = 4;                        // rows in flight per thread
    extern __shared__ __attribute__((aligned(16))) double s_memd[];
    double* s_g = s_memd;                       // mu | Gu | Gv | c   (float64: see tica_dF_kernel)
    float* s_gf = reinterpret_cast<float*>(s_memd + (2 * D + 2 * D * D));  // [groups][U][D] loss gradients of the rows in flight
    const int t = threadIdx.x;
    const int C4 = K / 4, groups = 256 / C4;    // a row group (C4 <= 64 lanes) lies inside one wave
    float* s_red = s_gf + groups * U * D;       // [groups][(D + 1) * K + D]
    for (int i = t; i < 2 * D + 2 * D * D; i += 256) s_g[i] = gradp[i];
    __syncthreads();
    const double* mu = s_g;
    const double* Gu = s_g + D;
    const double* Gv = Gu + D * D;
    const double* cv = Gv + D * D;
    const int c4 = t % C4, rl = t / C4;
    const int64_t rows = (int64_t)B + lag_off;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    float4 w[D], aw[D];
    float4 ab = make_float4(0.f, 0.f, 0.f, 0.f);
    float al[D];
#pragma unroll
    for (int j = 0; j < D; ++j) {
        w[j] = *reinterpret_cast<const float4*>(W + (int64_t)j * K + c4 * 4);
        aw[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        al[j] = 0.f;
    }
    float* gmine = s_gf + rl * U * D;
    for (int64_t rb = r0 + rl; rb < r1; rb += (int64_t)groups * U) {
        float4 h[U];
#pragma unroll
        for (int q = 0; q < U; ++q) {   // the K-wide loads first: U rows in flight
            const int64_t r = rb + (int64_t)q * groups;
            h[q] = r < r1 ? *reinterpret_cast<const float4*>(H + r * ldh + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        // dL/dz_last of the U rows: component i by lane i, i + C4, ... of the row group, shared through LDS.  The output
        // rows it needs (own row, the pair's lagged row, the row it is the lagged row of) are loaded up front, unconditionally
        // and for all U rows at once, from clamped row indices: inside the has_t / has_l branches they were two dependent
        // round trips per row (9.0 -> 8.3 us at 8202 rows; large batch 181 -> 185 M frames/s).
        float fro[U][D], fvo[U][D], fwo[U][D];
        if (c4 < D) {
            const bool fvec = D == 4 && (ldf & 3) == 0 && (reinterpret_cast<uintptr_t>(F) & 15) == 0;
#pragma unroll
            for (int q = 0; q < U; ++q) {
                const int64_t r = rb + (int64_t)q * groups;
                const int64_t rt = r < r1 ? r : r0;                          // a row of this block (not used when r >= r1)
                const int64_t rv = (r < r1 && r < B) ? r + lag_off : rt;     // < B + lag_off = rows
                const int64_t rw = (r < r1 && r >= lag_off) ? r - lag_off : rt;
                if constexpr (D == 4) {
                    if (fvec) {
                        const float4 x = *reinterpret_cast<const float4*>(F + rt * ldf);
                        const float4 y = *reinterpret_cast<const float4*>(F + rv * ldf);
                        const float4 z = *reinterpret_cast<const float4*>(F + rw * ldf);
                        fro[q][0] = x.x; fro[q][1] = x.y; fro[q][2] = x.z; fro[q][3] = x.w;
                        fvo[q][0] = y.x; fvo[q][1] = y.y; fvo[q][2] = y.z; fvo[q][3] = y.w;
                        fwo[q][0] = z.x; fwo[q][1] = z.y; fwo[q][2] = z.z; fwo[q][3] = z.w;
                        continue;
                    }
                }
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    fro[q][k] = F[rt * ldf + k];
                    fvo[q][k] = F[rv * ldf + k];
                    fwo[q][k] = F[rw * ldf + k];
                }
            }
        }
#pragma unroll
        for (int q = 0; q < U; ++q) {
            const int64_t r = rb + (int64_t)q * groups;
            for (int i = c4; i < D; i += C4) {
                float gi = 0.f;
                if (r < r1) {
                    const bool has_t = r < B, has_l = r >= lag_off;
                    double gd = 0.0;
                    if (has_t) {
                        gd = cv[i];
#pragma unroll
                        for (int k = 0; k < D; ++k) {
                            gd = fma(Gu[i * D + k], (double)fro[q][k] - mu[k], gd);
                            gd = fma(Gv[i * D + k], (double)fvo[q][k] - mu[k], gd);
                        }
                    }
                    if (has_l) {
                        double gl = 0.0;
#pragma unroll
                        for (int k = 0; k < D; ++k) gl = fma(Gv[i * D + k], (double)fwo[q][k] - mu[k], gl);
                        gd += gl;
                    }
                    float fri = fro[q][0];   // fro[q][i] without a dynamically indexed register array
#pragma unroll
                    for (int k = 1; k < D; ++k) fri = i == k ? fro[q][k] : fri;
                    gi = (float)gd * act_grad_from_out(act_last, fri);
                }
                gmine[q * D + i] = gi;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int q = 0; q < U; ++q) {
            const int64_t r = rb + (int64_t)q * groups;
            if (r >= r1) break;
            float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const float g = gmine[q * D + j];
                z.x = fmaf(g, w[j].x, z.x); z.y = fmaf(g, w[j].y, z.y); z.z = fmaf(g, w[j].z, z.z); z.w = fmaf(g, w[j].w, z.w);
                aw[j].x = fmaf(g, h[q].x, aw[j].x); aw[j].y = fmaf(g, h[q].y, aw[j].y);
                aw[j].z = fmaf(g, h[q].z, aw[j].z); aw[j].w = fmaf(g, h[q].w, aw[j].w);
                if (c4 == 0) al[j] += g;
            }
            z.x *= act_grad_from_out(act_prev, h[q].x * hscale_prev); z.y *= act_grad_from_out(act_prev, h[q].y * hscale_prev);
            z.z *= act_grad_from_out(act_prev, h[q].z * hscale_prev); z.w *= act_grad_from_out(act_prev, h[q].w * hscale_prev);
            if (drop_prev.thr != 0u) {   // H holds act(z) * keep / (1 - p): the same mask scales the gradient
                const float4 k = drop_prev.mult(r, c4 * 4);
                z.x *= k.x; z.y *= k.y; z.z *= k.z; z.w *= k.w;
            }
            handoff_store16(dZ + r * ldz + c4 * 4, hv4f{z.x, z.y, z.z, z.w});   // write-through: nothing to write back when the launch ends
            ab.x += z.x; ab.y += z.y; ab.z += z.z; ab.w += z.w;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();   // gmine is rewritten by the next iteration
    }
    // combine the row groups: s_red[rl] = [dW (D x K) | db' (K) | db (D)]
    const int stride = (D + 1) * K + D;
    float* mine = s_red + (int64_t)rl * stride;
#pragma unroll
    for (int j = 0; j < D; ++j) *reinterpret_cast<float4*>(mine + j * K + c4 * 4) = aw[j];
    *reinterpret_cast<float4*>(mine + D * K + c4 * 4) = ab;
    if (c4 == 0) {
#pragma unroll
        for (int j = 0; j < D; ++j) mine[(D + 1) * K + j] = al[j];
    }
    __syncthreads();
    for (int i = t; i < stride; i += 256) {
        float tot = 0.f;
        for (int q = 0; q < groups; ++q) tot += s_red[(int64_t)q * stride + i];
        if (i < D * K) slab[(int64_t)blockIdx.x * D * K + i] = tot;
        else if (i < (D + 1) * K) bpart_prev[(int64_t)blockIdx.x * K + (i - D * K)] = tot;
        else bpart_last[(int64_t)blockIdx.x * D + (i - (D + 1) * K)] = tot;
    }
}

typedef void (*head_backward_fn_t)(const float*, int64_t, int, int, const double*, int, const float*, int64_t, int, int, const float*, int64_t,
                                   float*, int64_t, float*, float*, float*, DropCfg, float);
static head_backward_fn_t head_backward_fn(int d) {
    switch (d) {
        case 1: return head_backward_kernel<1>;
        case 2: return head_backward_kernel<2>;
        case 3: return head_backward_kernel<3>;
        case 4: return head_backward_kernel<4>;
        case 5: return head_backward_kernel<5>;
        case 6: return head_backward_kernel<6>;
        case 7: return head_backward_kernel<7>;
        case 8: return head_backward_kernel<8>;
        default: return nullptr;
    }
}

// ------------------------------------------------------------------ autoencoder loss
// SSE = sum ((y - xn) * range)^2 over rows x F ; part[block]
// `ticket` != null: the last block to finish adds the partials up in block order (the sum sum_partials_kernel would
// produce) into out[0] and, when `log` != null, appends the step's loss record (ae_log_kernel) -- the one-GPU step
// then needs neither of those two launches.
__global__ __launch_bounds__(256) void ae_sse_kernel(const float* __restrict__ Y, int64_t ldy, const float* __restrict__ Xn,
                                                     int64_t ldx, RowMap rows, int64_t R, int F,
                                                     const float* __restrict__ range, double* __restrict__ part,
                                                     unsigned* __restrict__ ticket, double* __restrict__ out, double Bg,
                                                     double* __restrict__ log, int* __restrict__ log_count, int log_cap,
                                                     int log_width, int rows_per_block) {
    __shared__ double red[256];
    const int t = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < R ? r0 + rows_per_block : R;
    // the block's rows x F elements flat over the threads, eight independent element loads in flight per thread
    // (a thread that walked its rows one after the other spent the kernel waiting: 16 dependent round trips, 27 us)
    double s = 0.0;
    const int per_block = (int)(r1 - r0) * F;
    for (int e0 = t; e0 < per_block; e0 += 8 * 256) {
        float ev[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + 256 * u;
            ev[u] = 0.f;
            if (e < per_block) {
                const int rr = e / F, c = e - rr * F;
                const int64_t r = r0 + rr;
                ev[u] = (Y[r * ldy + c] - Xn[rows.template get<true>(r) * ldx + c]) * range[c];
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) s += (double)ev[u] * (double)ev[u];
    }
    red[t] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (t < off) red[t] += red[t + off];
        __syncthreads();
    }
    if (ticket == nullptr) {
        if (t == 0) part[blockIdx.x] = red[0];
        return;
    }
    if (t == 0) handoff_store(part + blockIdx.x, red[0]);
    __shared__ unsigned is_last;
    if (!handoff_arrive_last(ticket, gridDim.x, &is_last)) return;   // handoff.h; see tica_stats_rows_kernel
    if (t < 64) {   // one wave, the arithmetic of sum_partials_kernel: lanes over the blocks, shuffle tree
        double tot = 0.0;
        for (int b = t; b < (int)gridDim.x; b += 64) tot += handoff_load(part + b);
        for (int off = 32; off > 0; off >>= 1) tot += __shfl_down(tot, off, 64);
        if (t == 0) {
            out[0] = tot;
            if (log != nullptr) {
                const int slot = *log_count;
                if (slot < log_cap) {
                    log[(int64_t)slot * log_width + 0] = tot / (Bg * (double)F);
                    log[(int64_t)slot * log_width + 1] = Bg;
                }
                *log_count = slot + 1;
            }
        }
    }
}

// dY = scale * (y - xn) * range^2 * act'(y)
__global__ __launch_bounds__(256) void ae_dY_kernel(const float* __restrict__ Y, int64_t ldy, const float* __restrict__ Xn,
                                                    int64_t ldx, RowMap rows, int64_t R, int F,
                                                    const float* __restrict__ range, float scale, int act,
                                                    float* __restrict__ dZ, int64_t ldz, DropCfg drop, float hscale) {
    const int64_t total = R * F;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / F;
        const int c = (int)(i - r * F);
        const float y = Y[r * ldy + c];
        const float x = Xn[rows.template get<true>(r) * ldx + c];
        const float rg = range[c];
        float g = scale * (y - x) * rg * rg * act_grad_from_out(act, y * hscale);
        if (drop.thr != 0u) g *= f4c(drop.mult(r, c & ~3), c & 3);
        dZ[r * ldz + c] = g;
    }
}

// test hook: the keep / (1 - p) multipliers of a layer's dropout for rows [0, rows)
__global__ __launch_bounds__(256) void dropout_mask_kernel(float* __restrict__ out, int64_t rows, int width, DropCfg drop) {
    const int q4 = (width + 3) / 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < rows * q4; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / q4;
        const int c = (int)(i - r * q4) * 4;
        const float4 k = drop.thr != 0u ? drop.mult(r, c) : make_float4(1.f, 1.f, 1.f, 1.f);
        float* o = out + r * width + c;
        o[0] = k.x;
        if (c + 1 < width) o[1] = k.y;
        if (c + 2 < width) o[2] = k.z;
        if (c + 3 < width) o[3] = k.w;
    }
}

__global__ void ae_log_kernel(const double* __restrict__ stats, double Bg, int F, double* __restrict__ log,
                              int* __restrict__ log_count, int log_cap, int log_width) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int slot = *log_count;
    if (slot < log_cap) {
        log[(int64_t)slot * log_width + 0] = stats[0] / (Bg * (double)F);
        log[(int64_t)slot * log_width + 1] = Bg;
    }
    *log_count = slot + 1;
}

__global__ void fill_kernel(float* p, int64_t n, float v) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}

// kind: 0 forward, 1 weight gradient, 2 input gradient (-1: any).  Every class (layer, kind) counts its own samples: the
// host may sample the kinds on different steps (dcv_mlp_profile_pause: a profiled launch costs ~7 us of command-processor
// work, so bench.py staggers them instead of stamping both layer-0 products of every step)
static inline bool prof_on(const dcv_mlp* m, int layer, int kind = -1) {
    if (!(m->prof_level > 0 && !m->prof_paused && (m->prof_level > 1 || layer == 0))) return false;
    if (kind < 0) return true;
    return ((m->prof_kind_off >> kind) & 1) == 0 && m->prof_cnt[(size_t)3 * layer + kind] < m->prof_cap;
}
// which = 0: before the launch(es) of the class, 1: after.  The pair of events is offered to the block engine's launcher
// (g_launch_ev, common.h), which stamps it with the kernel's own begin / end; when the launch in between did not take it
// (the fused small-network step, a grouped launch), the events are recorded around the launch instead.
static inline void prof_mark(dcv_mlp* m, int layer, int kind, int which, hipStream_t s) {
    if (!prof_on(m, layer, kind)) return;
    const size_t cls = (size_t)3 * layer + kind;
    hipEvent_t* ev = &m->prof_ev[(cls * m->prof_cap + m->prof_cnt[cls]) * 2];
    if (which == 0) {
        (void)hipEventRecord(ev[0], s);
        if (g_launch_ev.start == nullptr) g_launch_ev = LaunchEvents{ev[0], ev[1]};   // (one offer at a time: a grouped launch is bracketed by two classes)
    } else {
        m->prof_cnt[cls] += 1;
        if (g_launch_taken == ev[0]) {   // the launcher took the pair: both events carry the kernel's own times
            g_launch_taken = nullptr;
            return;
        }
        if (g_launch_ev.start == ev[0]) g_launch_ev = LaunchEvents{};
        (void)hipEventRecord(ev[1], s);
    }
}

// Deep-TICA batches.  Gathered batch (idx given): rows [0,B) are the x_t rows, rows [B,2B) the x_lag rows.
// Contiguous batch (row0 .. row0+B-1, the sequential-split / unshuffled case): x_lag of sample i IS x_t of
// sample i + lag, so the network is evaluated once on the B + lag rows row0 .. row0+B+lag-1 and both
// halves read the shared outputs -- the same numbers as two separate passes (every row goes through the
// same weights), about half the matrix work.  The gradient of a shared row is the sum of its two roles.
// (Not with dropout in a training step: the reference evaluates x_t and x_lag in two forward calls with independent masks.)
static bool shared_rows(const dcv_mlp* m, const int64_t* idx, int batch) {
    return m->desc.model == DCV_MODEL_DEEPTICA && idx == nullptr && m->desc.lag >= 1 && m->desc.lag <= batch && !m->no_row_sharing &&
           !(m->fwd_train && (m->any_drop || m->any_bn));   // separate forward calls: independent dropout masks, separate batch statistics
}
// dropout behind Linear `layer` in the current step (off in evaluation mode)
static DropCfg drop_cfg(const dcv_mlp* m, int layer) {
    const float p = m->desc.dropout[layer];
    if (!m->fwd_train || !(p > 0.f)) return kNoDrop;
    double t = (double)p * 4294967296.0;
    if (t > 4294967295.0) t = 4294967295.0;
    if (t < 1.0) t = 1.0;
    // rank r of a data-parallel run draws from its own stream (key word 1 offset by r * golden ratio): every rank holds
    // the same seed, and with a shared key all ranks would mask their local rows alike
    return DropCfg{(uint32_t)t, 1.f / (1.f - p), (uint32_t)(m->desc.seed & 0xFFFFFFFFull), (uint32_t)(m->desc.seed >> 32) + 0x9E3779B9u * m->drop_rank, (uint32_t)layer,
                   (uint32_t)m->cur_step};
}
static float drop_hscale(const dcv_mlp* m, int layer) {
    const float p = m->desc.dropout[layer];
    return (m->fwd_train && p > 0.f) ? 1.f - p : 1.f;
}
static RowMap batch_rows(const dcv_mlp* m, const int64_t* idx, int64_t row0, int batch) {
    if (m->desc.model == DCV_MODEL_DEEPTICA && !shared_rows(m, idx, batch)) return RowMap{idx, row0, batch, m->desc.lag};
    return RowMap{idx, row0, 0, 0};
}
static int64_t rows_of(const dcv_mlp* m, const int64_t* idx, int batch) {
    if (m->desc.model != DCV_MODEL_DEEPTICA) return batch;
    return shared_rows(m, idx, batch) ? (int64_t)batch + m->desc.lag : 2 * (int64_t)batch;
}
static int lag_offset(const dcv_mlp* m, const int64_t* idx, int batch) { return shared_rows(m, idx, batch) ? m->desc.lag : batch; }

// wgrad split plan: enough workgroups to fill the chip twice, chunks a multiple of 32 rows
static void wgrad_plan(int out, int in, int64_t rows, int64_t* k_chunk, int64_t* splits) {
    const int64_t tiles = (out <= 32 ? 1 : cdiv(out, 128)) * (in <= 32 ? 1 : cdiv(in, 128));
    int64_t want = cdiv(2 * (int64_t)num_cus(), tiles);
    int64_t max_by_rows = cdiv(rows, 256);
    // Small batches.  When 64 x 64 tiles reach two workgroups per CU with a split count the rows allow, take them
    // (pick_cfg<kTN> follows: fewer than one 128 x 128 workgroup per CU): measured on the 256 x 512 x 8202 product, 16 chunks
    // of 64 x 64 tiles 21.9 us against 29 chunks of 128 x 128 tiles 23.7 -- and 16 slabs instead of 29 to write and reduce; a
    // multiple of 8 chunks lets the XCD map keep a chunk's rows in one L2.  Round 4 tried the same plan where the rows would
    // allow the 128 x 128 one but the chunks stay short (<= 1024 rows: the 16 384 rows of a gathered 8192-pair batch give 64
    // slabs of 512 KB per step -- 32 MB written and re-read by the reduction, 11.4 us -- where 16 chunks of 64 x 64 tiles leave
    // 8 MB): the reduction fell to 7.0 us, but the GATHERED weight gradient went from 38.7 to 60.0 us (per-thread 64-bit row
    // pointers: four times the stage loads per flop of the 128 x 128 tile) -- kept behind DCV_WGRAD_Q=1, off by default.
    static const bool q_wide = [] { const char* e = getenv("DCV_WGRAD_Q"); return e && e[0] == '1'; }();   // opt-in: see below
    const int64_t tiles_q = cdiv(out, 64) * cdiv(in, 64);
    const int64_t want_q = cdiv(cdiv(2 * (int64_t)num_cus(), tiles_q), 8) * 8;
    if (out > 32 && in > 32 && want_q <= max_by_rows) {
        int64_t kc = cdiv(cdiv(rows, want_q), 32) * 32;
        if (cdiv(rows, kc) % 8 != 0) kc = cdiv(rows, want_q);   // ragged chunk ends (the stage tail goes through registers)
        if (tiles * cdiv(rows, kc) <= (int64_t)num_cus() / 2 &&   // pick_cfg<kTN>'s condition for 64 x 64 tiles
            (want > max_by_rows || (q_wide && kc <= 1024))) {
            *k_chunk = kc;
            *splits = cdiv(rows, kc);
            return;
        }
    }
    if (want > max_by_rows) {
        // row-limited: a split count that is a multiple of 8 keeps tiles x splits on whole multiples of the CU count (33
        // splits x 8 tiles = 264 workgroups cost 1.6 x of 256: the 8 extra share SIMDs with 8 others)
        want = max_by_rows >= 8 ? max_by_rows / 8 * 8 : max_by_rows;
    }
    if (want < 1) want = 1;
    int64_t kc = cdiv(cdiv(rows, want), 32) * 32;
    *k_chunk = kc;
    *splits = cdiv(rows, kc);
}

// the fused backward of the last layer applies to a Deep-TICA network whose last Linear is narrow (<= 8 outputs)
// and whose input width tiles a 256-thread block in 16-byte segments
static size_t head_lds_bytes(int D, int K) {
    const int groups = 256 / (K / 4);
    return (size_t)(2 * D + 2 * D * D) * sizeof(double) + ((size_t)groups * 4 * D + (size_t)groups * ((size_t)(D + 1) * K + D)) * sizeof(float);
}
static bool head_fusable(const dcv_mlp* m) {
    static const bool off = [] { const char* e = getenv("DCV_NO_HEAD_FUSION"); return e && e[0] == '1'; }();
    if (off || m->L < 2 || m->any_bn) return false;
    if (m->desc.dropout[m->L - 1] > 0.f) return false;   // dropout on the network output: general kernels
    const LayerPlan& p = m->layers[m->L - 1];
    const LayerPlan& q = m->layers[m->L - 2];
    const int K = p.in;
    if (p.out > 8 || K % 4 != 0 || K / 4 > 256 || 256 % (K / 4) != 0) return false;
    if (!quad_ok(q.H, q.ldh) || !quad_ok(m->dZ[0], m->ld_dz) || !quad_ok(m->dZ[1], m->ld_dz) || !quad_ok(m->params + p.w_off, K)) return false;
    if (K / 4 > 64) return false;   // a row group must lie inside one wave
    return head_lds_bytes(p.out, K) <= 60 * 1024;
}
// blocks of the fused pass: about four per CU, at least 32 rows each, bounded by the partial buffers
static void head_plan(const dcv_mlp* m, int64_t R, int64_t* rows_per_block, int64_t* blocks) {
    const LayerPlan& p = m->layers[m->L - 1];
    int64_t want = 4 * (int64_t)num_cus();
    if (want > p.max_splits) want = p.max_splits;
    int64_t rpb = cdiv(cdiv(R, want), 32) * 32;
    if (rpb < 32) rpb = 32;
    *rows_per_block = rpb;
    *blocks = cdiv(R, rpb);
}

static void mlp_free(dcv_mlp* m) {
    if (!m) return;
    snet_free(m);
    snet_dt_free(m);
    snet_image_free(m);
    auto f = [](void* p) { if (p) (void)hipFree(p); };
    f(m->params); f(m->grads); f(m->adam_m); f(m->adam_v); f(m->opt_aux); f(m->dZ[0]); f(m->dZ[1]); f(m->stats); f(m->gradp);
    f(m->spart); f(m->log); f(m->log_count); f(m->ticket); f(m->feat_range); f(m->ident); f(m->zeros_d); f(m->ones_d); f(m->proj_ws);
    for (auto& l : m->layers) { f(l.H); f(l.slab); f(l.bpart); f(l.mask); f(l.Y); f(l.rm); f(l.rv); f(l.bn_stat); f(l.bn_part); f(l.bn_gpart); f(l.bn_bpart); }
    g_launch_ev = LaunchEvents{};   // no stale offer of events that are about to be destroyed
    g_launch_taken = nullptr;
    for (hipEvent_t e : m->prof_ev) (void)hipEventDestroy(e);
    f(m->tail.ws); f(m->tail.cnt);
    for (int i = 0; i < 4; ++i) if (m->gexec[i]) (void)hipGraphExecDestroy(m->gexec[i]);
    delete m;
}

template <class T>
static int dmalloc(T** p, size_t count) {
    *p = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(p), (count ? count : 1) * sizeof(T));
    if (e != hipSuccess) {
        set_error("hipMalloc of %zu bytes failed: %s", count * sizeof(T), hipGetErrorString(e));
        return DCV_ENOMEM;
    }
    return DCV_OK;
}

}  // namespace dcv

static int reset_bn_state(dcv_mlp* m, hipStream_t s);
// optimiser state as a freshly constructed torch optimiser holds it
static int reset_opt_state(dcv_mlp* m, hipStream_t s) {
    const size_t bytes = m->n_params * sizeof(float);
    DCV_CHECK_HIP(hipMemsetAsync(m->adam_m, 0, bytes, s));
    DCV_CHECK_HIP(hipMemsetAsync(m->adam_v, 0, bytes, s));
    if (m->opt_aux) DCV_CHECK_HIP(hipMemsetAsync(m->opt_aux, 0, bytes, s));
    if (m->desc.optimizer == DCV_OPT_ADAGRAD && m->desc.initial_accumulator_value != 0.0) {
        hipLaunchKernelGGL(fill_kernel, dim3(256), dim3(256), 0, s, m->adam_v, m->n_params, (float)m->desc.initial_accumulator_value);
        DCV_CHECK_LAUNCH();
    }
    // Rprop's step_size and ASGD's eta are created by torch inside the FIRST optimizer.step(), from the learning rate the
    // group holds at that moment -- after a scheduler's constructor (OneCycleLR, LinearLR, a warm-up LambdaLR) has already
    // rescaled it: seeded in first_step_state(), right before the first update, from m->lr
    m->nadam_mu_product = 1.0;
    m->asgd_eta = (double)(float)m->lr;
    m->adam_t = 0;
    m->drop_step = 0;
    return DCV_OK;
}

static bool graph_enabled() {
    static const bool on = [] { const char* e = getenv("DCV_GRAPH"); return e && e[0] == '1'; }();
    return on;
}
// =================================================================== C-ABI
extern "C" int dcv_mlp_create(const dcv_mlp_desc* desc, dcv_mlp** out) {
    DCV_REQUIRE(desc && out, "dcv_mlp_create: null argument");
    *out = nullptr;
    const int L = desc->n_layers;
    DCV_REQUIRE(desc->model == DCV_MODEL_DEEPTICA || desc->model == DCV_MODEL_AE, "dcv_mlp_create: unknown model %d", desc->model);
    DCV_REQUIRE(L >= 1 && L <= DCV_MAX_LAYERS, "dcv_mlp_create: n_layers=%d out of range", L);
    for (int l = 0; l <= L; ++l) DCV_REQUIRE(desc->dims[l] >= 1, "dcv_mlp_create: dims[%d]=%d", l, desc->dims[l]);
    for (int l = 0; l < L; ++l)
        DCV_REQUIRE(desc->act[l] >= DCV_ACT_NONE && desc->act[l] <= DCV_ACT_CUSTOM_SIGMOID, "dcv_mlp_create: act[%d]=%d unsupported", l, desc->act[l]);
    for (int l = 0; l < L; ++l)
        DCV_REQUIRE(desc->dropout[l] >= 0.f && desc->dropout[l] < 1.f, "dcv_mlp_create: dropout[%d]=%g outside [0, 1)", l, (double)desc->dropout[l]);
    DCV_REQUIRE(desc->optimizer >= DCV_OPT_ADAM && desc->optimizer <= DCV_OPT_RPROP, "dcv_mlp_create: optimizer %d unknown", desc->optimizer);
    DCV_REQUIRE(desc->max_batch >= 1, "dcv_mlp_create: max_batch=%d", desc->max_batch);
    if (desc->model == DCV_MODEL_DEEPTICA) {
        DCV_REQUIRE(desc->dims[L] <= kMaxTicaDim, "dcv_mlp_create: Deep-TICA output dimension %d > %d", desc->dims[L], kMaxTicaDim);
        DCV_REQUIRE(desc->lag >= 0, "dcv_mlp_create: lag=%d", desc->lag);
    } else {
        DCV_REQUIRE(desc->dims[L] == desc->dims[0], "dcv_mlp_create: autoencoder must map F=%d back to F (got %d)", desc->dims[0], desc->dims[L]);
        DCV_REQUIRE(desc->latent_layer >= 1 && desc->latent_layer < L, "dcv_mlp_create: latent_layer=%d", desc->latent_layer);
        DCV_REQUIRE(desc->dims[desc->latent_layer] <= 16, "dcv_mlp_create: latent dimension %d > 16", desc->dims[desc->latent_layer]);
    }
    dcv_mlp* m = new (std::nothrow) dcv_mlp();
    DCV_REQUIRE(m, "dcv_mlp_create: out of host memory");
    m->desc = *desc;
    m->L = L;
    m->d_out = desc->dims[L];
    m->no_row_sharing = 0;
    m->rows_cap = m->desc.model == DCV_MODEL_DEEPTICA ? 2 * (int64_t)desc->max_batch : desc->max_batch;
    m->lr = desc->lr;
    m->adam_t = 0;
    m->last_batch = 0;
    m->opt_aux = nullptr;
    m->momentum_rt = (desc->optimizer == DCV_OPT_ADAM || desc->optimizer == DCV_OPT_ADAMW || desc->optimizer == DCV_OPT_ADAMAX ||
                      desc->optimizer == DCV_OPT_NADAM || desc->optimizer == DCV_OPT_RADAM) ? desc->beta1 : desc->momentum;
    m->any_drop = false;
    m->any_bn = false;
    for (int l = 0; l < L; ++l) m->any_drop = m->any_drop || desc->dropout[l] > 0.f;
    m->fwd_train = false;
    m->head_done = false;
    m->upper_cb = nullptr;
    m->upper_cb_user = nullptr;
    m->tail = dcv::TailWs{};
    m->drop_step = 0;
    m->cur_step = 0;
    m->drop_rank = 0;
    m->snet = nullptr;
    m->snet_tried = false;
    m->snet_dt = nullptr;
    m->snet_dt_tried = false;
    m->snet_img = nullptr;
    m->snet_img_idx = nullptr;
    m->snet_img_floats = 0;
    m->snet_fwd_valid = false;
    m->last_path = 0;
    m->prof_level = m->prof_cap = 0;
    m->prof_kind_off = 0;
    for (int i = 0; i < 4; ++i) { m->gexec[i] = nullptr; m->gwarm[i] = false; }
    m->graph_on = graph_enabled();
    m->graph_off = false;
    m->graph_launches = 0;
    m->prof_paused = false;
    m->layers.resize(L);
    int64_t off = 0;
    int maxdim = 0;
    int rc = DCV_OK;
    for (int l = 0; l < L && rc == DCV_OK; ++l) {
        LayerPlan& p = m->layers[l];
        p.in = desc->dims[l];
        p.out = desc->dims[l + 1];
        p.act = desc->act[l];
        p.w_off = off;
        off += align_up((size_t)p.in * p.out, 4);
        p.b_off = off;
        off += align_up((size_t)p.out, 4);
        p.bn = desc->batchnorm[l] ? 1 : 0;
        p.g_off = p.be_off = -1;
        p.Y = nullptr; p.rm = p.rv = nullptr; p.bn_stat = p.bn_part = nullptr; p.bn_gpart = p.bn_bpart = nullptr;
        p.bn_batches = 0;
        if (p.bn) {
            p.g_off = off;
            off += align_up((size_t)p.out, 4);
            p.be_off = off;
            off += align_up((size_t)p.out, 4);
            m->any_bn = true;
        }
        p.ldh = align_up((size_t)p.out, 4);
        if (p.out > maxdim) maxdim = p.out;
        wgrad_plan(p.out, p.in, m->rows_cap, &p.k_chunk_cap, &p.max_splits);
        // a smaller batch may use smaller chunks; bound the splits by the 256-row floor
        p.max_splits = cdiv(m->rows_cap, 256) > p.max_splits ? p.max_splits : cdiv(m->rows_cap, 256);
        if (p.max_splits < 1) p.max_splits = 1;
        if (l == L - 1 && p.out <= 8) {   // the fused backward of a narrow last layer writes one partial per block (head_plan)
            int64_t hb = 4 * (int64_t)num_cus();
            if (hb > cdiv(m->rows_cap, 32)) hb = cdiv(m->rows_cap, 32);
            if (hb > p.max_splits) p.max_splits = hb;
        }
        rc = dmalloc(&p.H, (size_t)m->rows_cap * p.ldh);
        if (rc == DCV_OK) rc = dmalloc(&p.slab, (size_t)p.max_splits * p.in * p.out);
        if (rc == DCV_OK) rc = dmalloc(&p.bpart, (size_t)cdiv(m->rows_cap, 32) * p.out);  // row tiles of the dgrad epilogue can be as short as 32
        if (rc == DCV_OK && p.bn) {
            const size_t nblk = (size_t)cdiv(m->rows_cap, 256) + 2;
            rc = dmalloc(&p.Y, (size_t)m->rows_cap * p.ldh);
            if (rc == DCV_OK) rc = dmalloc(&p.rm, (size_t)p.out);
            if (rc == DCV_OK) rc = dmalloc(&p.rv, (size_t)p.out);
            if (rc == DCV_OK) rc = dmalloc(&p.bn_stat, (size_t)4 * p.out);
            if (rc == DCV_OK) rc = dmalloc(&p.bn_part, nblk * 2 * p.out);
            if (rc == DCV_OK) rc = dmalloc(&p.bn_gpart, nblk * p.out);
            if (rc == DCV_OK) rc = dmalloc(&p.bn_bpart, nblk * p.out);
        }
        p.mask = nullptr;
        p.mask_rows = -1;
        if (rc == DCV_OK && l + 1 < L && (p.act == DCV_ACT_RELU || p.act == DCV_ACT_LEAKY_RELU))   // one bit per element, whole tiles
            rc = dmalloc(&p.mask, (size_t)((m->rows_cap + 128) * (int64_t)(p.out + 128)) / 64 + 64);
    }
    m->n_params = off;
    m->ld_dz = align_up((size_t)maxdim, 4);
    const int d = m->d_out;
    m->stats_len = desc->model == DCV_MODEL_DEEPTICA ? 2 * d + 2 * d * d : 1;
    m->log_width = desc->model == DCV_MODEL_DEEPTICA ? 2 + 2 * d * d + d : 2;
    m->spart_blocks = desc->model == DCV_MODEL_DEEPTICA ? (int)cdiv(desc->max_batch, kStatBlockRows) : (int)cdiv(m->rows_cap, kSseRows);
    const int dl = desc->model == DCV_MODEL_AE ? desc->dims[desc->latent_layer] : d;
    if (rc == DCV_OK) rc = dmalloc(&m->params, (size_t)m->n_params);
    if (rc == DCV_OK) rc = dmalloc(&m->grads, (size_t)m->n_params);
    if (rc == DCV_OK) rc = dmalloc(&m->adam_m, (size_t)m->n_params);
    if (rc == DCV_OK) rc = dmalloc(&m->adam_v, (size_t)m->n_params);
    if (rc == DCV_OK && (((desc->optimizer == DCV_OPT_ADAM || desc->optimizer == DCV_OPT_ADAMW) && desc->amsgrad) ||
                         (desc->optimizer == DCV_OPT_RMSPROP && desc->centered)))
        rc = dmalloc(&m->opt_aux, (size_t)m->n_params);
    if (rc == DCV_OK) rc = dmalloc(&m->dZ[0], (size_t)m->rows_cap * m->ld_dz);
    if (rc == DCV_OK) rc = dmalloc(&m->dZ[1], (size_t)m->rows_cap * m->ld_dz);
    if (rc == DCV_OK) rc = dmalloc(&m->stats, (size_t)m->stats_len);
    if (rc == DCV_OK) rc = dmalloc(&m->gradp, (size_t)(2 * kMaxTicaDim + 2 * kMaxTicaDim * kMaxTicaDim));
    if (rc == DCV_OK) rc = dmalloc(&m->spart, (size_t)m->spart_blocks * m->stats_len);   // ticketed partials: handoff.h
    if (rc == DCV_OK) rc = dmalloc(&m->log_count, 1);
    if (rc == DCV_OK) rc = dmalloc(&m->ticket, 4);
    if (rc == DCV_OK) rc = dmalloc(&m->feat_range, (size_t)desc->dims[0]);
    if (rc == DCV_OK) rc = dmalloc(&m->ident, (size_t)dl * dl);
    if (rc == DCV_OK) rc = dmalloc(&m->zeros_d, (size_t)dl);
    if (rc == DCV_OK) rc = dmalloc(&m->ones_d, (size_t)dl);
    m->proj_ws_bytes = dcv_project_linear_workspace(m->rows_cap, dl, dl);
    if (rc == DCV_OK) rc = dmalloc(reinterpret_cast<char**>(&m->proj_ws), m->proj_ws_bytes);
    m->log = nullptr;
    m->log_cap = 0;
    if (rc != DCV_OK) {
        mlp_free(m);
        return rc;
    }
    hipError_t e = hipMemset(m->params, 0, m->n_params * sizeof(float));
    if (e == hipSuccess) e = hipMemset(m->grads, 0, m->n_params * sizeof(float));
    if (e == hipSuccess && reset_opt_state(m, nullptr) != DCV_OK) e = hipErrorUnknown;
    if (e == hipSuccess) e = hipMemset(m->log_count, 0, sizeof(int));
    if (e == hipSuccess) e = hipMemset(m->ticket, 0, 4 * sizeof(unsigned));
    if (e == hipSuccess) e = hipMemset(m->zeros_d, 0, dl * sizeof(float));
    if (e == hipSuccess) e = hipMemset(m->ident, 0, (size_t)dl * dl * sizeof(float));
    if (e == hipSuccess) {
        hipLaunchKernelGGL(fill_kernel, dim3(1), dim3(64), 0, 0, m->ones_d, (int64_t)dl, 1.f);
        hipLaunchKernelGGL(fill_kernel, dim3(4), dim3(256), 0, 0, m->feat_range, (int64_t)desc->dims[0], 1.f);
        std::vector<float> eye((size_t)dl * dl, 0.f);
        for (int i = 0; i < dl; ++i) eye[(size_t)i * dl + i] = 1.f;
        e = hipMemcpy(m->ident, eye.data(), eye.size() * sizeof(float), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) {
        for (auto& p : m->layers)
            if (p.bn) hipLaunchKernelGGL(fill_kernel, dim3(1), dim3(256), 0, 0, m->params + p.g_off, (int64_t)p.out, 1.f);   // BatchNorm1d: weight 1, bias 0
        if (reset_bn_state(m, nullptr) != DCV_OK) e = hipErrorUnknown;
    }
    if (e == hipSuccess) (void)alloc_tail_ws(&m->tail, 8);   // up to 8 column tiles; on failure the tail cut stays off
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        set_error("dcv_mlp_create: initialisation failed: %s", hipGetErrorString(e));
        mlp_free(m);
        return DCV_EHIP;
    }
    *out = m;
    return DCV_OK;
}

extern "C" void dcv_mlp_destroy(dcv_mlp* m) { mlp_free(m); }
extern "C" int64_t dcv_mlp_num_params(const dcv_mlp* m) { return m ? m->n_params : 0; }
extern "C" int64_t dcv_mlp_param_offset(const dcv_mlp* m, int32_t layer, int32_t which) {
    if (!m || layer < 0 || layer >= m->L) return -1;
    if (which == 2) return m->layers[layer].g_off;    // weight / bias of the batch normalisation behind the layer, -1 without one
    if (which == 3) return m->layers[layer].be_off;
    return which == 0 ? m->layers[layer].w_off : m->layers[layer].b_off;
}
extern "C" float* dcv_mlp_params(dcv_mlp* m) { return m ? m->params : nullptr; }
extern "C" float* dcv_mlp_grads(dcv_mlp* m) { return m ? m->grads : nullptr; }
extern "C" double* dcv_mlp_stats(dcv_mlp* m) { return m ? m->stats : nullptr; }
extern "C" int32_t dcv_mlp_stats_len(const dcv_mlp* m) { return m ? m->stats_len : 0; }
extern "C" int32_t dcv_mlp_log_width(const dcv_mlp* m) { return m ? m->log_width : 0; }

// running statistics of a freshly constructed BatchNorm1d: mean 0, variance 1, no batches tracked
static int reset_bn_state(dcv_mlp* m, hipStream_t s) {
    for (auto& p : m->layers) {
        if (!p.bn) continue;
        DCV_CHECK_HIP(hipMemsetAsync(p.rm, 0, (size_t)p.out * sizeof(float), s));
        hipLaunchKernelGGL(fill_kernel, dim3(1), dim3(256), 0, s, p.rv, (int64_t)p.out, 1.f);
        DCV_CHECK_LAUNCH();
        p.bn_batches = 0;
    }
    return DCV_OK;
}

extern "C" int dcv_mlp_set_params(dcv_mlp* m, const float* params_h, void* stream) {
    DCV_REQUIRE(m && params_h, "dcv_mlp_set_params: null argument");
    hipStream_t s = as_stream(stream);
    DCV_CHECK_HIP(hipMemcpyAsync(m->params, params_h, m->n_params * sizeof(float), hipMemcpyHostToDevice, s));
    int rc = reset_opt_state(m, s);
    if (rc == DCV_OK) rc = reset_bn_state(m, s);
    if (rc == DCV_OK) rc = snet_image_repack(m, s);   // the fused small-network kernels' weight image follows the parameters
    if (rc) return rc;
    DCV_CHECK_HIP(hipStreamSynchronize(s));
    return DCV_OK;
}

extern "C" int dcv_mlp_get_params(dcv_mlp* m, float* params_h, void* stream) {
    DCV_REQUIRE(m && params_h, "dcv_mlp_get_params: null argument");
    hipStream_t s = as_stream(stream);
    DCV_CHECK_HIP(hipMemcpyAsync(params_h, m->params, m->n_params * sizeof(float), hipMemcpyDeviceToHost, s));
    DCV_CHECK_HIP(hipStreamSynchronize(s));
    return DCV_OK;
}

extern "C" int dcv_mlp_bn_state(dcv_mlp* m, int32_t layer, float* running_mean_h, float* running_var_h, int64_t* num_batches_tracked,
                                int32_t set, void* stream) {
    DCV_REQUIRE(m && layer >= 0 && layer < m->L && running_mean_h && running_var_h && num_batches_tracked, "dcv_mlp_bn_state: bad arguments");
    LayerPlan& p = m->layers[layer];
    DCV_REQUIRE(p.bn, "dcv_mlp_bn_state: layer %d has no batch normalisation", layer);
    hipStream_t s = as_stream(stream);
    const size_t bytes = (size_t)p.out * sizeof(float);
    if (set) {
        DCV_CHECK_HIP(hipMemcpyAsync(p.rm, running_mean_h, bytes, hipMemcpyHostToDevice, s));
        DCV_CHECK_HIP(hipMemcpyAsync(p.rv, running_var_h, bytes, hipMemcpyHostToDevice, s));
        p.bn_batches = *num_batches_tracked;
    } else {
        DCV_CHECK_HIP(hipMemcpyAsync(running_mean_h, p.rm, bytes, hipMemcpyDeviceToHost, s));
        DCV_CHECK_HIP(hipMemcpyAsync(running_var_h, p.rv, bytes, hipMemcpyDeviceToHost, s));
        *num_batches_tracked = p.bn_batches;
    }
    DCV_CHECK_HIP(hipStreamSynchronize(s));
    return DCV_OK;
}

extern "C" int dcv_mlp_set_row_sharing(dcv_mlp* m, int32_t enable) {
    DCV_REQUIRE(m, "dcv_mlp_set_row_sharing: null");
    m->no_row_sharing = enable ? 0 : 1;
    return DCV_OK;
}

extern "C" int dcv_mlp_set_lr(dcv_mlp* m, double lr) {
    DCV_REQUIRE(m, "dcv_mlp_set_lr: null");
    m->lr = lr;
    return DCV_OK;
}

extern "C" int dcv_mlp_set_momentum(dcv_mlp* m, double value) {
    DCV_REQUIRE(m, "dcv_mlp_set_momentum: null");
    m->momentum_rt = value;
    return DCV_OK;
}

extern "C" int dcv_mlp_set_upper_grads_callback(dcv_mlp* m, void (*fn)(void*), void* user) {
    DCV_REQUIRE(m, "dcv_mlp_set_upper_grads_callback: null");
    m->upper_cb = fn;
    m->upper_cb_user = user;
    return DCV_OK;
}

extern "C" int dcv_mlp_set_rank(dcv_mlp* m, int32_t rank) {
    DCV_REQUIRE(m && rank >= 0, "dcv_mlp_set_rank: bad arguments");
    m->drop_rank = (uint32_t)rank;
    return DCV_OK;
}

extern "C" int dcv_mlp_layer_output(dcv_mlp* m, int32_t layer, int64_t rows, float* out_d, void* stream) {
    DCV_REQUIRE(m && out_d && layer >= 0 && layer < m->L && rows >= 1 && rows <= m->rows_cap, "dcv_mlp_layer_output: bad arguments");
    if (m->last_path != 0) {
        set_error("dcv_mlp_layer_output: the last forward ran as a fused small-network launch (activations never left LDS); set DCV_NO_SNET=1");
        return DCV_ESTATE;
    }
    const LayerPlan& p = m->layers[layer];
    DCV_CHECK_HIP(hipMemcpy2DAsync(out_d, (size_t)p.out * sizeof(float), p.H, (size_t)p.ldh * sizeof(float), (size_t)p.out * sizeof(float),
                                   (size_t)rows, hipMemcpyDeviceToDevice, as_stream(stream)));
    return DCV_OK;
}

extern "C" int64_t dcv_mlp_dropout_step(const dcv_mlp* m) { return m ? m->drop_step : 0; }
extern "C" int32_t dcv_mlp_last_path(const dcv_mlp* m) { return m ? m->last_path : -1; }

extern "C" int dcv_mlp_dropout_mask(dcv_mlp* m, int32_t layer, int64_t step, int64_t rows, float* out_d, void* stream) {
    DCV_REQUIRE(m && out_d && layer >= 0 && layer < m->L && rows >= 1 && step >= 0, "dcv_mlp_dropout_mask: bad arguments");
    const bool keep_mode = m->fwd_train;
    const int64_t keep_step = m->cur_step;
    m->fwd_train = true;
    m->cur_step = step;
    const DropCfg dc = drop_cfg(m, layer);
    m->fwd_train = keep_mode;
    m->cur_step = keep_step;
    const int width = m->layers[layer].out;
    int64_t blocks = cdiv(rows * ((width + 3) / 4), 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(dropout_mask_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), out_d, rows, width, dc);
    DCV_CHECK_LAUNCH();
    return DCV_OK;
}

extern "C" int dcv_mlp_set_feature_range(dcv_mlp* m, const float* range_h, void* stream) {
    DCV_REQUIRE(m && range_h, "dcv_mlp_set_feature_range: null argument");
    hipStream_t s = as_stream(stream);
    DCV_CHECK_HIP(hipMemcpyAsync(m->feat_range, range_h, m->desc.dims[0] * sizeof(float), hipMemcpyHostToDevice, s));
    DCV_CHECK_HIP(hipStreamSynchronize(s));
    return DCV_OK;
}

extern "C" int dcv_mlp_reset_log(dcv_mlp* m, int32_t capacity, void* stream) {
    DCV_REQUIRE(m && capacity >= 1, "dcv_mlp_reset_log: bad arguments");
    hipStream_t s = as_stream(stream);
    if (capacity > m->log_cap) {
        DCV_CHECK_HIP(hipStreamSynchronize(s));
        if (m->log) (void)hipFree(m->log);
        m->log = nullptr;
        m->log_cap = 0;
        int rc = dmalloc(&m->log, (size_t)capacity * m->log_width);
        if (rc) return rc;
        m->log_cap = capacity;
    }
    DCV_CHECK_HIP(hipMemsetAsync(m->log_count, 0, sizeof(int), s));
    return DCV_OK;
}

extern "C" int dcv_mlp_read_log(dcv_mlp* m, double* out_h, int32_t max_records, int32_t* n_records, void* stream) {
    DCV_REQUIRE(m && out_h && n_records, "dcv_mlp_read_log: null argument");
    hipStream_t s = as_stream(stream);
    int cnt = 0;
    DCV_CHECK_HIP(hipMemcpyAsync(&cnt, m->log_count, sizeof(int), hipMemcpyDeviceToHost, s));
    DCV_CHECK_HIP(hipStreamSynchronize(s));
    if (cnt > m->log_cap) cnt = m->log_cap;
    if (cnt > max_records) cnt = max_records;
    if (cnt > 0) {
        DCV_CHECK_HIP(hipMemcpyAsync(out_h, m->log, (size_t)cnt * m->log_width * sizeof(double), hipMemcpyDeviceToHost, s));
        DCV_CHECK_HIP(hipStreamSynchronize(s));
    }
    *n_records = cnt;
    return DCV_OK;
}

static bool act_mask_enabled() {
    static const bool off = [] { const char* e = getenv("DCV_NO_ACT_MASK"); return e && e[0] == '1'; }();
    return !off;
}
// layer l + 1 can ride in the epilogue of layer l: it is narrow and layer l's output fits one column tile
static bool next_layer_fusable(const dcv_mlp* m, int l) {
    static const bool off = [] { const char* e = getenv("DCV_NO_HEAD_FUSION"); return e && e[0] == '1'; }();
    if (off || l + 1 >= m->L || m->any_bn) return false;
    if (m->desc.dropout[l + 1] > 0.f) return false;
    return m->layers[l + 1].out <= 8 && m->layers[l].out <= 128;
}

// what the layer behind Linear l hands on: the batch-normalised values when it has a normalisation, else the activations
static inline float* layer_out(const dcv_mlp* m, int l) { return m->layers[l].bn ? m->layers[l].Y : m->layers[l].H; }

// forward through layers [0, n_run) for `rows` logical rows
// (dropout follows m->fwd_train, which the callers set: training forward on, everything else off)
static int run_forward(dcv_mlp* m, const float* Xn, int64_t ld, const RowMap& rows_map, int64_t rows, int n_run, hipStream_t s,
                       bool for_backward = false) {
    for (int l = 0; l < n_run; ++l) {
        LayerPlan& p = m->layers[l];
        p.mask_rows = -1;
        Operand A = l == 0 ? make_operand(Xn, ld, p.in, rows_map) : make_operand(layer_out(m, l - 1), m->layers[l - 1].ldh, p.in);
        Operand B = make_operand(m->params + p.w_off, p.in, p.in);
        if (l + 1 < n_run && next_layer_fusable(m, l)) {
            // the narrow Linear behind this layer rides in its epilogue (the whole row of H is in the workgroup)
            LayerPlan& nx = m->layers[l + 1];
            const bool vec = quad_ok(p.H, p.ldh) && quad_ok(m->params + p.b_off, 4) && quad_ok(m->params + nx.w_off, nx.in);
            prof_mark(m, l, 0, 0, s);
            int rc;
            if (nx.out <= 4) {
                EpiBiasActHead<4> epi{p.H, p.ldh, m->params + p.b_off, p.act, vec, m->params + nx.w_off, nx.in, m->params + nx.b_off, nx.out, nx.act, nx.H, nx.ldh};
                epi.drop = drop_cfg(m, l);
                rc = gemm_nt_head4(A, B, rows, p.out, p.in, epi, s, &m->tail);
            } else {
                EpiBiasActHead<8> epi{p.H, p.ldh, m->params + p.b_off, p.act, vec, m->params + nx.w_off, nx.in, m->params + nx.b_off, nx.out, nx.act, nx.H, nx.ldh};
                epi.drop = drop_cfg(m, l);
                rc = gemm_nt_head8(A, B, rows, p.out, p.in, epi, s, &m->tail);
            }
            if (rc) return rc;
            prof_mark(m, l, 0, 1, s);
            prof_mark(m, l + 1, 0, 0, s);
            prof_mark(m, l + 1, 0, 1, s);
            ++l;
            continue;
        }
        EpiBiasAct epi{p.H, p.ldh, m->params + p.b_off, p.act, quad_ok(p.H, p.ldh) && quad_ok(m->params + p.b_off, 4)};
        epi.drop = drop_cfg(m, l);
        if (for_backward && p.mask && act_mask_enabled()) {   // the dgrad of the next layer reads sign(H) instead of H
            epi.mask = p.mask;
            p.mask_rows = rows;
        }
        prof_mark(m, l, 0, 0, s);
        if (p.bn) {   // the derivative of the normalised layer needs the activations themselves, not their signs
            epi.mask = nullptr;
            p.mask_rows = -1;
        }
        int rc = gemm_nt_bias_act(A, B, rows, p.out, p.in, epi, s, &m->tail);
        if (rc) return rc;
        if (p.bn) {
            // training: batch statistics of every forward call of the step -- a Deep-TICA batch is two (x_t rows, then x_lag
            // rows), each normalised by its own statistics and each updating the running ones; evaluation: running statistics
            const bool train = m->fwd_train && for_backward;
            if (train && rows_map.half > 0) {
                rc = bn_forward(m, l, 0, rows_map.half, true, s);
                if (rc == DCV_OK) rc = bn_forward(m, l, rows_map.half, rows - rows_map.half, true, s);
            } else {
                rc = bn_forward(m, l, 0, rows, train, s);
            }
            if (rc) return rc;
        }
        prof_mark(m, l, 0, 1, s);
    }
    return DCV_OK;
}

// Optional (DCV_GRAPH=1; off by default): the launch sequence of a step is captured into a hipGraph each call and the
// instantiated graph of the slot is UPDATED in place (same topology, new kernel arguments: batch offset, Adam bias
// corrections), then launched once.  Needs a capturing-capable (non-null) stream; anything else -- null stream,
// profiling on, first call of a slot, an update the runtime refuses -- takes the plain launches.  Measured on
// MI355X / ROCm 7.2 it buys nothing: the 27 us of gaps per step stay (2.909 vs 2.910 ms at the bench size, 0.454 vs
// 0.449 ms at one eighth of it, and the per-call capture costs the 8192-pair step 3 %), so plain launches are the
// default and the path is kept as a tested option (tests/test_mlp_gpu.py::test_graphed_steps_match_plain_launches).
template <class F>
static int run_graphed(dcv_mlp* m, int slot, hipStream_t s, F&& body) {
    if (!m->graph_on || m->graph_off || s == nullptr || (m->prof_level > 0 && !m->prof_paused) || !m->gwarm[slot]) {
        m->gwarm[slot] = true;
        return body();
    }
    if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) {
        (void)hipGetLastError();
        return body();
    }
    // the host state a replay of body() must not advance twice: step counters, NAdam's mu_product, ASGD's eta, the
    // batch counts of the normalisations (all advanced inside body(): next_opt_args, bn_forward)
    struct HostStep {
        int64_t adam_t, drop_step, cur_step;
        double mu_product, eta;
        int64_t bn_batches[DCV_MAX_LAYERS];
    };
    auto snap = [&]() {
        HostStep h{m->adam_t, m->drop_step, m->cur_step, m->nadam_mu_product, m->asgd_eta, {}};
        for (int l = 0; l < m->L; ++l) h.bn_batches[l] = m->layers[l].bn_batches;
        return h;
    };
    auto restore = [&](const HostStep& h) {
        m->adam_t = h.adam_t; m->drop_step = h.drop_step; m->cur_step = h.cur_step;
        m->nadam_mu_product = h.mu_product; m->asgd_eta = h.eta;
        for (int l = 0; l < m->L; ++l) m->layers[l].bn_batches = h.bn_batches[l];
    };
    const HostStep h0 = snap();
    const int rc = body();
    hipGraph_t g = nullptr;
    const hipError_t ec = hipStreamEndCapture(s, &g);
    if (rc != DCV_OK || ec != hipSuccess || g == nullptr) {
        if (g) (void)hipGraphDestroy(g);
        (void)hipGetLastError();
        if (rc != DCV_OK) return rc;
        m->gwarm[slot] = false;   // capture failed: run this call (and relearn) without it
        restore(h0);
        return body();
    }
    if (m->gexec[slot]) {
        hipGraphNode_t err_node = nullptr;
        hipGraphExecUpdateResult res = hipGraphExecUpdateSuccess;
        if (hipGraphExecUpdate(m->gexec[slot], g, &err_node, &res) != hipSuccess || res != hipGraphExecUpdateSuccess) {
            (void)hipGetLastError();
            (void)hipGraphExecDestroy(m->gexec[slot]);
            m->gexec[slot] = nullptr;
        }
    }
    if (!m->gexec[slot] && hipGraphInstantiate(&m->gexec[slot], g, nullptr, nullptr, 0) != hipSuccess) {
        (void)hipGetLastError();
        m->gexec[slot] = nullptr;
    }
    (void)hipGraphDestroy(g);
    if (!m->gexec[slot]) {   // nothing was launched yet: replay as plain launches and stop trying on this slot
        m->graph_off = true;
        restore(h0);
        return body();
    }
    DCV_CHECK_HIP(hipGraphLaunch(m->gexec[slot], s));
    m->graph_launches += 1;
    return DCV_OK;
}

// fuse_head: 0 = statistics only (a data-parallel caller all-reduces them before dcv_mlp_backward); 1 / 2 = one-GPU
// training / evaluation step: the last block of the statistics launch also runs the loss head (batch = global batch)
static int forward_impl(dcv_mlp* m, const float* Xn_d, int64_t ld, const int64_t* idx_d, int64_t row0, int32_t batch,
                        int32_t train, void* stream, int fuse_head = 0) {
    DCV_REQUIRE(m && Xn_d, "dcv_mlp_forward: null argument");
    g_launch_ev = LaunchEvents{};   // an offer left behind by a launch that failed half way
    m->head_done = false;
    m->fwd_train = train != 0;
    if (m->fwd_train) m->cur_step = m->drop_step++;
    DCV_REQUIRE(batch >= 1 && batch <= m->desc.max_batch, "dcv_mlp_forward: batch=%d exceeds max_batch=%d", batch, m->desc.max_batch);
    DCV_REQUIRE(ld >= m->desc.dims[0], "dcv_mlp_forward: ld=%lld < F=%d", (long long)ld, m->desc.dims[0]);
    hipStream_t s = as_stream(stream);
    m->snet_fwd_valid = false;
    m->last_path = 0;
    if (m->desc.model == DCV_MODEL_DEEPTICA && !(m->snet_dt_tried && m->snet_dt == nullptr)) {
        // a network that fits in LDS: forward, batch statistics and (one-GPU steps) the loss head in ONE launch (snet_dt.hip)
        if (fuse_head) DCV_REQUIRE(m->log && m->log_cap > 0, "dcv_mlp step: call dcv_mlp_reset_log first");
        prof_mark(m, 0, 0, 0, s);
        // the loss head runs inside the forward launch only when no backward follows (evaluation step); a training step's
        // head is evaluated by the backward launch's workgroups beside their staging (snet_dt.hip)
        const int rcs = snet_dt_forward(m, Xn_d, ld, idx_d, row0, batch, fuse_head == 2 ? 2 : 0, fuse_head != 2, s);
        if (rcs < 0) return rcs;
        if (rcs == DCV_OK) {
            prof_mark(m, 0, 0, 1, s);
            m->snet_fwd_valid = fuse_head != 2;
            m->last_path = 2;
            m->last_batch = batch;
            m->head_done = fuse_head == 2;
            return DCV_OK;
        }
        if (prof_on(m, 0)) g_launch_ev = LaunchEvents{};   // not applicable: the layer-by-layer path marks its own launches
    }
    const RowMap rm = batch_rows(m, idx_d, row0, batch);
    const int64_t R = rows_of(m, idx_d, batch);
    int rc = run_forward(m, Xn_d, ld, rm, R, m->L, s, fuse_head != 2);   // the one-GPU evaluation step has no backward: no sign masks
    if (rc) return rc;
    const LayerPlan& last = m->layers[m->L - 1];
    const float* net_out = layer_out(m, m->L - 1);   // the network's output: behind the last layer's normalisation when it has one
    if (m->desc.model == DCV_MODEL_DEEPTICA) {
        int nb;
        if (tica_stats_fn_t fast = tica_stats_rows_fn(m->d_out)) {
            const int rpb = stats_rows_per_block(batch);
            nb = (int)cdiv(batch, rpb);
            FusedHead fh{0, 0.0, 0.0, nullptr, nullptr, nullptr, 0, 0};
            if (fuse_head) {
                DCV_REQUIRE(m->log && m->log_cap > 0, "dcv_mlp step: call dcv_mlp_reset_log first");
                fh = FusedHead{1, (double)batch, m->desc.tica_reg, fuse_head == 1 ? m->gradp : nullptr, m->log, m->log_count, m->log_cap, m->log_width};
            }
            hipLaunchKernelGGL(fast, dim3(nb), dim3(256), 0, s, net_out, last.ldh, (int)batch, lag_offset(m, idx_d, batch), rpb,
                               m->spart, m->ticket, m->stats, fh);
            DCV_CHECK_LAUNCH();
            m->last_batch = batch;
            m->head_done = fuse_head != 0;
            return DCV_OK;
        } else {
            nb = (int)cdiv(batch, kStatBlockRows);
            hipLaunchKernelGGL(tica_stats_kernel, dim3(nb), dim3(256), (size_t)2 * kStatBlockRows * m->d_out * sizeof(double), s, net_out,
                               last.ldh, batch, m->d_out, lag_offset(m, idx_d, batch), m->spart);
        }
        DCV_CHECK_LAUNCH();
        hipLaunchKernelGGL(sum_partials_kernel, dim3(m->stats_len), dim3(64), 0, s, m->spart, nb, m->stats_len, m->stats);
        DCV_CHECK_LAUNCH();
    } else {
        // Two opposing costs: every block ends on a release fence + ticket (~70 ns apiece, serialised: 1024 blocks measured
        // 81 us for a 2 MB pass), and every 8 elements per thread are one more round trip of loads (64 blocks x 32 elements
        // per thread measured 27 us at 4096 x 128).  16 elements per thread, at most 128 blocks up to 4M elements, then
        // 64 elements per thread up to 512 blocks; never fewer than kSseRows rows per block.
        const int64_t elems = R * (int64_t)m->desc.dims[0];
        int64_t want = cdiv(elems, 256 * 16);
        if (want > 128) want = cdiv(elems, 256 * 64) > 128 ? cdiv(elems, 256 * 64) : 128;
        if (want > 512) want = 512;
        if (want < 1) want = 1;
        int64_t rpb = cdiv(R, want);
        if (rpb < kSseRows) rpb = kSseRows;
        const int nb = (int)cdiv(R, rpb);
        if (fuse_head) {   // one-GPU step: final sum and loss record in the last block of the same launch
            DCV_REQUIRE(m->log && m->log_cap > 0, "dcv_mlp step: call dcv_mlp_reset_log first");
            hipLaunchKernelGGL(ae_sse_kernel, dim3(nb), dim3(256), 0, s, net_out, last.ldh, Xn_d, ld, rm, R, m->desc.dims[0], m->feat_range, m->spart,
                               m->ticket, m->stats, (double)batch, m->log, m->log_count, m->log_cap, m->log_width, (int)rpb);
            DCV_CHECK_LAUNCH();
            m->head_done = true;
        } else {
            hipLaunchKernelGGL(ae_sse_kernel, dim3(nb), dim3(256), 0, s, net_out, last.ldh, Xn_d, ld, rm, R, m->desc.dims[0], m->feat_range, m->spart,
                               (unsigned*)nullptr, (double*)nullptr, 0.0, (double*)nullptr, (int*)nullptr, 0, 0, (int)rpb);
            DCV_CHECK_LAUNCH();
            hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(64), 0, s, m->spart, nb, 1, m->stats);
            DCV_CHECK_LAUNCH();
        }
    }
    m->last_batch = batch;
    return DCV_OK;
}

// Gradient reduction of the layers [l0, l1) of `ra` (split-K slabs + bias partials -> m->grads), optionally with the
// optimiser update fused in.
static int launch_reduce(dcv_mlp* m, const ReduceArgs& ra_all, int l0, int l1, bool fuse_opt, const OptArgs& oa, hipStream_t s) {
    ReduceArgs ra;
    ra.L = 0;
    int max_splits = 0, max_bblocks = 0;
    int64_t max_total = 0;
    auto take = [&](const ReduceDesc& d) {
        ra.l[ra.L++] = d;
        if (d.splits > max_splits) max_splits = d.splits;
        if (d.bblocks > max_bblocks) max_bblocks = d.bblocks;
        if (d.w_count + d.out > max_total) max_total = d.w_count + d.out;
    };
    for (int l = l0; l < l1; ++l) take(ra_all.l[l]);
    for (int l = l0; l < l1; ++l)
        if (m->layers[l].bn) take(ra_all.l[m->L + l]);   // weight / bias of the batch normalisation behind layer l
    if (ra.L <= 0) return DCV_OK;
    static const bool quad_off = [] { const char* e = getenv("DCV_REDUCE_QUAD"); return e && e[0] == '0'; }();
    if (max_splits <= 512 && max_bblocks <= 1024 && !quad_off) {
        QuadArgs qa;
        qa.n = 0;
        int64_t blocks = 0;
        auto item = [&](const float* src, int64_t dst, int64_t count, int parts, int64_t stride) {
            if (count <= 0) return;
            if (!src || parts < 0) parts = 0;   // no partials: a zero gradient, as the other two kernels give
            QuadItem& q = qa.it[qa.n++];
            q = QuadItem{src, dst, stride, (int)count, parts, (int)blocks, quad_groups(parts)};
            blocks += cdiv(count, 1024 / q.groups);
        };
        for (int l = 0; l < ra.L; ++l) {
            DCV_REQUIRE(ra.l[l].w_count < (1ll << 31), "reduce: layer too large");
            item(ra.l[l].slab, ra.l[l].w_off, ra.l[l].w_count, ra.l[l].splits, rd_wstride(ra.l[l]));
            item(ra.l[l].bpart, ra.l[l].b_off, ra.l[l].out, ra.l[l].bblocks, rd_bstride(ra.l[l]));
        }
        if (blocks <= 0) return DCV_OK;
        DCV_REQUIRE(blocks < (1ll << 31), "reduce: grid out of range");
        hipLaunchKernelGGL(reduce_grads_quad_kernel, dim3((unsigned)blocks), dim3(256), 0, s, qa, m->grads, 1.f, fuse_opt ? 1 : 0, m->params,
                           m->adam_m, m->adam_v, m->opt_aux, oa);
    } else if (max_splits <= 512 && max_bblocks <= 1024) {   // few partials per weight (the few bias elements may see more)
        int64_t bx = cdiv(max_total, 64);
        if (bx > 2048) bx = 2048;
        hipLaunchKernelGGL(reduce_grads_small_kernel, dim3((unsigned)bx, ra.L), dim3(256), 0, s, ra, m->grads, 1.f, fuse_opt ? 1 : 0, m->params,
                           m->adam_m, m->adam_v, m->opt_aux, oa);
    } else {
        hipLaunchKernelGGL(reduce_grads_kernel, dim3(512, ra.L), dim3(64 * kRedWaves), 0, s, ra, m->grads, 1.f, fuse_opt ? 1 : 0, m->params, m->adam_m,
                           m->adam_v, m->opt_aux, oa);
    }
    DCV_CHECK_LAUNCH();
    return DCV_OK;
}

static OptArgs next_opt_args(dcv_mlp* m);
// state that torch.optim creates lazily in its first step(): called (with the stream of the update) before next_opt_args
static int first_step_state(dcv_mlp* m, hipStream_t s) {
    if (m->adam_t != 0) return DCV_OK;
    if (m->desc.optimizer == DCV_OPT_RPROP) {   // step_size = full_like(grad, lr)
        hipLaunchKernelGGL(fill_kernel, dim3(256), dim3(256), 0, s, m->adam_v, m->n_params, (float)m->lr);
        DCV_CHECK_LAUNCH();
    }
    if (m->desc.optimizer == DCV_OPT_ASGD) m->asgd_eta = (double)(float)m->lr;   // eta = as_tensor(lr)
    return DCV_OK;
}
static int backward_impl(dcv_mlp* m, const float* Xn_d, int64_t ld, const int64_t* idx_d, int64_t row0, int32_t batch,
                         int64_t global_batch, int32_t train, void* stream, bool fuse_opt = false) {
    DCV_REQUIRE(m && Xn_d, "dcv_mlp_backward: null argument");
    g_launch_ev = LaunchEvents{};
    if (m->last_batch != batch) {
        set_error("dcv_mlp_backward: batch=%d does not match the preceding forward (%d)", batch, m->last_batch);
        return DCV_ESTATE;
    }
    DCV_REQUIRE(global_batch >= batch, "dcv_mlp_backward: global_batch=%lld < batch=%d", (long long)global_batch, batch);
    if (m->any_drop && train && !m->fwd_train) {
        set_error("dcv_mlp_backward: train=1 after an evaluation-mode forward (dropout masks would not match)");
        return DCV_ESTATE;
    }
    DCV_REQUIRE(m->log && m->log_cap > 0, "dcv_mlp_backward: call dcv_mlp_reset_log first");
    hipStream_t s = as_stream(stream);
    const RowMap rm = batch_rows(m, idx_d, row0, batch);
    const int64_t R = rows_of(m, idx_d, batch);
    const int L = m->L;
    const LayerPlan& last = m->layers[L - 1];
    // a normalised last layer: the loss gradient is taken w.r.t. the normalised output; activation derivative and dropout
    // of the Linear underneath are applied by the normalisation's backward pass
    const float* net_out = layer_out(m, L - 1);
    const int last_act = last.bn ? DCV_ACT_NONE : last.act;
    const DropCfg last_drop = last.bn ? kNoDrop : drop_cfg(m, L - 1);
    const float last_hscale = last.bn ? 1.f : drop_hscale(m, L - 1);
    float* dz_cur = m->dZ[0];
    float* dz_nxt = m->dZ[1];
    bool fused_head = false;
    if (m->desc.model == DCV_MODEL_DEEPTICA) {
        const bool head_in_bwd = m->last_path == 2 && train && !m->head_done && m->snet_fwd_valid;   // the fused backward evaluates the head itself
        if (!m->head_done && !head_in_bwd) {
            if (TicaGradWaveFn wf = tica_grad_wave_fn(m->d_out)) {
                hipLaunchKernelGGL(wf, dim3(1), dim3(64), 0, s, (const double*)m->stats, (double)global_batch, m->desc.tica_reg,
                                   train ? m->gradp : nullptr, m->log, m->log_count, m->log_cap, m->log_width);
            } else {
                hipLaunchKernelGGL(tica_grad_fn(m->d_out), dim3(1), dim3(64), 0, s, m->stats, m->d_out, (double)global_batch, m->desc.tica_reg,
                                   train ? m->gradp : nullptr, m->log, m->log_count, m->log_cap, m->log_width);
            }
            DCV_CHECK_LAUNCH();
        }
        m->head_done = false;
        if (!train) return DCV_OK;
        if (m->last_path == 2) {
            // the forward ran fused (snet_dt.hip): one backward launch from its blob, then the reduction (+ optimiser)
            if (!m->snet_fwd_valid) {
                set_error("dcv_mlp_backward: the fused forward of this batch kept no activations (evaluation step)");
                return DCV_ESTATE;
            }
            ReduceArgsView v;
            prof_mark(m, 0, 1, 0, s);
            int rcb = snet_dt_backward(m, batch, global_batch, head_in_bwd, &v, s);
            if (rcb) return rcb;
            prof_mark(m, 0, 1, 1, s);
            ReduceArgs raf{};
            raf.L = L;
            for (int l = 0; l < L; ++l) {
                const LayerPlan& p = m->layers[l];
                ReduceDesc& rd = raf.l[l];
                rd.slab = v.slab[l];
                rd.bpart = v.bpart[l];
                rd.w_off = p.w_off;
                rd.b_off = p.b_off;
                rd.w_count = (int64_t)p.out * p.in;
                rd.out = p.out;
                rd.splits = v.splits[l];
                rd.bblocks = v.bblocks[l];
                rd.w_stride = v.wstride[l];
                rd.b_stride = v.bstride[l];
            }
            bool upper = false;
            if (L > 1 && m->upper_cb && !fuse_opt) {   // data-parallel overlap hook: the upper layers' gradients first
                rcb = launch_reduce(m, raf, 1, L, false, OptArgs{}, s);
                if (rcb) return rcb;
                upper = true;
                m->upper_cb(m->upper_cb_user);
            }
            OptArgs oaf{};
            if (fuse_opt) {
                rcb = first_step_state(m, s);
                if (rcb) return rcb;
                oaf = next_opt_args(m);
            }
            rcb = launch_reduce(m, raf, 0, upper ? 1 : L, fuse_opt, oaf, s);
            if (rcb) return rcb;
            return DCV_OK;
        }
        fused_head = head_fusable(m);
        if (!fused_head) {
            hipLaunchKernelGGL(tica_dF_kernel, dim3((unsigned)cdiv(R, 256)), dim3(256), 0, s, net_out, last.ldh, batch, m->d_out,
                               lag_offset(m, idx_d, batch), m->gradp, last_act, dz_cur, m->ld_dz, last_drop, last_hscale);
            DCV_CHECK_LAUNCH();
        }
    } else {
        const int F = m->desc.dims[0];
        if (!m->head_done) {
            hipLaunchKernelGGL(ae_log_kernel, dim3(1), dim3(64), 0, s, m->stats, (double)global_batch, F, m->log, m->log_count, m->log_cap, m->log_width);
            DCV_CHECK_LAUNCH();
        }
        m->head_done = false;
        if (!train) return DCV_OK;
        const float scale = (float)(2.0 / ((double)global_batch * (double)F));
        int64_t blocks = cdiv(R * F, 256);
        const int64_t cap = (int64_t)num_cus() * 16;
        if (blocks > cap) blocks = cap;
        hipLaunchKernelGGL(ae_dY_kernel, dim3((unsigned)blocks), dim3(256), 0, s, net_out, last.ldh, Xn_d, ld, rm, R, F, m->feat_range, scale,
                           last_act, dz_cur, m->ld_dz, last_drop, last_hscale);
        DCV_CHECK_LAUNCH();
    }
    ReduceArgs ra{};
    ra.L = L;
    // bias-gradient partials of the last layer come from a column-sum pass over dZ_last; those of
    // every other layer fall out of the dgrad epilogue that produces its dZ
    int bblocks = (int)cdiv(R, kColsumRows);
    if (!fused_head && !last.bn) {
        hipLaunchKernelGGL(colsum_kernel, dim3(bblocks), dim3(256), 0, s, dz_cur, R, last.out, m->ld_dz, m->layers[L - 1].bpart);
        DCV_CHECK_LAUNCH();
    }
    bool upper_done = false;
    for (int l = L - 1; l >= 0; --l) {
        LayerPlan& p = m->layers[l];
        if (l == 0 && L > 1 && m->upper_cb && !fuse_opt) {
            // Data-parallel overlap: everything the gradients of layers 1 .. L-1 need has been enqueued (their slabs, and
            // the bias partials of every layer).  Reduce them now and tell the caller, who starts their all-reduce on a
            // side stream while the largest product of the step -- the layer-0 weight gradient -- still runs here.
            int rcu = launch_reduce(m, ra, 1, L, false, OptArgs{}, s);
            if (rcu) return rcu;
            upper_done = true;
            m->upper_cb(m->upper_cb_user);
        }
        if (p.bn) {
            // dz_cur holds dL/d(normalised output): back through the normalisation, the dropout and the activation of this
            // layer, in place; its passes also leave the gradient partials of the normalisation's weight / bias and the
            // bias-gradient partials of this Linear
            const bool two = rm.half > 0;   // two forward calls (x_t rows, x_lag rows), each with its own statistics
            int nb = 0;
            int rcb = bn_backward(m, l, dz_cur, m->ld_dz, two ? 2 : 1, two ? (int64_t)rm.half : R, p.act, drop_hscale(m, l), drop_cfg(m, l), &nb, s);
            if (rcb) return rcb;
            bblocks = nb;
            ReduceDesc& bd = ra.l[L + l];
            bd.slab = p.bn_gpart;
            bd.bpart = p.bn_bpart;
            bd.w_off = p.g_off;
            bd.b_off = p.be_off;
            bd.w_count = p.out;
            bd.out = p.out;
            bd.splits = nb;
            bd.bblocks = nb;
        }
        // wgrad: dW = dZ^T In  (M = out, N = in, K = rows)
        int64_t kc, splits;
        wgrad_plan(p.out, p.in, R, &kc, &splits);
        if (splits > p.max_splits) {
            splits = p.max_splits;
            kc = cdiv(cdiv(R, splits), 32) * 32;
            splits = cdiv(R, kc);
        }
        if (fused_head && l == L - 1) {
            // loss gradient, both bias gradients, wgrad and dgrad of the narrow last layer in one pass over H_{L-2}
            LayerPlan& q = m->layers[l - 1];
            const int D = p.out, K = p.in;
            head_plan(m, R, &kc, &splits);
            prof_mark(m, l, 1, 0, s);
            hipLaunchKernelGGL(head_backward_fn(D), dim3((unsigned)splits), dim3(256), head_lds_bytes(D, K), s, (const float*)p.H, p.ldh,
                               (int)batch, lag_offset(m, idx_d, batch), (const double*)m->gradp, p.act, (const float*)q.H, q.ldh, K, q.act,
                               (const float*)(m->params + p.w_off), kc, dz_nxt, m->ld_dz, p.slab, p.bpart, q.bpart, drop_cfg(m, l - 1),
                               drop_hscale(m, l - 1));
            DCV_CHECK_LAUNCH();
            prof_mark(m, l, 1, 1, s);
            prof_mark(m, l, 2, 0, s);
            prof_mark(m, l, 2, 1, s);
            bblocks = (int)splits;
            ReduceDesc& rd = ra.l[l];
            rd.slab = p.slab;
            rd.bpart = p.bpart;
            rd.w_off = p.w_off;
            rd.b_off = p.b_off;
            rd.w_count = (int64_t)p.out * p.in;
            rd.out = p.out;
            rd.splits = (int)splits;
            rd.bblocks = bblocks;
            float* tmp = dz_cur;
            dz_cur = dz_nxt;
            dz_nxt = tmp;
            continue;
        }
        Operand A = make_operand(dz_cur, m->ld_dz, p.out);
        Operand B = l == 0 ? make_operand(Xn_d, ld, p.in, rm) : make_operand(layer_out(m, l - 1), m->layers[l - 1].ldh, p.in);
        EpiSlab epi{p.slab, p.out, p.in, 1, 0, quad_ok(p.slab, p.in), p.max_splits};
        ReduceDesc& rd = ra.l[l];
        rd.slab = p.slab;
        rd.bpart = p.bpart;
        rd.w_off = p.w_off;
        rd.b_off = p.b_off;
        rd.w_count = (int64_t)p.out * p.in;
        rd.out = p.out;
        rd.splits = (int)splits;
        rd.bblocks = bblocks;
        if (l == 0) {
            prof_mark(m, l, 1, 0, s);
            int rc = gemm_tn_slab(A, B, p.out, p.in, R, kc, epi, s);
            if (rc) return rc;
            prof_mark(m, l, 1, 1, s);
            continue;
        }
        // dgrad: dZ_prev = (dZ W) * act'(H_prev)   (M = rows, N = in, K = out)
        LayerPlan& q = m->layers[l - 1];
        Operand Ad = make_operand(dz_cur, m->ld_dz, p.out);
        Operand Bd = make_operand(m->params + p.w_off, p.in, p.in);
        EpiActGrad eg{dz_nxt, m->ld_dz, q.H, q.ldh, q.act, q.bpart, q.out, quad_ok(dz_nxt, m->ld_dz) && quad_ok(q.H, q.ldh)};
        if (q.mask && q.mask_rows == R) {   // written by this step's forward with the same (rows, width) => same tiling
            eg.mask = q.mask;
            eg.slope = q.act == DCV_ACT_LEAKY_RELU ? 0.01f : 0.f;
        }
        eg.drop = drop_cfg(m, l - 1);
        eg.hscale = drop_hscale(m, l - 1);
        if (q.bn) {   // a normalised layer below: hand down the raw product dL/d(its normalised output); bn_backward does the rest
            eg.act = DCV_ACT_NONE;
            eg.mask = nullptr;
            eg.drop = kNoDrop;
            eg.hscale = 1.f;
        }
        // the two products read the same dZ and neither reads the other's output: one launch when the pair form applies
        prof_mark(m, l, 1, 0, s);
        prof_mark(m, l, 2, 0, s);
        int rc = launch_wgrad_dgrad(A, B, p.out, p.in, R, kc, epi, Ad, Bd, R, p.in, p.out, eg, &bblocks, &m->tail, s);
        if (rc < 0) return rc;
        if (rc == 1) {
            rc = gemm_tn_slab(A, B, p.out, p.in, R, kc, epi, s);
            if (rc) return rc;
            prof_mark(m, l, 1, 1, s);
            prof_mark(m, l, 2, 0, s);
            rc = gemm_nn_act_grad(Ad, Bd, R, p.in, p.out, eg, s, &bblocks, &m->tail);
            if (rc) return rc;
            prof_mark(m, l, 2, 1, s);
        } else {
            prof_mark(m, l, 1, 1, s);
            prof_mark(m, l, 2, 1, s);
        }
        float* tmp = dz_cur;
        dz_cur = dz_nxt;
        dz_nxt = tmp;
    }
    OptArgs oa{};
    if (fuse_opt) {
        const int rcf = first_step_state(m, s);
        if (rcf) return rcf;
        oa = next_opt_args(m);
    }
    int rc2 = launch_reduce(m, ra, 0, upper_done ? 1 : L, fuse_opt, oa, s);   // layer 0 only when the upper layers went out early
    if (rc2) return rc2;
    return DCV_OK;
}

extern "C" int dcv_mlp_profile_begin(dcv_mlp* m, int32_t max_steps, int32_t level) {
    DCV_REQUIRE(m && max_steps >= 1 && (level == 1 || level == 2), "dcv_mlp_profile_begin: bad arguments");
    const size_t need = (size_t)3 * m->L * max_steps * 2;
    while (m->prof_ev.size() < need) {
        hipEvent_t e;
        DCV_CHECK_HIP(hipEventCreate(&e));
        m->prof_ev.push_back(e);
    }
    m->prof_cap = max_steps;
    m->prof_cnt.assign((size_t)3 * m->L, 0);
    m->prof_kind_off = 0;
    m->prof_paused = false;
    m->prof_level = level;
    return DCV_OK;
}

extern "C" int dcv_mlp_profile_pause(dcv_mlp* m, int32_t paused) {
    DCV_REQUIRE(m, "dcv_mlp_profile_pause: null");
    m->prof_paused = (paused & 1) != 0;
    m->prof_kind_off = (paused >> 1) & 7;   // bit 1 / 2 / 3: forward / weight-gradient / input-gradient launches carry no events
    return DCV_OK;
}

extern "C" int dcv_mlp_profile_end(dcv_mlp* m, double* ms_h, int32_t* counts_h) {
    DCV_REQUIRE(m && ms_h && counts_h, "dcv_mlp_profile_end: null argument");
    const int level = m->prof_level;
    m->prof_level = 0;
    for (int c = 0; c < 3 * m->L; ++c) {
        ms_h[c] = 0.0;
        counts_h[c] = 0;
        const int layer = c / 3, kind = c % 3;
        if (level < 2 && layer != 0) continue;
        if (kind == 2 && layer == 0) continue;  // the first layer has no dgrad
        const int steps = (size_t)c < m->prof_cnt.size() ? m->prof_cnt[c] : 0;
        for (int i = 0; i < steps; ++i) {
            hipEvent_t a = m->prof_ev[((size_t)c * m->prof_cap + i) * 2 + 0];
            hipEvent_t b = m->prof_ev[((size_t)c * m->prof_cap + i) * 2 + 1];
            DCV_CHECK_HIP(hipEventSynchronize(b));
            float ms = 0.f;
            DCV_CHECK_HIP(hipEventElapsedTime(&ms, a, b));
            ms_h[c] += (double)ms;
            counts_h[c] += 1;
        }
    }
    return DCV_OK;
}

// arguments of the next optimiser update; advances the step count
static OptArgs next_opt_args(dcv_mlp* m) {
    m->adam_t += 1;
    const dcv_mlp_desc& d = m->desc;
    const double t = (double)m->adam_t;
    OptArgs a;
    a.kind = d.optimizer;
    a.flag = 0;
    a.first = m->adam_t == 1;
    a.lr = (float)m->lr;
    a.b1 = a.b2 = a.c1 = a.c2 = a.w1 = a.w2 = 0.f;
    a.p0 = a.p1 = a.p2 = a.p3 = 0.f;
    a.decay = 1.f;
    a.eps = (float)d.eps;
    a.wd = (float)d.weight_decay;
    a.maximize = d.maximize ? 1 : 0;
    a.img = m->snet_img;
    a.img_idx = m->snet_img_idx;
    switch (d.optimizer) {
        case DCV_OPT_ADAM:
        case DCV_OPT_ADAMW: {
            const double b1 = m->momentum_rt, b2 = d.beta2;   // beta1 may be cycled by a scheduler (dcv_mlp_set_momentum)
            a.flag = d.amsgrad ? 1 : 0;
            a.b1 = (float)b1;
            a.b2 = (float)b2;
            a.w1 = (float)(1.0 - b1);
            a.w2 = (float)(1.0 - b2);
            a.c1 = (float)(m->lr / (1.0 - pow(b1, t)));
            a.c2 = (float)sqrt(1.0 - pow(b2, t));
            a.decay = (float)(1.0 - m->lr * d.weight_decay);
            break;
        }
        case DCV_OPT_SGD:
            a.flag = d.nesterov ? 1 : 0;
            a.b1 = (float)m->momentum_rt;
            a.w1 = (float)(1.0 - d.dampening);
            break;
        case DCV_OPT_RMSPROP:
            a.flag = d.centered ? 1 : 0;
            a.b1 = (float)m->momentum_rt;
            a.b2 = (float)d.alpha;
            a.w2 = (float)(1.0 - d.alpha);
            break;
        case DCV_OPT_ADAMAX: {
            const double b1 = m->momentum_rt;
            a.b1 = (float)b1;
            a.b2 = (float)d.beta2;
            a.w1 = (float)(1.0 - b1);
            a.c1 = (float)(m->lr / (1.0 - pow(b1, t)));
            break;
        }
        case DCV_OPT_NADAM: {
            const double b1 = m->momentum_rt, b2 = d.beta2, md = d.opt_p[0];
            a.flag = d.opt_p[1] != 0.0 ? 1 : 0;
            a.b1 = (float)b1;
            a.b2 = (float)b2;
            a.w1 = (float)(1.0 - b1);
            a.w2 = (float)(1.0 - b2);
            a.c2 = (float)(1.0 - pow(b2, t));
            a.decay = (float)(1.0 - m->lr * d.weight_decay);
            const double mu = b1 * (1.0 - 0.5 * pow(0.96, t * md));
            const double mu_next = b1 * (1.0 - 0.5 * pow(0.96, (t + 1.0) * md));
            // the state tensor mu_product is float32: `mu_product *= mu`, then read back through .item()
            m->nadam_mu_product = (double)(float)((double)(float)m->nadam_mu_product * mu);
            const double mp = m->nadam_mu_product, mp_next = mp * mu_next;
            a.p0 = (float)(-m->lr * (1.0 - mu) / (1.0 - mp));
            a.p1 = (float)((-m->lr * mu_next) / (1.0 - mp_next));
            break;
        }
        case DCV_OPT_RADAM: {
            const double b1 = m->momentum_rt, b2 = d.beta2;
            a.flag = d.opt_p[1] != 0.0 ? 1 : 0;
            a.b1 = (float)b1;
            a.b2 = (float)b2;
            a.w1 = (float)(1.0 - b1);
            a.w2 = (float)(1.0 - b2);
            a.decay = (float)(1.0 - m->lr * d.weight_decay);
            const double bc1 = 1.0 - pow(b1, t), bc2 = 1.0 - pow(b2, t);
            a.c1 = (float)bc1;
            a.c2 = (float)sqrt(bc2);
            const double rho_inf = 2.0 / (1.0 - b2) - 1.0;
            const double rho_t = rho_inf - 2.0 * t * pow(b2, t) / bc2;
            a.p0 = rho_t > 5.0 ? (float)sqrt((rho_t - 4.0) * (rho_t - 2.0) * rho_inf / ((rho_inf - 4.0) * (rho_inf - 2.0) * rho_t)) : 0.f;
            break;
        }
        case DCV_OPT_ADADELTA:
            a.b2 = (float)d.opt_p[0];
            a.w2 = (float)(1.0 - d.opt_p[0]);
            break;
        case DCV_OPT_ASGD: {
            // eta is a float32 state tensor: written as lr / (1 + lambd lr step)^alpha after every step, read back through .item()
            const double eta = m->asgd_eta;
            a.p0 = (float)(1.0 - d.opt_p[0] * eta);
            a.p1 = (float)eta;
            m->asgd_eta = (double)(float)(m->lr / pow(1.0 + d.opt_p[0] * m->lr * t, d.opt_p[1]));
            break;
        }
        case DCV_OPT_RPROP:
            a.p0 = (float)d.opt_p[0];
            a.p1 = (float)d.opt_p[1];
            a.p2 = (float)d.opt_p[2];
            a.p3 = (float)d.opt_p[3];
            break;
        default:   // DCV_OPT_ADAGRAD
            a.c1 = (float)(m->lr / (1.0 + (t - 1.0) * d.lr_decay));
            break;
    }
    return a;
}

static int apply_impl(dcv_mlp* m, void* stream) {
    DCV_REQUIRE(m, "dcv_mlp_apply: null");
    hipStream_t s = as_stream(stream);
    const int rcf = first_step_state(m, s);
    if (rcf) return rcf;
    const OptArgs a = next_opt_args(m);
    int64_t blocks = cdiv(m->n_params, 256);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(optimizer_kernel, dim3((unsigned)blocks), dim3(256), 0, s, m->params, (const float*)m->grads, m->adam_m, m->adam_v,
                       m->opt_aux, m->n_params, a);
    DCV_CHECK_LAUNCH();
    return DCV_OK;
}

// One-GPU autoencoder step as ONE fused launch (+ the gradient reduction with the optimiser update) when the network
// fits in LDS (snet.hip); 1 = not applicable: the caller runs the layer-by-layer path.
static int snet_step(dcv_mlp* m, const float* Xn_d, int64_t ld, const int64_t* idx_d, int64_t row0, int32_t batch, int32_t train, void* stream) {
    if (m->desc.model != DCV_MODEL_AE || m->any_drop || m->any_bn || (m->snet_tried && m->snet == nullptr)) return 1;
    if (!(Xn_d && batch >= 1 && batch <= m->desc.max_batch && ld >= m->desc.dims[0] && m->log && m->log_cap > 0)) return 1;   // the general path reports it
    hipStream_t s = as_stream(stream);
    const RowMap rm = RowMap{idx_d, row0, 0, 0};
    ReduceArgsView v;
    prof_mark(m, 0, 0, 0, s);   // profiling: the fused launch is reported under both layer-0 classes (forward, weight gradient)
    prof_mark(m, 0, 1, 0, s);
    int rc = snet_ae_step(m, Xn_d, ld, rm, batch, batch, train, &v, s);
    if (rc) return rc;
    prof_mark(m, 0, 0, 1, s);
    prof_mark(m, 0, 1, 1, s);
    m->fwd_train = train != 0;
    if (m->fwd_train) m->cur_step = m->drop_step++;
    m->head_done = false;
    m->last_batch = batch;
    m->last_path = 1;
    if (!train) return DCV_OK;
    ReduceArgs ra;
    ra.L = m->L;
    for (int l = 0; l < m->L; ++l) {
        const LayerPlan& p = m->layers[l];
        ReduceDesc& rd = ra.l[l];
        rd.slab = v.slab[l];
        rd.bpart = v.bpart[l];
        rd.w_off = p.w_off;
        rd.b_off = p.b_off;
        rd.w_count = (int64_t)p.out * p.in;
        rd.out = p.out;
        rd.splits = v.splits[l];
        rd.bblocks = v.bblocks[l];
        rd.w_stride = v.wstride[l];
        rd.b_stride = v.bstride[l];
    }
    const int rcf = first_step_state(m, s);
    if (rcf) return rcf;
    return launch_reduce(m, ra, 0, m->L, true, next_opt_args(m), s);
}

extern "C" int dcv_mlp_forward(dcv_mlp* m, const float* Xn_d, int64_t ld, const int64_t* idx_d, int64_t row0, int32_t batch,
                               int32_t train, void* stream) {
    DCV_REQUIRE(m, "dcv_mlp_forward: null");
    return run_graphed(m, 1, as_stream(stream), [&] { return forward_impl(m, Xn_d, ld, idx_d, row0, batch, train, stream); });
}

extern "C" int dcv_mlp_backward(dcv_mlp* m, const float* Xn_d, int64_t ld, const int64_t* idx_d, int64_t row0, int32_t batch,
                                int64_t global_batch, int32_t train, void* stream) {
    DCV_REQUIRE(m, "dcv_mlp_backward: null");
    if (!train) return backward_impl(m, Xn_d, ld, idx_d, row0, batch, global_batch, train, stream);   // one tiny launch
    return run_graphed(m, 2, as_stream(stream), [&] { return backward_impl(m, Xn_d, ld, idx_d, row0, batch, global_batch, train, stream); });
}

extern "C" int dcv_mlp_apply(dcv_mlp* m, void* stream) { return apply_impl(m, stream); }
extern "C" int64_t dcv_mlp_graph_launches(const dcv_mlp* m) { return m ? m->graph_launches : 0; }
extern "C" int dcv_mlp_set_graph(dcv_mlp* m, int32_t enable) {
    DCV_REQUIRE(m, "dcv_mlp_set_graph: null");
    m->graph_on = enable != 0;
    return DCV_OK;
}

extern "C" int dcv_mlp_train_step(dcv_mlp* m, const float* Xn_d, int64_t ld, const int64_t* idx_d, int64_t row0, int32_t batch,
                                  void* stream) {
    DCV_REQUIRE(m, "dcv_mlp_train_step: null");
    return run_graphed(m, 0, as_stream(stream), [&] {
        int rc = snet_step(m, Xn_d, ld, idx_d, row0, batch, 1, stream);
        if (rc != 1) return rc;
        rc = forward_impl(m, Xn_d, ld, idx_d, row0, batch, 1, stream, 1);
        if (rc) return rc;
        return backward_impl(m, Xn_d, ld, idx_d, row0, batch, batch, 1, stream, true);   // reduction + optimiser update in one launch
    });
}

extern "C" int dcv_mlp_eval_step(dcv_mlp* m, const float* Xn_d, int64_t ld, const int64_t* idx_d, int64_t row0, int32_t batch,
                                 void* stream) {
    DCV_REQUIRE(m, "dcv_mlp_eval_step: null");
    return run_graphed(m, 3, as_stream(stream), [&] {
        int rc = snet_step(m, Xn_d, ld, idx_d, row0, batch, 0, stream);
        if (rc != 1) return rc;
        rc = forward_impl(m, Xn_d, ld, idx_d, row0, batch, 0, stream, 2);
        if (rc) return rc;
        return backward_impl(m, Xn_d, ld, idx_d, row0, batch, batch, 0, stream);
    });
}

// nsteps consecutive training steps in one call: step j takes samples [j * batch, (j + 1) * batch) of the index list / the row
// range, exactly as nsteps calls of dcv_mlp_train_step would (same launches, same records).  What it removes is the caller's
// per-step cost: a small-network step is ~30 - 40 us of device time and a Python caller spends 10 - 30 us per call on some
// hosts (bench.py c2: 46 us per step from Python against 39 us of launches) -- the epoch loop of a fit belongs on this side
// of the boundary.
extern "C" int dcv_mlp_train_steps(dcv_mlp* m, const float* Xn_d, int64_t ld, const int64_t* idx_d, int64_t row0, int32_t batch, int32_t nsteps,
                                   void* stream) {
    DCV_REQUIRE(m && Xn_d, "dcv_mlp_train_steps: null argument");
    DCV_REQUIRE(nsteps >= 0 && batch >= 1, "dcv_mlp_train_steps: nsteps=%d batch=%d", nsteps, batch);
    for (int32_t j = 0; j < nsteps; ++j) {
        const int64_t off = (int64_t)j * batch;
        const int rc = dcv_mlp_train_step(m, Xn_d, ld, idx_d ? idx_d + off : nullptr, idx_d ? row0 : row0 + off, batch, stream);
        if (rc) return rc;
    }
    return DCV_OK;
}

// nbatches consecutive evaluation steps -- batch j = samples [j * batch, (j + 1) * batch) of the index list (idx_d + j * batch)
// or of the row range (row0 + j * batch) -- with one loss record each, in batch order: the records dcv_mlp_eval_step would
// append one call at a time.  A small network (snet.hip / snet_dt.hip) evaluates up to kEvalBatchesPerLaunch batches per
// launch -- a validation pass is then one launch instead of one (autoencoder) or one (Deep-TICA) per batch, each of which is
// mostly launch latency and weight staging; every other engine runs the steps one after the other.
extern "C" int dcv_mlp_eval_steps(dcv_mlp* m, const float* Xn_d, int64_t ld, const int64_t* idx_d, int64_t row0, int32_t batch, int32_t nbatches,
                                  void* stream) {
    DCV_REQUIRE(m && Xn_d, "dcv_mlp_eval_steps: null argument");
    DCV_REQUIRE(nbatches >= 0, "dcv_mlp_eval_steps: nbatches=%d", nbatches);
    DCV_REQUIRE(batch >= 1 && batch <= m->desc.max_batch, "dcv_mlp_eval_steps: batch=%d exceeds max_batch=%d", batch, m->desc.max_batch);
    DCV_REQUIRE(ld >= m->desc.dims[0], "dcv_mlp_eval_steps: ld=%lld < F=%d", (long long)ld, m->desc.dims[0]);
    DCV_REQUIRE(m->log && m->log_cap > 0, "dcv_mlp_eval_steps: call dcv_mlp_reset_log first");
    hipStream_t s = as_stream(stream);
    int32_t j = 0;
    while (j < nbatches) {
        const int64_t off = (int64_t)j * batch;
        const int64_t* idx_j = idx_d ? idx_d + off : nullptr;
        const int64_t row_j = idx_d ? row0 : row0 + off;
        int nb = nbatches - j < kEvalBatchesPerLaunch ? nbatches - j : kEvalBatchesPerLaunch;
        int rc = 1;
        if (nb > 1 && !m->any_drop && !m->any_bn && !prof_on(m, 0)) {   // (a profiled run samples single steps)
            g_launch_ev = LaunchEvents{};
            if (m->desc.model == DCV_MODEL_AE && !(m->snet_tried && m->snet == nullptr)) {
                const int tr = snet_ae_tile_rows(m, batch);
                const int64_t per = cdiv((int64_t)batch, tr > 0 ? tr : 16);   // workgroups per batch
                while (nb > 1 && per * nb > kEvalWorkgroupsPerLaunch) --nb;
                if (nb > 1) rc = snet_ae_step(m, Xn_d, ld, RowMap{idx_j, row_j, 0, 0}, batch, batch, 0, nullptr, s, true, nb);
                if (rc == DCV_OK) m->last_path = 1;
            } else if (m->desc.model == DCV_MODEL_DEEPTICA && !(m->snet_dt_tried && m->snet_dt == nullptr)) {
                const int64_t per = cdiv((int64_t)batch, 8);   // (an upper bound of the workgroups per batch: tiles of >= 8 pairs)
                while (nb > 1 && per * nb > kEvalWorkgroupsPerLaunch) --nb;
                if (nb > 1) rc = snet_dt_forward(m, Xn_d, ld, idx_j, row_j, batch, 2, false, s, nb);
                if (rc == DCV_OK) m->last_path = 2;
            }
            if (rc < 0) return rc;
            if (rc == DCV_OK) {
                m->fwd_train = false;
                m->head_done = m->desc.model == DCV_MODEL_DEEPTICA;
                m->snet_fwd_valid = false;
                m->last_batch = batch;
                j += nb;
                continue;
            }
        }
        rc = dcv_mlp_eval_step(m, Xn_d, ld, idx_j, row_j, batch, stream);
        if (rc) return rc;
        j += 1;
    }
    return DCV_OK;
}

// ---- data-parallel step: the whole sequence behind one entry point, the collectives through a host callback
namespace {
struct DpTrampoline {
    dcv_mlp* m;
    dcv_allreduce_fn fn;
    void* user;
    int rc;
};
void dp_upper_cb(void* p) {
    DpTrampoline* t = static_cast<DpTrampoline*>(p);
    const int64_t off = t->m->layers[1].w_off;
    t->rc = t->fn(t->user, t->m->grads + off, t->m->n_params - off, DCV_DTYPE_F32, DCV_DP_UPPER_START);
}
}  // namespace

// Data-parallel autoencoder step through the fused small-network launch (snet.hip): the loss gradient of a row needs nothing
// from the other ranks (the scale 2 / (global batch * F) is known up front), so the whole local forward + backward is the
// one launch of the single-GPU step; then the squared-error sum is all-reduced and logged, the gradient partials are
// reduced, the gradient buffer all-reduced, the update applied.  1 = not applicable.
static int dp_ae_fused(dcv_mlp* m, const float* Xn_d, int64_t ld, const int64_t* idx_d, int64_t row0, int32_t batch, int64_t global_batch,
                       int32_t train, dcv_allreduce_fn fn, void* user, void* stream) {
    if (m->desc.model != DCV_MODEL_AE || m->any_drop || m->any_bn || (m->snet_tried && m->snet == nullptr)) return 1;
    if (!(Xn_d && batch >= 1 && batch <= m->desc.max_batch && ld >= m->desc.dims[0] && m->log && m->log_cap > 0 && global_batch >= batch)) return 1;
    hipStream_t s = as_stream(stream);
    g_launch_ev = LaunchEvents{};
    ReduceArgsView v;
    prof_mark(m, 0, 0, 0, s);
    prof_mark(m, 0, 1, 0, s);
    int rc = snet_ae_step(m, Xn_d, ld, RowMap{idx_d, row0, 0, 0}, batch, global_batch, train, &v, s, false);
    if (rc) return rc;
    prof_mark(m, 0, 0, 1, s);
    prof_mark(m, 0, 1, 1, s);
    m->fwd_train = train != 0;
    if (m->fwd_train) m->cur_step = m->drop_step++;
    m->head_done = false;
    m->last_batch = batch;
    m->last_path = 1;
    if (fn(user, m->stats, m->stats_len, DCV_DTYPE_F64, DCV_DP_STATS) != 0) {
        set_error("dcv_mlp_dp_step: the all-reduce callback failed (statistics)");
        return DCV_ECALLBACK;
    }
    hipLaunchKernelGGL(ae_log_kernel, dim3(1), dim3(64), 0, s, m->stats, (double)global_batch, m->desc.dims[0], m->log, m->log_count, m->log_cap,
                       m->log_width);
    DCV_CHECK_LAUNCH();
    if (!train) return DCV_OK;
    ReduceArgs ra;
    ra.L = m->L;
    for (int l = 0; l < m->L; ++l) {
        const LayerPlan& p = m->layers[l];
        ReduceDesc& rd = ra.l[l];
        rd.slab = v.slab[l];
        rd.bpart = v.bpart[l];
        rd.w_off = p.w_off;
        rd.b_off = p.b_off;
        rd.w_count = (int64_t)p.out * p.in;
        rd.out = p.out;
        rd.splits = v.splits[l];
        rd.bblocks = v.bblocks[l];
        rd.w_stride = v.wstride[l];
        rd.b_stride = v.bstride[l];
    }
    rc = launch_reduce(m, ra, 0, m->L, false, OptArgs{}, s);
    if (rc) return rc;
    if (fn(user, m->grads, m->n_params, DCV_DTYPE_F32, DCV_DP_GRADS) != 0) {
        set_error("dcv_mlp_dp_step: the all-reduce callback failed (gradients)");
        return DCV_ECALLBACK;
    }
    return apply_impl(m, stream);
}

extern "C" int dcv_mlp_dp_step(dcv_mlp* m, const float* Xn_d, int64_t ld, const int64_t* idx_d, int64_t row0, int32_t batch,
                               int64_t global_batch, int32_t train, int32_t overlap, dcv_allreduce_fn fn, void* user, void* stream) {
    DCV_REQUIRE(m && fn, "dcv_mlp_dp_step: null argument");
    {
        const int rcf = dp_ae_fused(m, Xn_d, ld, idx_d, row0, batch, global_batch, train, fn, user, stream);
        if (rcf != 1) return rcf;
    }
    // Batch normalisation normalises with the rows of ONE forward call: in a frame-sharded step that would be each rank's
    // local rows, the running statistics would drift apart between the ranks, and N ranks would no longer equal one process
    // on the union batch (which every other part of this step guarantees).  Refused rather than silently different.
    if (m->any_bn && global_batch != batch) {
        set_error("dcv_mlp_dp_step: batch normalisation is not implemented for data-parallel fits (global batch %lld != local batch %d: "
                  "the normalisation would use each rank's local rows only); fit on one GPU or drop `batchnorm`", (long long)global_batch, batch);
        return DCV_EINVAL;
    }
    // whatever fails from here on, a reduction started with DCV_DP_UPPER_START must be joined before the caller gets the
    // gradient buffer back: every exit below goes through fail()
    bool started = false;
    auto fail = [&](int rc) {
        if (started) (void)fn(user, nullptr, 0, DCV_DTYPE_F32, DCV_DP_WAIT);
        return rc;
    };
    int rc = forward_impl(m, Xn_d, ld, idx_d, row0, batch, train, stream, 0);
    if (rc) return rc;
    if (fn(user, m->stats, m->stats_len, DCV_DTYPE_F64, DCV_DP_STATS) != 0) {
        set_error("dcv_mlp_dp_step: the all-reduce callback failed (statistics)");
        return DCV_ECALLBACK;
    }
    if (!train) return backward_impl(m, Xn_d, ld, idx_d, row0, batch, global_batch, 0, stream);
    if (overlap && m->L > 1) {
        // the gradients of layers 1.. are reduced and handed to the callback before the layer-0 weight gradient is
        // enqueued (dcv_mlp_set_upper_grads_callback): their exchange runs under the largest product of the step
        DpTrampoline t{m, fn, user, 0};
        void (*keep_cb)(void*) = m->upper_cb;
        void* keep_user = m->upper_cb_user;
        m->upper_cb = dp_upper_cb;
        m->upper_cb_user = &t;
        started = true;   // (the trampoline may have started the exchange even when the backward fails behind it)
        rc = backward_impl(m, Xn_d, ld, idx_d, row0, batch, global_batch, 1, stream);
        m->upper_cb = keep_cb;
        m->upper_cb_user = keep_user;
        if (rc) return fail(rc);
        if (t.rc != 0 || fn(user, m->grads, m->layers[1].w_off, DCV_DTYPE_F32, DCV_DP_GRADS) != 0) {
            set_error("dcv_mlp_dp_step: the all-reduce callback failed (gradients)");
            return fail(DCV_ECALLBACK);
        }
        started = false;
        if (fn(user, nullptr, 0, DCV_DTYPE_F32, DCV_DP_WAIT) != 0) {
            set_error("dcv_mlp_dp_step: the all-reduce callback failed (join)");
            return DCV_ECALLBACK;
        }
    } else {
        void (*keep_cb)(void*) = m->upper_cb;
        m->upper_cb = nullptr;
        rc = backward_impl(m, Xn_d, ld, idx_d, row0, batch, global_batch, 1, stream);
        m->upper_cb = keep_cb;
        if (rc) return rc;
        if (fn(user, m->grads, m->n_params, DCV_DTYPE_F32, DCV_DP_GRADS) != 0) {
            set_error("dcv_mlp_dp_step: the all-reduce callback failed (gradients)");
            return DCV_ECALLBACK;
        }
    }
    return apply_impl(m, stream);
}

extern "C" int dcv_mlp_infer(dcv_mlp* m, const float* Xn_d, int64_t n, int64_t ld, const float* tmean_d, const float* tevecs_d,
                             const float* pmean_d, const float* prange_d, float* out_d, float* minmax_d, void* stream) {
    DCV_REQUIRE(m && Xn_d && n >= 1, "dcv_mlp_infer: bad arguments");
    DCV_REQUIRE(n <= m->rows_cap, "dcv_mlp_infer: n=%lld exceeds the row capacity %lld (chunk the call)", (long long)n, (long long)m->rows_cap);
    DCV_REQUIRE((tmean_d == nullptr) == (tevecs_d == nullptr), "dcv_mlp_infer: tmean/tevecs must come together");
    hipStream_t s = as_stream(stream);
    const int n_run = m->desc.model == DCV_MODEL_AE ? m->desc.latent_layer : m->L;
    const RowMap rm = identity_rows();
    m->fwd_train = false;
    int rc = run_forward(m, Xn_d, ld, rm, n, n_run, s);
    if (rc) return rc;
    const LayerPlan& last = m->layers[n_run - 1];
    const int d = last.out;
    // y = (h - tmean) @ tevecs ; out = (y - pmean) / prange   -- the linear projection kernel
    return dcv_project_linear(layer_out(m, n_run - 1), n, d, last.ldh, tmean_d ? tmean_d : m->zeros_d, m->ones_d, tevecs_d ? tevecs_d : m->ident, d,
                              nullptr, pmean_d, prange_d, out_d, minmax_d, m->proj_ws, m->proj_ws_bytes, stream);
}

// Input-gradient pass of the sensitivity analysis (reference cv_calculator.py:1893-1921 ->
// mlcolvar.explain.sensitivity_analysis, metric "mean_abs_val"): for the rows given,
// sens[i] = sum_r | d(sum_j cv_j)/d xn[r][i] | * scale[i], where the layers behind the network (TICA,
// post-normalisation) enter through the constant vector g = d(sum_j cv_j)/d(network output).
extern "C" size_t dcv_mlp_input_sensitivity_workspace(const dcv_mlp* m, int64_t n) {
    if (!m || n < 1) return 0;
    const size_t F = (size_t)m->desc.dims[0];
    return align_up((size_t)n * F * sizeof(float), 256) + (size_t)cdiv(n, kAbsRows) * F * sizeof(double);
}

extern "C" int dcv_mlp_input_sensitivity(dcv_mlp* m, const float* Xn_d, int64_t n, int64_t ld, const float* gout_d,
                                         const float* scale_d, double* sens_d, void* workspace, size_t workspace_bytes,
                                         void* stream) {
    DCV_REQUIRE(m && Xn_d && gout_d && scale_d && sens_d && workspace, "dcv_mlp_input_sensitivity: null argument");
    DCV_REQUIRE(n >= 1 && n <= m->rows_cap, "dcv_mlp_input_sensitivity: n=%lld outside [1, %lld] (chunk the call)", (long long)n,
                (long long)m->rows_cap);
    DCV_REQUIRE(workspace_bytes >= dcv_mlp_input_sensitivity_workspace(m, n), "dcv_mlp_input_sensitivity: workspace too small");
    hipStream_t s = as_stream(stream);
    const int F = m->desc.dims[0];
    DCV_REQUIRE(ld >= F, "dcv_mlp_input_sensitivity: ld=%lld < F=%d", (long long)ld, F);
    const int n_run = m->desc.model == DCV_MODEL_AE ? m->desc.latent_layer : m->L;
    m->fwd_train = false;
    int rc = run_forward(m, Xn_d, ld, identity_rows(), n, n_run, s);
    if (rc) return rc;
    const LayerPlan& last = m->layers[n_run - 1];
    float* dz_cur = m->dZ[0];
    float* dz_nxt = m->dZ[1];
    hipLaunchKernelGGL(seed_grad_kernel, dim3((unsigned)cdiv(n * last.out, 256)), dim3(256), 0, s, (const float*)layer_out(m, n_run - 1), last.ldh, n,
                       last.out, last.bn ? DCV_ACT_NONE : last.act, gout_d, dz_cur, m->ld_dz);
    DCV_CHECK_LAUNCH();
    if (last.bn) {   // evaluation-mode normalisation: a per-column scale, then the activation derivative of the Linear underneath
        rc = bn_eval_backward(m, n_run - 1, dz_cur, m->ld_dz, n, last.act, s);
        if (rc) return rc;
    }
    for (int l = n_run - 1; l >= 1; --l) {   // dZ_{l-1} = (dZ_l W_l) * act'(H_{l-1})
        LayerPlan& p = m->layers[l];
        LayerPlan& q = m->layers[l - 1];
        Operand Ad = make_operand(dz_cur, m->ld_dz, p.out);
        Operand Bd = make_operand(m->params + p.w_off, p.in, p.in);
        EpiActGrad eg{dz_nxt, m->ld_dz, q.H, q.ldh, q.bn ? DCV_ACT_NONE : q.act, q.bpart, q.out, quad_ok(dz_nxt, m->ld_dz) && quad_ok(q.H, q.ldh)};
        int bblocks = 0;
        rc = gemm_nn_act_grad(Ad, Bd, n, p.in, p.out, eg, s, &bblocks, nullptr);
        if (rc) return rc;
        if (q.bn) {
            rc = bn_eval_backward(m, l - 1, dz_nxt, m->ld_dz, n, q.act, s);
            if (rc) return rc;
        }
        float* tmp = dz_cur;
        dz_cur = dz_nxt;
        dz_nxt = tmp;
    }
    float* G = static_cast<float*>(workspace);
    double* part = reinterpret_cast<double*>(static_cast<char*>(workspace) + align_up((size_t)n * F * sizeof(float), 256));
    {   // dXn = dZ_0 W_0
        const LayerPlan& p = m->layers[0];
        Operand Ad = make_operand(dz_cur, m->ld_dz, p.out);
        Operand Bd = make_operand(m->params + p.w_off, p.in, p.in);
        EpiStore es{G, F, quad_ok(G, F)};
        rc = gemm_nn_store(Ad, Bd, n, p.in, p.out, es, s);
        if (rc) return rc;
    }
    const int nb = (int)cdiv(n, kAbsRows);
    hipLaunchKernelGGL(abs_colsum_kernel, dim3(nb), dim3(256), 0, s, G, (int64_t)F, n, F, scale_d, part);
    DCV_CHECK_LAUNCH();
    hipLaunchKernelGGL(sum_partials_kernel, dim3(F), dim3(64), 0, s, part, nb, F, sens_d);
    DCV_CHECK_LAUNCH();
    return DCV_OK;
}
