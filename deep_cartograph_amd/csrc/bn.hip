// torch.nn.BatchNorm1d behind a Linear of the MLP engine (mlcolvar FeedForward order: Linear, activation, dropout,
// batchnorm; reference option `batchnorm` / `last_layer_batchnorm`, cv_calculator.py:1155-1219).
//
// Training forward: per-column mean and biased variance of the layer's (post-activation, post-dropout) output A over the
// rows of one forward call -- a Deep-TICA batch is two calls, x_t then x_lag, each with its own statistics and its own
// running-statistics update -- then y = (a - mean) * invstd * weight + bias into a second buffer (A stays: the
// backward pass needs the normalised values and the activation derivative).  Evaluation / inference: the running
// statistics.  Backward: dA = weight * invstd / n * (n dY - sum dY - xhat sum(dY xhat)), then the activation derivative
// and dropout of the Linear underneath; the partial sums of dY and dY * xhat per row block are at once the gradient
// partials of bias and weight for the split-K reduction (reduce_grads_*_kernel).
//
// These are streaming passes in the plain layout (a thread per column, rows strided over the block): batch
// normalisation is an option of the reference no bundled configuration uses; the kernels are correct and coalesced,
// not tuned.
#include "mlp_state.h"

namespace dcv {

constexpr int kBnRows = 256;   // rows per block of the statistics passes

// thread layout of a 256-thread block over `out` columns: cw columns x (256 / cw) row lanes
__device__ __forceinline__ void bn_layout(int out, int& cw, int& lanes) {
    cw = 1;
    while (cw < out && cw < 256) cw <<= 1;
    lanes = 256 / cw;
}

// part[block][0][c] = sum a, part[block][1][c] = sum a^2 over the block's rows (float64)
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ A, int64_t ld, int64_t row0, int64_t rows, int out,
                                                       double* __restrict__ part) {
    __shared__ double red[2][256];
    int cw, lanes;
    bn_layout(out, cw, lanes);
    const int t = threadIdx.x, tx = t % cw, ty = t / cw;
    const int64_t r0 = row0 + (int64_t)blockIdx.x * kBnRows;
    const int64_t r1 = r0 + kBnRows < row0 + rows ? r0 + kBnRows : row0 + rows;
    for (int c0 = 0; c0 < out; c0 += cw) {
        const int c = c0 + tx;
        double s = 0.0, ss = 0.0;
        if (c < out)
            for (int64_t r = r0 + ty; r < r1; r += lanes) {
                const double v = (double)A[r * ld + c];
                s += v;
                ss = fma(v, v, ss);
            }
        red[0][t] = s;
        red[1][t] = ss;
        __syncthreads();
        if (ty == 0 && c < out) {
            double a = 0.0, b = 0.0;
            for (int q = 0; q < lanes; ++q) {
                a += red[0][q * cw + tx];
                b += red[1][q * cw + tx];
            }
            part[((int64_t)blockIdx.x * 2 + 0) * out + c] = a;
            part[((int64_t)blockIdx.x * 2 + 1) * out + c] = b;
        }
        __syncthreads();
    }
}

// stat[0][c] = mean, stat[1][c] = 1 / sqrt(biased var + eps); running statistics as torch updates them
// (running = (1 - momentum) * running + momentum * batch value, the variance unbiased)
__global__ void bn_finalize_kernel(const double* __restrict__ part, int blocks, int out, double n, double eps, double momentum,
                                   float* __restrict__ rm, float* __restrict__ rv, double* __restrict__ stat) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= out) return;
    double s = 0.0, ss = 0.0;
    for (int b = 0; b < blocks; ++b) {
        s += part[((int64_t)b * 2 + 0) * out + c];
        ss += part[((int64_t)b * 2 + 1) * out + c];
    }
    const double mean = s / n;
    double var = ss / n - mean * mean;
    if (var < 0.0) var = 0.0;
    stat[c] = mean;
    stat[out + c] = 1.0 / sqrt(var + eps);
    const float mf = (float)momentum;
    const float unb = n > 1.0 ? (float)(var * n / (n - 1.0)) : (float)var;
    rm[c] = (1.f - mf) * rm[c] + mf * (float)mean;
    rv[c] = (1.f - mf) * rv[c] + mf * unb;
}

// Y = (A - mean) * invstd * weight + bias.  stat != null: batch statistics; else the running ones.
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ A, int64_t lda, float* __restrict__ Y, int64_t ldy, int64_t row0,
                                                       int64_t rows, int out, const double* __restrict__ stat, const float* __restrict__ rm,
                                                       const float* __restrict__ rv, float eps, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta) {
    const int64_t total = rows * out;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = row0 + i / out;
        const int c = (int)(i % out);
        const float mean = stat ? (float)stat[c] : rm[c];
        const float invstd = stat ? (float)stat[out + c] : 1.f / sqrtf(rv[c] + eps);
        Y[r * ldy + c] = (A[r * lda + c] - mean) * invstd * gamma[c] + beta[c];
    }
}

// gpart[block][c] = sum dY * xhat, bpart[block][c] = sum dY over the block's rows
__global__ __launch_bounds__(256) void bn_bwd_stats_kernel(const float* __restrict__ dY, int64_t ldd, const float* __restrict__ A, int64_t lda,
                                                           int64_t row0, int64_t rows, int out, const double* __restrict__ stat,
                                                           float* __restrict__ gpart, float* __restrict__ bpart) {
    __shared__ double red[2][256];
    int cw, lanes;
    bn_layout(out, cw, lanes);
    const int t = threadIdx.x, tx = t % cw, ty = t / cw;
    const int64_t r0 = row0 + (int64_t)blockIdx.x * kBnRows;
    const int64_t r1 = r0 + kBnRows < row0 + rows ? r0 + kBnRows : row0 + rows;
    for (int c0 = 0; c0 < out; c0 += cw) {
        const int c = c0 + tx;
        double sg = 0.0, sb = 0.0;
        if (c < out) {
            const float mean = (float)stat[c], invstd = (float)stat[out + c];
            for (int64_t r = r0 + ty; r < r1; r += lanes) {
                const float d = dY[r * ldd + c];
                const float xh = (A[r * lda + c] - mean) * invstd;
                sg += (double)d * (double)xh;
                sb += (double)d;
            }
        }
        red[0][t] = sg;
        red[1][t] = sb;
        __syncthreads();
        if (ty == 0 && c < out) {
            double a = 0.0, b = 0.0;
            for (int q = 0; q < lanes; ++q) {
                a += red[0][q * cw + tx];
                b += red[1][q * cw + tx];
            }
            gpart[(int64_t)blockIdx.x * out + c] = (float)a;
            bpart[(int64_t)blockIdx.x * out + c] = (float)b;
        }
        __syncthreads();
    }
}

// dZ = weight * invstd * (dY - (sum dY + xhat * sum(dY xhat)) / n) * k * act'(a): in place over dY;
// dbpart[block][c] = column sums of dZ over the block's rows (bias gradient partials of the Linear underneath).
// The totals come from the `blocks` partial rows of THIS forward call (one half of a Deep-TICA batch).
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(float* __restrict__ dY, int64_t ldd, const float* __restrict__ A, int64_t lda,
                                                           int64_t row0, int64_t rows, int out, const double* __restrict__ stat,
                                                           const float* __restrict__ gamma, const float* __restrict__ gpart,
                                                           const float* __restrict__ bpart, int blocks, double n, int act, float hscale,
                                                           DropCfg drop, float* __restrict__ dbpart) {
    __shared__ double red[256];
    int cw, lanes;
    bn_layout(out, cw, lanes);
    const int t = threadIdx.x, tx = t % cw, ty = t / cw;
    const int64_t r0 = row0 + (int64_t)blockIdx.x * kBnRows;
    const int64_t r1 = r0 + kBnRows < row0 + rows ? r0 + kBnRows : row0 + rows;
    for (int c0 = 0; c0 < out; c0 += cw) {
        const int c = c0 + tx;
        double sdz = 0.0;
        if (c < out) {
            double Sg = 0.0, Sb = 0.0;
            for (int b = 0; b < blocks; ++b) {
                Sg += (double)gpart[(int64_t)b * out + c];
                Sb += (double)bpart[(int64_t)b * out + c];
            }
            const float mean = (float)stat[c], invstd = (float)stat[out + c];
            const float k = gamma[c] * invstd, mb = (float)(Sb / n), mg = (float)(Sg / n);
            for (int64_t r = r0 + ty; r < r1; r += lanes) {
                const float a = A[r * lda + c];
                const float xh = (a - mean) * invstd;
                float dz = k * (dY[r * ldd + c] - mb - xh * mg) * act_grad_from_out(act, a * hscale);
                if (drop.thr != 0u) dz *= f4c(drop.mult(r, c & ~3), c & 3);
                dY[r * ldd + c] = dz;
                sdz += (double)dz;
            }
        }
        red[t] = sdz;
        __syncthreads();
        if (ty == 0 && c < out) {
            double a = 0.0;
            for (int q = 0; q < lanes; ++q) a += red[q * cw + tx];
            dbpart[(int64_t)blockIdx.x * out + c] = (float)a;
        }
        __syncthreads();
    }
}

// evaluation-mode backward (input sensitivity): dZ = dY * weight / sqrt(running var + eps) * act'(a), in place
__global__ __launch_bounds__(256) void bn_eval_bwd_kernel(float* __restrict__ dY, int64_t ldd, const float* __restrict__ A, int64_t lda,
                                                          int64_t rows, int out, const float* __restrict__ rv, float eps,
                                                          const float* __restrict__ gamma, int act) {
    const int64_t total = rows * out;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / out;
        const int c = (int)(i % out);
        dY[r * ldd + c] *= gamma[c] / sqrtf(rv[c] + eps) * act_grad_from_out(act, A[r * lda + c]);
    }
}

static unsigned bn_ew_blocks(int64_t total) {
    int64_t b = cdiv(total, 256);
    const int64_t cap = (int64_t)num_cus() * 16;
    return (unsigned)(b > cap ? cap : (b < 1 ? 1 : b));
}

// ------------------------------------------------------------------ host entry points (used by mlp.hip)
// forward of the batch normalisation behind layer l over rows [row0, row0 + rows) of its activation buffer
int bn_forward(dcv_mlp* m, int l, int64_t row0, int64_t rows, bool train, hipStream_t s) {
    LayerPlan& p = m->layers[l];
    const float* gamma = m->params + p.g_off;
    const float* beta = m->params + p.be_off;
    if (train) {
        const int blocks = (int)cdiv(rows, kBnRows);
        double* stat = p.bn_stat + (row0 > 0 ? 2 * p.out : 0);   // second forward call of the step (x_lag) keeps its own statistics
        hipLaunchKernelGGL(bn_stats_kernel, dim3(blocks), dim3(256), 0, s, (const float*)p.H, p.ldh, row0, rows, p.out, p.bn_part);
        DCV_CHECK_LAUNCH();
        hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)cdiv(p.out, 64)), dim3(64), 0, s, (const double*)p.bn_part, blocks, p.out,
                           (double)rows, m->desc.bn_eps, m->desc.bn_momentum, p.rm, p.rv, stat);
        DCV_CHECK_LAUNCH();
        hipLaunchKernelGGL(bn_apply_kernel, dim3(bn_ew_blocks(rows * p.out)), dim3(256), 0, s, (const float*)p.H, p.ldh, p.Y, p.ldh, row0, rows,
                           p.out, (const double*)stat, (const float*)nullptr, (const float*)nullptr, 0.f, gamma, beta);
        DCV_CHECK_LAUNCH();
        p.bn_batches += 1;
    } else {
        hipLaunchKernelGGL(bn_apply_kernel, dim3(bn_ew_blocks(rows * p.out)), dim3(256), 0, s, (const float*)p.H, p.ldh, p.Y, p.ldh, row0, rows,
                           p.out, (const double*)nullptr, (const float*)p.rm, (const float*)p.rv, (float)m->desc.bn_eps, gamma, beta);
        DCV_CHECK_LAUNCH();
    }
    return DCV_OK;
}

// Backward through the batch normalisation AND the activation / dropout of layer l: dz holds dL/dY on entry and dL/dz
// (pre-activation gradient of the Linear) on return.  halves: 1, or 2 forward calls of `rows_half` rows each.
// Leaves the gradient partials of weight / bias of the normalisation in p.bn_gpart / p.bn_bpart and the bias-gradient
// partials of the Linear in p.bpart; *blocks_out = number of partial rows of each.
int bn_backward(dcv_mlp* m, int l, float* dz, int64_t ld_dz, int halves, int64_t rows_half, int act, float hscale, const DropCfg& drop,
                int* blocks_out, hipStream_t s) {
    LayerPlan& p = m->layers[l];
    const int bh = (int)cdiv(rows_half, kBnRows);
    for (int h = 0; h < halves; ++h) {
        const int64_t row0 = (int64_t)h * rows_half;
        const double* stat = p.bn_stat + (h > 0 ? 2 * p.out : 0);
        float* gp = p.bn_gpart + (int64_t)h * bh * p.out;
        float* bp = p.bn_bpart + (int64_t)h * bh * p.out;
        hipLaunchKernelGGL(bn_bwd_stats_kernel, dim3(bh), dim3(256), 0, s, (const float*)dz, ld_dz, (const float*)p.H, p.ldh, row0, rows_half, p.out,
                           stat, gp, bp);
        DCV_CHECK_LAUNCH();
        hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(bh), dim3(256), 0, s, dz, ld_dz, (const float*)p.H, p.ldh, row0, rows_half, p.out, stat,
                           (const float*)(m->params + p.g_off), (const float*)gp, (const float*)bp, bh, (double)rows_half, act, hscale, drop,
                           p.bpart + (int64_t)h * bh * p.out);
        DCV_CHECK_LAUNCH();
    }
    *blocks_out = halves * bh;
    return DCV_OK;
}

int bn_eval_backward(dcv_mlp* m, int l, float* dz, int64_t ld_dz, int64_t rows, int act, hipStream_t s) {
    LayerPlan& p = m->layers[l];
    hipLaunchKernelGGL(bn_eval_bwd_kernel, dim3(bn_ew_blocks(rows * p.out)), dim3(256), 0, s, dz, ld_dz, (const float*)p.H, p.ldh, rows, p.out,
                       (const float*)p.rv, (float)m->desc.bn_eps, (const float*)(m->params + p.g_off), act);
    DCV_CHECK_LAUNCH();
    return DCV_OK;
}

}  // namespace dcv
