// torch.optim's single-tensor updates as a device function (one thread per parameter), shared by the optimiser / reduction
// kernels of mlp.hip and the in-launch gradient reduction of the fused small-network kernels (snet.h: snet_reduce_update).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include "common.h"
#include "handoff.h"

namespace dcv {

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
// Arguments of the in-launch gradient reduction + update of the fused small-network kernels (snet.h: snet_reduce_update)
struct SnetReduce {
    int on;
    float* grads;
    float* params;
    float *s1, *s2, *s3;
    OptArgs oa;
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
}  // namespace dcv
