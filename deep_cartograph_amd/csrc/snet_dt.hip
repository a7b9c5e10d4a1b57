// Fused Deep-TICA step of a SMALL network (the reference's own sizes: cv_calculator.py:2569-2590 builds 54-16-8-2 in its
// test configuration, F-15-15-d from tools/train_colvars/default_config.yml): every weight resident in one CU's LDS, as in
// snet.hip -- but the Deep-TICA loss couples ALL samples of the batch (the d x d covariances of the network outputs), so
// one grid-wide dependency sits between the forward and the backward pass.  A kernel boundary is the cheapest grid-wide
// synchronisation on this part: the step is TWO fused launches plus the gradient reduction,
//
//   snet_dt_fwd_kernel   a workgroup takes TR / 2 PAIRS: rows [0, TR/2) of its tile are the x_t rows, rows [TR/2, TR) the
//                        x_lag rows of the same pairs (gathered through the RowMap: contiguous, gathered and two-half
//                        batches alike) -- both halves of a pair in ONE tile, so the 2d + 2d^2 batch statistics of its
//                        pairs are formed from LDS.  Forward chain with the activations kept in LDS; statistics partial
//                        (float64, pair order); the tile's activations leave as one contiguous blob (write-through); the
//                        ticketed last arriver (handoff.h) adds the partials in block order and -- one-GPU steps -- runs
//                        the wave-parallel d x d loss head (tica_head.h).  An evaluation step ends here.
//   snet_dt_bwd_kernel   stages the weights of layers >= 1 (the input gradients need them) and its blob back into LDS,
//                        evaluates the loss gradient of its pairs in float64 from the head's matrices (the arithmetic of
//                        tica_dF_kernel / head_backward_kernel), walks back through the layers as snet_ae_kernel does and
//                        leaves one gradient partial per workgroup and layer for reduce_grads_quad_kernel + the optimiser.
//
// Three launches instead of eight for a 3-layer network, none of them a matrix product over a few thousand rows on 516
// workgroups.  Row sharing is given up: a contiguous batch evaluates 2 B rows instead of B + lag -- at these
// widths the step is latency, not flops.  A data-parallel step uses the same two kernels with the head left to
// tica_grad_wave_kernel behind the statistics all-reduce.  Not taken (the layer-by-layer path runs): d > 4, dropout, batch
// normalisation, a width above 256, a network that does not fit in LDS, more than 512 tiles (DCV_NO_SNET=1 forces it off).
#include "snet.h"
#include "tica_head.h"
#include <new>

namespace dcv {

struct SnetDtArgs {
    SnetLayer l[DCV_MAX_LAYERS];
    int L;
    const int2* stage_tab;        // staging table (snet_layout), entries ordered by layer
    int stage_n;                  // all entries (forward)
    int stage_bwd0;               // first entry of layer 1: the backward kernel stages [stage_bwd0, stage_n)
    int lh[DCV_MAX_LAYERS + 1];   // LDS float offset of H_l [TR][ps_l]   (H_0 = the input tile)
    int ps[DCV_MAX_LAYERS + 1];
    int act_len;                  // floats of the activation region [lh[0], lh[0] + act_len): the blob of a workgroup
    const float* params;
    const float* img;             // global weight image in the LDS layout (snet.hip: snet_image_build), or null: table-driven staging
    int img_floats;               //   floats of the whole image; the backward stages [img_bwd0, img_floats): layers >= 1
    int img_bwd0;
    const float* Xn;
    int64_t ld;
    RowMap rows;                  // half = batch: logical row p < B is x_t of pair p, row B + p its x_lag
    int B;                        // pairs of this rank's batch
    int nb, wgpb;                 // batches of this launch (> 1: evaluation with the fused head only) and workgroups per batch
    int d;                        // network outputs (<= 4)
    int store_blob;
    float* blob;                  // [workgroups][blob_stride]
    int64_t blob_stride;
    double* spart;                // statistics partials [workgroups][2d + 2d^2]
    unsigned* ticket;
    double* stats;
    FusedHead fused;
    const double* gradp;          // backward: [mu d | Gu d*d | Gv d*d | c d]
    float* part;                  // backward: gradient partials
};

template <int NARGS>
__device__ __forceinline__ unsigned touch_kernargs() {
    unsigned x = 0;
    const __attribute__((address_space(4))) unsigned* kp = (const __attribute__((address_space(4))) unsigned*)__builtin_amdgcn_kernarg_segment_ptr();
#pragma unroll
    for (int off = 0; off < NARGS; off += 64) x ^= kp[off / 4];
    return x;
}

union TicaWaveLdsAny {
    TicaWaveLds<1> h1;
    TicaWaveLds<2> h2;
    TicaWaveLds<3> h3;
    TicaWaveLds<4> h4;
};
// TR rows per workgroup = TR / 2 pairs (16 for small batches, 32, 64 or 128: larger tiles mean fewer statistics / gradient partials for the
// ticketed sums and the reduction launch behind; the work of a tile is latency, not arithmetic, at these widths)
template <int TR>
__global__ __launch_bounds__(kSnetThreads) void snet_dt_fwd_kernel(SnetDtArgs a) {
    constexpr int NT = kSnetThreads, HP = TR / 2;
    constexpr int RG = TR / 16, CG = kSnetWaves / RG;
    constexpr int XU = TR / 8;   // 16-byte units of the input tile per thread (TR * pin / 4 <= XU * NT: pin <= 256 * 32 / TR ... checked by the plan)
    const int D = a.d, W = 2 * D + 2 * D * D;
    extern __shared__ __attribute__((aligned(16))) float sl[];
    __shared__ TicaWaveLdsAny s_head;
    __shared__ double s_stat[40];
    __shared__ unsigned s_flag;
    __shared__ int s_slot;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int rg = wave % RG, cg = wave / RG;
    const int q = lane >> 4, n = lane & 15;
    const int L = a.L;
    const unsigned ka_touch = touch_kernargs<(int)sizeof(SnetDtArgs)>();
    // ---- input tile: local row r is pair p0 + r % HP, half r / HP
    // workgroup -> (batch of the launch, tile of the batch); batch j of a batched evaluation = the pairs [j * B, (j + 1) * B)
    const int bj = a.nb > 1 ? (int)blockIdx.x / a.wgpb : 0;
    const unsigned wgpb = a.nb > 1 ? (unsigned)a.wgpb : gridDim.x, wg0 = (unsigned)bj * wgpb;
    const int64_t p0 = (int64_t)(blockIdx.x - wg0) * HP;
    RowMap rows = a.rows;
    if (bj != 0) {
        if (rows.idx != nullptr) rows.idx += (int64_t)bj * a.B;
        rows.row0 += (int64_t)bj * a.B;
    }
    const int F0 = a.l[0].in, pin0 = a.l[0].pin, ps0 = a.ps[0];
    float* H0 = sl + a.lh[0];
    const bool x_vec = (F0 & 3) == 0 && (a.ld & 3) == 0 && (reinterpret_cast<uintptr_t>(a.Xn) & 15) == 0;
    const int x_sh = a.l[0].c4_shift, x_tot = TR << x_sh;
    auto src_row = [&](int r) -> int64_t {   // matrix row of local row r, -1 past the batch
        const int half = r >= HP ? 1 : 0;
        const int64_t p = p0 + (r - half * HP);
        return p < a.B ? rows.template get<true>(half ? (int64_t)a.B + p : p) : -1;
    };
    constexpr int XA = XU < 8 ? XU : 8;   // units that ride along the weight staging; the rest (TR = 128) in a second batch
    float4 xv[XA];
    bool x_issued = false;
    auto load_x = [&](int u) -> float4 {
        const int f4 = F0 >> 2;
        const int i = t + NT * u, r = i >> x_sh, c = i - (r << x_sh);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < x_tot && c < f4) {
            const int64_t row = src_row(r);
            if (row >= 0) v = *reinterpret_cast<const float4*>(a.Xn + row * a.ld + 4 * c);
        }
        return v;
    };
    auto store_x = [&](int u, const float4& v) {
        const int i = t + NT * u, r = i >> x_sh, c = i - (r << x_sh);
        if (i < x_tot) *reinterpret_cast<float4*>(H0 + r * ps0 + 4 * c) = v;
    };
    auto issue_x = [&]() {
#pragma unroll
        for (int u = 0; u < XA; ++u) xv[u] = load_x(u);
    };
    // ---- stage every weight image and bias (snet.hip: one flat table, twelve loads in flight, two dependent round trips)
    if (a.img != nullptr) {   // one contiguous LDS-DMA copy of the weight image (kept current by the optimiser), the input rows behind it
        snet_stage_image<NT>(a.img, sl, 0, a.img_floats, t);
    } else {
        // the table entries of pass p + 1 are requested behind the data loads of pass p and arrive in the same round trip: one
        // dependent round trip per pass (+ the first table read) instead of two (round 4: 8 -> 5 for the C2 network's four passes)
        int2 e[12];
    #pragma unroll
        for (int u = 0; u < 12; ++u) {
            const int i = t + NT * u;
            e[u] = i < a.stage_n ? a.stage_tab[i] : make_int2(-1, -1);
        }
        for (int i0 = t; i0 < a.stage_n; i0 += 12 * NT) {
            float4 v[12];
    #pragma unroll
            for (int u = 0; u < 12; ++u) {
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (e[u].x >= 0) {
                    const float* src = a.params + e[u].x;
                    const int nv = (e[u].y >> 20) & 7;
                    if ((e[u].y >> 24) & 1) {
                        v[u] = *reinterpret_cast<const float4*>(src);
                    } else {
                        v[u].x = src[0];
                        if (nv > 1) v[u].y = src[1];
                        if (nv > 2) v[u].z = src[2];
                        if (nv > 3) v[u].w = src[3];
                    }
                }
            }
            if (x_vec && !x_issued) {   // the rows of X ride along the data round trip
                x_issued = true;
                issue_x();
            }
            int2 en[12];
    #pragma unroll
            for (int u = 0; u < 12; ++u) {
                const int i = i0 + 12 * NT + NT * u;
                en[u] = i < a.stage_n ? a.stage_tab[i] : make_int2(-1, -1);
            }
    #pragma unroll
            for (int u = 0; u < 12; ++u)
                if (e[u].y >= 0) *reinterpret_cast<float4*>(sl + (e[u].y & 0xFFFFF)) = v[u];
    #pragma unroll
            for (int u = 0; u < 12; ++u) e[u] = en[u];
        }
    }
    if (x_vec && !x_issued) issue_x();
    if (a.img != nullptr) vm_wait<0>();   // this wave's image copies have landed (the barrier below covers the other waves)
    asm volatile("" ::"s"(ka_touch));
    if (x_vec) {
#pragma unroll
        for (int u = 0; u < XA; ++u) store_x(u, xv[u]);
        if constexpr (XU > XA) {
            float4 xw[XU - XA];
#pragma unroll
            for (int u = XA; u < XU; ++u) xw[u - XA] = load_x(u);
#pragma unroll
            for (int u = XA; u < XU; ++u) store_x(u, xw[u - XA]);
        }
    } else {
        for (int i = t; i < TR * pin0; i += NT) {
            const int r = i / pin0, c = i - r * pin0;
            float v = 0.f;
            if (c < F0) {
                const int64_t row = src_row(r);
                if (row >= 0) v = a.Xn[row * a.ld + c];
            }
            H0[r * ps0 + c] = v;
        }
    }
    __syncthreads();
    // ---- forward chain
    for (int l = 0; l < L; ++l) {
        const SnetLayer& y = a.l[l];
        const float* Hin = sl + a.lh[l];
        float* Hout = sl + a.lh[l + 1];
        const int psin = a.ps[l], pso = a.ps[l + 1];
        const float* ap = Hin + (rg * 16 + n) * psin + 4 * q;
        const float* Wl = sl + y.lw + n * y.pws + 4 * q;
#define SNET_FWD(NK)                                                                                          \
        SnetFrags<NK> A;                                                                                       \
        A.load(ap);                                                                                            \
        for (int ct = cg; ct < y.nk_out; ct += CG) {                                                           \
            const sv4f acc = snet_fwd_tile<NK>(A, Wl + ct * 16 * y.pws);                                       \
            const int col = ct * 16 + n;                                                                       \
            const float bias = sl[y.lb + col];                                                                 \
            sv4f h = snet_act4(y.act, acc + bias);                                                             \
            if (col >= y.out) h = sv4f{0.f, 0.f, 0.f, 0.f};                                                    \
            _Pragma("unroll") for (int v = 0; v < 4; ++v) Hout[(rg * 16 + 4 * q + v) * pso + col] = h[v];      \
        }
        SNET_NK_SWITCH(y.nk_in, SNET_FWD)
#undef SNET_FWD
        __syncthreads();
    }
    // ---- statistics partial of the tile's pairs: [sum f_t | sum f_lag | sum f_t f_t^T | sum f_t f_lag^T], float64, pair order
    {
        const float* FL = sl + a.lh[L];
        const int psL = a.ps[L];
        const int64_t left = (int64_t)a.B - p0;
        const int nvalid = left >= HP ? HP : (left > 0 ? (int)left : 0);
        if (t < W) {
            double s = 0.0;
            if (t < D) {
                for (int pl = 0; pl < nvalid; ++pl) s += (double)FL[pl * psL + t];
            } else if (t < 2 * D) {
                for (int pl = 0; pl < nvalid; ++pl) s += (double)FL[(HP + pl) * psL + t - D];
            } else if (t < 2 * D + D * D) {
                const int e = t - 2 * D, i = e / D, j = e - i * D;
                for (int pl = 0; pl < nvalid; ++pl) s += (double)FL[pl * psL + i] * (double)FL[pl * psL + j];
            } else {
                const int e = t - 2 * D - D * D, i = e / D, j = e - i * D;
                for (int pl = 0; pl < nvalid; ++pl) s += (double)FL[pl * psL + i] * (double)FL[(HP + pl) * psL + j];
            }
            handoff_store(a.spart + (int64_t)blockIdx.x * W + t, s);
        }
    }
    // ---- the tile's activations (input tile, every layer's output) for the backward launch: one contiguous blob
    if (a.store_blob) {
        const float* src = sl + a.lh[0];
        float* dst = a.blob + (int64_t)blockIdx.x * a.blob_stride;
        for (int i = 4 * t; i < a.act_len; i += 4 * NT) {
            const sv4f v = *reinterpret_cast<const sv4f*>(src + i);
            handoff_store16(dst + i, v);   // write-through: nothing left for the write-back at the end of the launch
        }
    }
    // ---- last workgroup: the partials in block order, then the d x d loss head
    if (!handoff_arrive_last(a.ticket + bj, wgpb, &s_flag)) return;
    {
        const int G = NT / W;
        double* s_grp = reinterpret_cast<double*>(sl);   // [G][W]: the weight images are dead
        const int g = t / W, o = t - g * W;
        if (g < G) {
            double s = 0.0;
            const double* sp = a.spart + (int64_t)wg0 * W;
            for (unsigned b0 = g; b0 < wgpb; b0 += 8 * G) {   // eight loads in flight, added in block order
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const unsigned b = b0 + (unsigned)u * G;
                    v[u] = handoff_load(sp + (int64_t)(b < wgpb ? b : b0) * W + o);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (b0 + (unsigned)u * G < wgpb) s += v[u];
            }
            s_grp[g * W + o] = s;
        }
        __syncthreads();
        if (t < W) {
            double s = 0.0;
            for (int g2 = 0; g2 < G; ++g2) s += s_grp[g2 * W + t];
            if (a.nb <= 1) a.stats[t] = s;
            s_stat[t] = s;
        }
        if (a.fused.on) {
            __syncthreads();
            if (wave == 0) {
                const FusedHead& f = a.fused;
                // a batched evaluation appends its records in batch order: the head of batch j writes record (counter + j) -- it is
                // handed a log base moved by j records and a private copy of the counter -- and the last head to finish (a second
                // ticket, taken behind its read of the counter) moves the counter by nb
                double* logp = f.log;
                int* lc = f.log_count;
                int cap = f.log_cap, slot0 = 0;
                if (a.nb > 1) {
                    slot0 = *f.log_count;
                    asm volatile("s_waitcnt vmcnt(0) ; the counter has been read" ::: "memory");
                    if (lane == 0) s_slot = slot0;
                    wave_sync_lds();
                    logp = f.log + (int64_t)bj * f.log_width;
                    lc = &s_slot;
                    cap = f.log_cap - bj;
                }
                switch (D) {
                    case 1: tica_grad_wave<1>(s_head.h1, s_stat, f.Bg, f.reg, f.gradp, logp, lc, cap, f.log_width, lane); break;
                    case 2: tica_grad_wave<2>(s_head.h2, s_stat, f.Bg, f.reg, f.gradp, logp, lc, cap, f.log_width, lane); break;
                    case 3: tica_grad_wave<3>(s_head.h3, s_stat, f.Bg, f.reg, f.gradp, logp, lc, cap, f.log_width, lane); break;
                    default: tica_grad_wave<4>(s_head.h4, s_stat, f.Bg, f.reg, f.gradp, logp, lc, cap, f.log_width, lane); break;
                }
                if (a.nb > 1 && lane == 0) {
                    const unsigned prev = __hip_atomic_fetch_add(a.ticket + a.nb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (prev == (unsigned)a.nb - 1u) {
                        __hip_atomic_store(a.ticket + a.nb, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        *f.log_count = slot0 + a.nb;
                    }
                }
            }
        }
    }
}

template <int TR>
__global__ __launch_bounds__(kSnetThreads) void snet_dt_bwd_kernel(SnetDtArgs a) {
    constexpr int NT = kSnetThreads, HP = TR / 2;
    constexpr int RG = TR / 16, CG = kSnetWaves / RG;
    const int D = a.d, NG = 2 * D + 2 * D * D;   // mu | Gu | Gv | c
    extern __shared__ __attribute__((aligned(16))) float sl[];
    __shared__ double s_g[40];
    __shared__ double s_stat[40];
    __shared__ TicaWaveLdsAny s_head;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int rg = wave % RG, cg = wave / RG;
    const int q = lane >> 4, n = lane & 15;
    const int L = a.L;
    const unsigned ka_touch = touch_kernargs<(int)sizeof(SnetDtArgs)>();
    const int64_t p0 = (int64_t)blockIdx.x * HP;
    // ---- stage the weight images of layers >= 1 (the input gradients read them; no bias, no layer 0) and the blob.
    //      Order of issue: table entries, the first eight blob units, the weight data (needs the entries: loads return in
    //      order, so waiting for the entries does not wait for the blob), LDS stores at the end.
    const float* bsrc = a.blob + (int64_t)blockIdx.x * a.blob_stride;
    float* bdst = sl + a.lh[0];
    const int n4 = a.act_len >> 2;
    float4 bv[8];
    bool blob_issued = false;
    auto issue_blob = [&]() {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = t + NT * u;
            bv[u] = i < n4 ? *reinterpret_cast<const float4*>(bsrc + 4 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    // The d x d loss head HERE, by the last wave of every workgroup, from the batch statistics the forward launch left
    // (all-reduced in between in a data-parallel step): ~3 us of dependent float64 algebra that sat on the forward launch's
    // critical path (its last arriver, the whole chip waiting) now runs while this wave's staging loads are in flight and
    // the other waves stage theirs.  Every workgroup computes the same matrices (deterministic: same inputs, same code);
    // the first one appends the loss record.
    bool head_ran = false;
    auto run_head = [&]() {
        head_ran = true;
        if (!a.fused.on) return;
        if (wave == kSnetWaves - 1) {
            if (lane < NG) s_stat[lane] = a.stats[lane];
            wave_sync_lds();
            const FusedHead& f = a.fused;
            double* logp = blockIdx.x == 0 ? f.log : nullptr;
            int* lc = blockIdx.x == 0 ? f.log_count : nullptr;
            switch (D) {
                case 1: tica_grad_wave<1>(s_head.h1, s_stat, f.Bg, f.reg, s_g, logp, lc, f.log_cap, f.log_width, lane); break;
                case 2: tica_grad_wave<2>(s_head.h2, s_stat, f.Bg, f.reg, s_g, logp, lc, f.log_cap, f.log_width, lane); break;
                case 3: tica_grad_wave<3>(s_head.h3, s_stat, f.Bg, f.reg, s_g, logp, lc, f.log_cap, f.log_width, lane); break;
                default: tica_grad_wave<4>(s_head.h4, s_stat, f.Bg, f.reg, s_g, logp, lc, f.log_cap, f.log_width, lane); break;
            }
        }
    };
    if (a.img != nullptr) {   // layers >= 1 of the weight image by LDS-DMA, the blob's first units behind it
        snet_stage_image<NT>(a.img, sl, a.img_bwd0, a.img_floats, t);
    } else {
        // the table entries of pass p + 1 are requested behind the data loads of pass p and arrive in the same round trip: one
        // dependent round trip per pass (+ the first table read) instead of two (round 4: 8 -> 5 for the C2 network's four passes)
        int2 e[12];
    #pragma unroll
        for (int u = 0; u < 12; ++u) {
            const int i = a.stage_bwd0 + t + NT * u;
            e[u] = i < a.stage_n ? a.stage_tab[i] : make_int2(-1, -1);
        }
        for (int i0 = a.stage_bwd0 + t; i0 < a.stage_n; i0 += 12 * NT) {
            float4 v[12];
            if (!blob_issued) {
                blob_issued = true;
                issue_blob();
            }
    #pragma unroll
            for (int u = 0; u < 12; ++u) {
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (e[u].x >= 0) {
                    const float* src = a.params + e[u].x;
                    const int nv = (e[u].y >> 20) & 7;
                    if ((e[u].y >> 24) & 1) {
                        v[u] = *reinterpret_cast<const float4*>(src);
                    } else {
                        v[u].x = src[0];
                        if (nv > 1) v[u].y = src[1];
                        if (nv > 2) v[u].z = src[2];
                        if (nv > 3) v[u].w = src[3];
                    }
                }
            }
            if (!head_ran) run_head();   // this wave's table, blob and data loads are in flight
            int2 en[12];
    #pragma unroll
            for (int u = 0; u < 12; ++u) {
                const int i = i0 + 12 * NT + NT * u;
                en[u] = i < a.stage_n ? a.stage_tab[i] : make_int2(-1, -1);
            }
    #pragma unroll
            for (int u = 0; u < 12; ++u)
                if (e[u].y >= 0) *reinterpret_cast<float4*>(sl + (e[u].y & 0xFFFFF)) = v[u];
    #pragma unroll
            for (int u = 0; u < 12; ++u) e[u] = en[u];
        }
    }
    if (!blob_issued) issue_blob();
    if (!head_ran) run_head();
    if (a.img != nullptr) vm_wait<0>();   // (the blob loads above are covered too)
    if (!a.fused.on && t < NG) s_g[t] = a.gradp[t];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int i = t + NT * u;
        if (i < n4) *reinterpret_cast<float4*>(bdst + 4 * i) = bv[u];
    }
    for (int i0 = t + 8 * NT; i0 < n4; i0 += 8 * NT) {   // larger tiles: the rest of the blob, eight units in flight
        float4 w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + NT * u;
            w[u] = i < n4 ? *reinterpret_cast<const float4*>(bsrc + 4 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + NT * u;
            if (i < n4) *reinterpret_cast<float4*>(bdst + 4 * i) = w[u];
        }
    }
    asm volatile("" ::"s"(ka_touch));
    __syncthreads();
    // ---- loss gradient of the tile's pairs, float64, rounded once (tica_dF_kernel: the rows of the exact gradient sum to
    //      zero over the batch; see DESIGN.md section 2, noise-driven parameters): dL/df_t = Gu u + Gv v + c, dL/df_lag = Gv u
    //      with u = f_t - mu, v = f_lag - mu; times act'(f) of the last layer; written over f (H_L becomes dZ_L)
    {
        float* FL = sl + a.lh[L];
        const int psL = a.ps[L];
        const int act_last = a.l[L - 1].act;
        const int pl = t / D, i = t - pl * D;
        const bool mine = t < HP * D;
        const bool valid = mine && p0 + pl < a.B;
        float ft[4], fg[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            ft[k] = (mine && k < D) ? FL[pl * psL + k] : 0.f;
            fg[k] = (mine && k < D) ? FL[(HP + pl) * psL + k] : 0.f;
        }
        __syncthreads();   // every thread has read its pair before any value is overwritten
        if (mine) {
            float gt = 0.f, gl = 0.f;
            if (valid) {
                const double* mu = s_g;
                const double* Gu = s_g + D;
                const double* Gv = Gu + D * D;
                const double* cv = Gv + D * D;
                double g1 = cv[i], g2 = 0.0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (k < D) {
                        const double u = (double)ft[k] - mu[k], v = (double)fg[k] - mu[k];
                        g1 = fma(Gu[i * D + k], u, g1);
                        g1 = fma(Gv[i * D + k], v, g1);
                        g2 = fma(Gv[i * D + k], u, g2);
                    }
                }
                float fti = ft[0], fgi = fg[0];   // ft[i] / fg[i] without a dynamically indexed register array
#pragma unroll
                for (int k = 1; k < 4; ++k) {
                    fti = i == k ? ft[k] : fti;
                    fgi = i == k ? fg[k] : fgi;
                }
                gt = (float)g1 * act_grad_from_out(act_last, fti);
                gl = (float)g2 * act_grad_from_out(act_last, fgi);
            }
            FL[pl * psL + i] = gt;
            FL[(HP + pl) * psL + i] = gl;
        }
        __syncthreads();
    }
    // ---- backward chain (snet_ae_kernel): dZ_l lives in the buffer of H_{l+1}; dZ_{l-1} is written over H_l once the weight
    //      gradient of layer l (which reads H_l) has been formed by every wave
    for (int l = L - 1; l >= 0; --l) {
        const SnetLayer& y = a.l[l];
        const float* dZ = sl + a.lh[l + 1];
        float* Hin = sl + a.lh[l];
        const int psz = a.ps[l + 1], psh = a.ps[l];
        sv4f dg[kSnetMaxTiles];
        if (l > 0) {
            const float* ap = dZ + (rg * 16 + n) * psz + 4 * q;
            const float* Wl = sl + y.lw + (4 * q) * y.pws + n;
#define SNET_DGRAD(NK)                                                                                   \
            SnetFrags<NK> A;                                                                             \
            A.load(ap);                                                                                  \
            _Pragma("unroll") for (int j = 0; j < kSnetMaxTiles; ++j) {                                  \
                const int it = cg + j * CG;                                                              \
                if (it < y.nk_in) dg[j] = snet_dgrad_tile<NK>(A, Wl + it * 16, y.pws);                   \
            }
            SNET_NK_SWITCH(y.nk_out, SNET_DGRAD)
#undef SNET_DGRAD
        }
        {
            const int nti = y.nk_in, ntot = y.nk_out * nti;
            float* pw = a.part + y.pw_off + (int64_t)blockIdx.x * y.pw_stride;
            const bool vec_ok = (y.in & 3) == 0 && ((y.pw_off + (int64_t)blockIdx.x * y.pw_stride) & 3) == 0;
            int ot = 0, it = wave;
            while (it >= nti) { it -= nti; ++ot; }
#pragma unroll 2
            for (int tile = wave; tile < ntot; tile += kSnetWaves) {
                const sv4f acc = snet_wgrad_tile<TR>(Hin + q * psh + it * 16 + n, psh, dZ + q * psz + ot * 16 + n, psz);
                const int o = ot * 16 + n, i0 = it * 16 + 4 * q;
                if (o < y.out) {
                    float* dst = pw + (int64_t)o * y.in + i0;
                    if (vec_ok && i0 + 4 <= y.in) {
                        handoff_store16(dst, acc);
                    } else {
#pragma unroll
                        for (int v = 0; v < 4; ++v)
                            if (i0 + v < y.in) dst[v] = acc[v];
                    }
                }
                it += kSnetWaves;
                while (it >= nti) { it -= nti; ++ot; }
            }
            for (int o4 = t; o4 < 4 * y.pout; o4 += NT) {
                const int o = o4 >> 2, part = o4 & 3;
                float s = 0.f;
#pragma unroll
                for (int r = 0; r < TR / 4; ++r) s += dZ[(part * (TR / 4) + r) * psz + o];
                s += __shfl_xor(s, 1, 64);
                s += __shfl_xor(s, 2, 64);
                if (part == 0 && o < y.out) a.part[y.pb_off + (int64_t)blockIdx.x * y.pb_stride + o] = s;
            }
        }
        if (l == 0) break;
        __syncthreads();   // every wave is done reading H_l
        const int act_prev = a.l[l - 1].act, out_prev = a.l[l - 1].out;
#pragma unroll
        for (int j = 0; j < kSnetMaxTiles; ++j) {
            const int it = cg + j * CG;
            if (it < y.nk_in) {
                const int col = it * 16 + n;
                float* p = Hin + (rg * 16 + 4 * q) * psh + col;
                sv4f h;
#pragma unroll
                for (int v = 0; v < 4; ++v) h[v] = p[v * psh];
                const sv4f dh = snet_actgrad4(act_prev, h);
#pragma unroll
                for (int v = 0; v < 4; ++v) p[v * psh] = col < out_prev ? dg[j][v] * dh[v] : 0.f;
            }
        }
        __syncthreads();
    }
}

struct SnetDtPlan {
    SnetDtArgs base;       // layer table, staging table; the activation map is laid out per launch (it depends on TR)
    int fl;                // LDS floats of the weight images
    int64_t per_wg;        // floats of one workgroup's gradient partials (dense)
    int2* stage_tab;
    float* part;           // gradient partials
    int64_t part_floats;
    float* blob;
    int64_t blob_floats;
    double* spart;         // statistics partials
    int64_t spart_n;
    unsigned* ev_ticket;   // batched evaluation: one ticket per batch + the one that moves the log counter (zero between launches)
    int64_t ev_ticket_n;
    int64_t last_wg;       // workgroups and tile rows of the last forward (the backward launches the same grid)
    int last_tr;
};
constexpr size_t kSnetDtLdsMax = 160 * 1024 - 8 * 1024;   // static LDS of the kernels (loss head, flags) and a margin

// activation map of a TR-row tile behind the weight images; returns the LDS floats needed
static int snet_dt_map(SnetDtArgs& a, int fl, int TR) {
    int f = fl;
    for (int l = 0; l <= a.L; ++l) {
        const int P = l == 0 ? a.l[0].pin : a.l[l - 1].pout;
        a.ps[l] = P + 4;
        a.lh[l] = f;
        f += TR * (P + 4);
    }
    a.act_len = f - a.lh[0];
    return f < 2048 ? 2048 : f;   // the last arriver sums the statistics partials in the first 4 KB
}
// rows per tile for a batch of B pairs: the smallest tile that keeps the launch at <= 256 workgroups (one per CU; the ticketed
// partial sums and the gradient reduction walk one partial per workgroup), provided the tile fits: TR * pin / 4 <= (TR / 8) * 512
// input units per thread (pin <= 64 * 32 / TR ... i.e. always for pin <= 256 at TR = 32, pin <= 256 at 64 and 128 too since
// the unit count per thread grows with TR) and the activation map fits in LDS
static int snet_dt_pick_tr(const SnetDtPlan* pl, int64_t B) {
    static const int tr_env = [] { const char* e = getenv("DCV_SNET_TR"); return e ? atoi(e) : 0; }();
    int best = 0;
    // 16-row tiles (8 pairs per workgroup) up to 1024 pairs: a shorter latency chain per workgroup while the twice-as-many
    // partials stay cheap.  A/B on one box, 54-16-8-2, contiguous batches (tools/dbg/dt_tr_probe.py): 24.2 -> 23.2 us per step
    // at 64 pairs, 24.7 -> 23.7 at 128, 25.5 -> 23.9 at 256, 25.4 -> 24.6 at 512, 26.1 -> 24.7 at 1024, and 26.1 -> 26.8
    // (slower) at 2048; the gathered batch of 128 of bench.py's ref_small block: 29.1 -> 27.2.  DCV_SNET_DT16=0 turns them off.
    static const bool dt16 = [] { const char* e = getenv("DCV_SNET_DT16"); return !(e && e[0] == '0'); }();
    // (128-64-32-4 on the same box, 16 | 32 rows: 25.8 | 28.0 at 128 pairs, 27.9 | 29.3 at 1024 -- 5.4 MB of partials and still ahead --
    // 32.4 | 30.4 at 2048, 53 | 35 at 4096: two rounds of the chip)
    if (tr_env == 16 || (tr_env == 0 && dt16 && cdiv(B, 8) <= 128)) {
        SnetDtArgs tmp = pl->base;
        if ((size_t)snet_dt_map(tmp, pl->fl, 16) * sizeof(float) <= kSnetDtLdsMax) return 16;
    }
    for (int TR : {32, 64, 128}) {
        SnetDtArgs tmp = pl->base;
        if ((size_t)snet_dt_map(tmp, pl->fl, TR) * sizeof(float) > kSnetDtLdsMax) break;
        // the backward keeps a layer's input-gradient tiles of a wave in registers: kSnetMaxTiles * (8 waves / (TR / 16) row groups)
        // column tiles of 16 must cover the widest hidden layer
        bool fits = true;
        for (int l = 1; l < tmp.L; ++l) fits = fits && tmp.l[l].nk_in <= kSnetMaxTiles * (kSnetWaves / (TR / 16));
        if (!fits) break;
        if (tr_env != 0) {
            if (tr_env == TR) return TR;
            continue;
        }
        best = TR;
        if (cdiv(B, TR / 2) <= 256) break;   // one round of the chip; measured at 4096 pairs: 128 tiles of 64 rows 41.8 us / step, 256 of 32 rows 38.2
    }
    return best;
}

template <class T>
static bool grow(T** p, int64_t* have, int64_t need) {
    if (*have >= need) return true;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *have = 0;
    if (hipMalloc(reinterpret_cast<void**>(p), (size_t)need * sizeof(T)) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    *have = need;
    return true;
}

static bool snet_dt_build(dcv_mlp* m) {
    if (m->desc.model != DCV_MODEL_DEEPTICA || m->any_drop || m->any_bn || m->d_out > 4 || m->d_out < 1 || snet_disabled()) return false;
    SnetDtPlan* pl = new (std::nothrow) SnetDtPlan();
    if (!pl) return false;
    *pl = SnetDtPlan{};
    SnetDtArgs& a = pl->base;
    a.L = m->L;
    a.d = m->d_out;
    int fl = 0;
    int64_t per_wg = 0;
    std::vector<int2> tab;
    int tab_begin[DCV_MAX_LAYERS];
    if (!snet_layout(m, a.l, tab, tab_begin, fl, per_wg)) { delete pl; return false; }
    pl->per_wg = per_wg;
    (void)snet_image_build(m);   // on failure the kernels keep the table-driven staging
    pl->fl = fl;
    {
        SnetDtArgs tmp = a;
        if ((size_t)snet_dt_map(tmp, fl, 32) * sizeof(float) > kSnetDtLdsMax) { delete pl; return false; }
    }
    if (hipMalloc(reinterpret_cast<void**>(&pl->stage_tab), tab.size() * sizeof(int2)) != hipSuccess ||
        hipMemcpy(pl->stage_tab, tab.data(), tab.size() * sizeof(int2), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipGetLastError();
        if (pl->stage_tab) (void)hipFree(pl->stage_tab);
        delete pl;
        return false;
    }
    a.stage_tab = pl->stage_tab;
    a.stage_n = (int)tab.size();
    a.stage_bwd0 = m->L > 1 ? tab_begin[1] : (int)tab.size();
    // batched validation passes: statistics partials and tickets for the bounds of dcv_mlp_eval_steps, allocated now so that no
    // allocation lands in a timed pass (without them the passes go batch by batch)
    if (grow(&pl->spart, &pl->spart_n, kEvalWorkgroupsPerLaunch * (int64_t)m->stats_len) && grow(&pl->ev_ticket, &pl->ev_ticket_n, (int64_t)kEvalBatchesPerLaunch + 1)) {
        if (hipMemset(pl->ev_ticket, 0, (size_t)pl->ev_ticket_n * sizeof(unsigned)) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipFree(pl->ev_ticket);
            pl->ev_ticket = nullptr;
            pl->ev_ticket_n = 0;
        }
    }
    m->snet_dt = pl;
    return true;
}

void snet_dt_free(dcv_mlp* m) {
    SnetDtPlan* pl = static_cast<SnetDtPlan*>(m->snet_dt);
    if (!pl) return;
    if (pl->part) (void)hipFree(pl->part);
    if (pl->blob) (void)hipFree(pl->blob);
    if (pl->spart) (void)hipFree(pl->spart);
    if (pl->ev_ticket) (void)hipFree(pl->ev_ticket);
    if (pl->stage_tab) (void)hipFree(pl->stage_tab);
    delete pl;
    m->snet_dt = nullptr;
}

template <class K>
static int snet_dt_launch(K kern, int slot, size_t lds_bytes, const SnetDtArgs& a, int64_t nwg, hipStream_t s) {
    static int attr_state[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // per kernel instantiation: 0 unknown, 1 set, -1 refused by the runtime
    if (attr_state[slot] == 0) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSnetDtLdsMax);
        if (e != hipSuccess) (void)hipGetLastError();
        attr_state[slot] = e == hipSuccess ? 1 : -1;
    }
    if (attr_state[slot] < 0) return 1;
    if (g_launch_ev.start != nullptr) {   // a profiled launch: events stamped with the kernel's own begin / end (common.h)
        const LaunchEvents ev = g_launch_ev;
        g_launch_ev = LaunchEvents{};
        g_launch_taken = ev.start;
        hipExtLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(kSnetThreads), (uint32_t)lds_bytes, s, ev.start, ev.stop, 0u, a);
    } else {
        hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(kSnetThreads), lds_bytes, s, a);
    }
    DCV_CHECK_LAUNCH();
    return DCV_OK;
}

// Fused forward of one Deep-TICA batch (+ batch statistics, + the loss head when head != 0: 1 = training, the head's
// matrices go to m->gradp; 2 = evaluation).  keep_blob: a backward may follow.  Returns 1 when the fused form does not
// apply (the caller runs the layer-by-layer path), DCV_OK when the launch was enqueued.
// nb > 1 (head == 2, no blob): nb batches of `batch` pairs each in the one launch, one loss record per batch in batch order.
int snet_dt_forward(dcv_mlp* m, const float* Xn_d, int64_t ld, const int64_t* idx_d, int64_t row0, int32_t batch, int head, bool keep_blob,
                    hipStream_t s, int nb) {
    static const int64_t kMaxBytes = 64ll << 20;
    if (m->snet_dt == nullptr) {
        if (m->snet_dt_tried || !snet_dt_build(m)) {
            m->snet_dt_tried = true;
            return 1;
        }
        m->snet_dt_tried = true;
    }
    SnetDtPlan* pl = static_cast<SnetDtPlan*>(m->snet_dt);
    const int TR = snet_dt_pick_tr(pl, batch);
    if (TR == 0) return 1;
    SnetDtArgs a = pl->base;
    const size_t lds_bytes = (size_t)snet_dt_map(a, pl->fl, TR) * sizeof(float);
    const int64_t wgpb = cdiv(batch, TR / 2);
    const int W = m->stats_len;
    if (wgpb > 512 || wgpb * pl->per_wg * (int64_t)sizeof(float) > kMaxBytes || wgpb * a.act_len * (int64_t)sizeof(float) > kMaxBytes) return 1;
    if (nb < 1 || (nb > 1 && (head != 2 || keep_blob))) return 1;
    const int64_t nwg = wgpb * nb;
    if (!grow(&pl->spart, &pl->spart_n, nwg * W)) return 1;
    if (nb > 1 && nb + 1 > pl->ev_ticket_n) return 1;   // (sized by snet_dt_build for the bounds of dcv_mlp_eval_steps)
    if (keep_blob && !grow(&pl->blob, &pl->blob_floats, nwg * (int64_t)a.act_len)) return 1;
    a.params = m->params;
    a.img = m->snet_img;
    a.img_floats = m->snet_img_floats;
    a.img_bwd0 = m->L > 1 ? a.l[1].lw : m->snet_img_floats;
    a.Xn = Xn_d;
    a.ld = ld;
    a.rows = RowMap{idx_d, row0, batch, m->desc.lag};
    a.B = batch;
    a.nb = nb;
    a.wgpb = (int)wgpb;
    a.store_blob = keep_blob && pl->blob != nullptr ? 1 : 0;
    a.blob = pl->blob;
    a.blob_stride = a.act_len;
    a.spart = pl->spart;
    a.ticket = nb > 1 ? pl->ev_ticket : m->ticket;
    a.stats = m->stats;
    a.fused = FusedHead{0, 0.0, 0.0, nullptr, nullptr, nullptr, 0, 0};
    if (head) a.fused = FusedHead{1, (double)batch, m->desc.tica_reg, head == 1 ? m->gradp : nullptr, m->log, m->log_count, m->log_cap, m->log_width};
    a.gradp = nullptr;
    a.part = nullptr;
    pl->last_wg = nb > 1 ? -1 : nwg;   // (no backward behind a batched evaluation)
    pl->last_tr = TR;
    switch (TR) {
        case 16: return snet_dt_launch(snet_dt_fwd_kernel<16>, 3, lds_bytes, a, nwg, s);
        case 32: return snet_dt_launch(snet_dt_fwd_kernel<32>, 0, lds_bytes, a, nwg, s);
        case 64: return snet_dt_launch(snet_dt_fwd_kernel<64>, 1, lds_bytes, a, nwg, s);
        default: return snet_dt_launch(snet_dt_fwd_kernel<128>, 2, lds_bytes, a, nwg, s);
    }
}

// Fused backward of the batch whose forward snet_dt_forward ran last (its blob is in place): gradient partials for the
// reduction, whose descriptors go to `ra`.  head: the loss head has not run yet -- every workgroup evaluates it from m->stats
// (the sums over the GLOBAL batch) and the first one appends the loss record; otherwise m->gradp holds the head's matrices.
int snet_dt_backward(dcv_mlp* m, int32_t batch, int64_t global_batch, bool head, ReduceArgsView* ra, hipStream_t s) {
    SnetDtPlan* pl = static_cast<SnetDtPlan*>(m->snet_dt);
    if (!pl || !pl->blob) {
        set_error("snet_dt_backward: no fused forward to go back through");
        return DCV_ESTATE;
    }
    const int TR = pl->last_tr;
    const int64_t nwg = cdiv(batch, TR / 2);
    if (nwg != pl->last_wg) {
        set_error("snet_dt_backward: batch=%d does not match the fused forward", batch);
        return DCV_ESTATE;
    }
    const int64_t part_need = nwg * pl->per_wg + 8 * (int64_t)m->L;
    if (!grow(&pl->part, &pl->part_floats, part_need)) {
        set_error("snet_dt_backward: out of device memory (%lld floats of gradient partials)", (long long)part_need);
        return DCV_ENOMEM;
    }
    SnetDtArgs a = pl->base;
    const size_t lds_bytes = (size_t)snet_dt_map(a, pl->fl, TR) * sizeof(float);
    int64_t off = 0;
    for (int l = 0; l < m->L; ++l) {
        SnetLayer& y = a.l[l];
        y.pw_off = off; off += nwg * (int64_t)y.pw_stride;
        y.pb_off = off; off += nwg * (int64_t)y.pb_stride;
        ra->slab[l] = pl->part + y.pw_off;
        ra->bpart[l] = pl->part + y.pb_off;
        ra->splits[l] = (int)nwg;
        ra->bblocks[l] = (int)nwg;
        ra->wstride[l] = y.pw_stride;
        ra->bstride[l] = y.pb_stride;
    }
    a.params = m->params;
    a.img = m->snet_img;
    a.img_floats = m->snet_img_floats;
    a.img_bwd0 = m->L > 1 ? a.l[1].lw : m->snet_img_floats;
    a.B = batch;
    a.blob = pl->blob;
    a.blob_stride = a.act_len;
    a.gradp = m->gradp;
    a.part = pl->part;
    a.stats = m->stats;
    a.fused = FusedHead{0, 0.0, 0.0, nullptr, nullptr, nullptr, 0, 0};
    if (head) a.fused = FusedHead{1, (double)global_batch, m->desc.tica_reg, nullptr, m->log, m->log_count, m->log_cap, m->log_width};
    switch (TR) {
        case 16: return snet_dt_launch(snet_dt_bwd_kernel<16>, 7, lds_bytes, a, nwg, s);
        case 32: return snet_dt_launch(snet_dt_bwd_kernel<32>, 4, lds_bytes, a, nwg, s);
        case 64: return snet_dt_launch(snet_dt_bwd_kernel<64>, 5, lds_bytes, a, nwg, s);
        default: return snet_dt_launch(snet_dt_bwd_kernel<128>, 6, lds_bytes, a, nwg, s);
    }
}

}  // namespace dcv
