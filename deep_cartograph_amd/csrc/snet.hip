// Fused training / evaluation step of a SMALL network: every weight of the MLP resident in one CU's LDS.
//
// The reference's own configurations are small networks (its autoencoder test model is 54-16-8-2-4-8-54; BASELINE C2 is
// 128-64-32-2-32-64-128 = 82 KB of parameters).  Layer by layer, such a step is ~20 launches of 5-16 us each on a
// few thousand rows -- launch- and latency-bound, < 2 % of any roofline.  Here ONE launch does the whole step for the
// autoencoder: a workgroup stages all weights into LDS once (padded rows: conflict-free 128-bit fragment reads), takes a
// tile of TR rows, runs the forward chain with the activations kept in LDS, forms the loss gradient in the epilogue of
// the last layer, and walks back through the layers -- weight gradient of the tile (contraction over its TR rows) and
// input gradient, the latter written over the activation it consumes -- without an activation ever leaving the CU.
// Out: one gradient partial per workgroup and layer (slab layout of the split-K reduction, so reduce_grads_small_kernel
// + the fused optimiser update finish the step), the squared-error partial, and through a ticketed hand-off (handoff.h)
// the step's loss record.
//
// Arithmetic: v_mfma_f32_16x16x4_f32 (exact f32 products, f32 accumulate) in both arithmetic flavours of the library --
// the network is far too small for the matrix rate to matter, and no operand splitting is needed.
// A wave owns 16-row groups: TR = 32 -> waves (row group, column-tile parity); the k-slot permutation of gemm.h lets
// one ds_read_b128 feed four MFMA steps (lane (n, q) holds k = k0 + 4q + s in step s, identically for A and B).
#include "mlp_state.h"
#include <new>

namespace dcv {

typedef float sv4f __attribute__((ext_vector_type(4)));

struct SnetLayer {
    int in, out, pin, pout, act;
    int64_t w_off, b_off;       // flat parameter buffer
    int lw, lb, pws;            // LDS float offsets of the weight image [pout][pws] and the bias [pout]; pws = pin + 4
    int64_t pw_off, pb_off;     // gradient partials: part + pw_off + wg * out * in ; part + pb_off + wg * out
};
struct SnetArgs {
    SnetLayer l[DCV_MAX_LAYERS];
    int L;
    int lh[DCV_MAX_LAYERS + 1];   // LDS float offset of H_l [TR][ps_l]   (H_0 = the input tile)
    int ps[DCV_MAX_LAYERS + 1];   // row stride of H_l = round_up(dims[l], 16) + 4
    int lred;                     // LDS float offset of the reduction scratch (256 doubles)
    const float* params;
    const float* Xn;
    int64_t ld;
    RowMap rows;
    int64_t R;
    const float* range;
    float scale;                  // 2 / (global batch * F)
    int train;
    float* part;
    double* sse_part;
    unsigned* ticket;
    double* stats;
    double Bg;
    double* log;
    int* log_count;
    int log_cap, log_width;
};

__device__ __forceinline__ sv4f mfma4(float a, float b, sv4f c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// D[r][c] (16 rows of row group rg, 16 columns of column tile ct) = sum_k Hin[r][k] W[c][k], k < pin
__device__ __forceinline__ sv4f snet_fwd_tile(const float* Hin, int psin, const float* W, int pws, int rg, int ct, int pin, int lane) {
    sv4f acc = {0.f, 0.f, 0.f, 0.f};
    const float* ap = Hin + (rg * 16 + (lane & 15)) * psin + 4 * (lane >> 4);
    const float* bp = W + (ct * 16 + (lane & 15)) * pws + 4 * (lane >> 4);
#pragma unroll 2
    for (int k0 = 0; k0 < pin; k0 += 16) {
        const sv4f a = *reinterpret_cast<const sv4f*>(ap + k0);
        const sv4f b = *reinterpret_cast<const sv4f*>(bp + k0);
        acc = mfma4(a[0], b[0], acc);
        acc = mfma4(a[1], b[1], acc);
        acc = mfma4(a[2], b[2], acc);
        acc = mfma4(a[3], b[3], acc);
    }
    return acc;
}
// D[r][i] = sum_o dZ[r][o] W[o][i], o < pout  (input gradient before the activation derivative)
__device__ __forceinline__ sv4f snet_dgrad_tile(const float* dZ, int psz, const float* W, int pws, int rg, int it, int pout, int lane) {
    sv4f acc = {0.f, 0.f, 0.f, 0.f};
    const int q = lane >> 4, n = lane & 15;
    const float* ap = dZ + (rg * 16 + n) * psz + 4 * q;
    const float* bp = W + (4 * q) * pws + it * 16 + n;
#pragma unroll 2
    for (int k0 = 0; k0 < pout; k0 += 16) {
        const sv4f a = *reinterpret_cast<const sv4f*>(ap + k0);
        const float* b = bp + k0 * pws;
        acc = mfma4(a[0], b[0], acc);
        acc = mfma4(a[1], b[pws], acc);
        acc = mfma4(a[2], b[2 * pws], acc);
        acc = mfma4(a[3], b[3 * pws], acc);
    }
    return acc;
}
// D[o][i] = sum_r dZ[r][o] Hin[r][i], r < TR  (weight gradient of the tile)
template <int TR>
__device__ __forceinline__ sv4f snet_wgrad_tile(const float* dZ, int psz, const float* Hin, int psh, int ot, int it, int lane) {
    sv4f acc = {0.f, 0.f, 0.f, 0.f};
    const int q = lane >> 4, n = lane & 15;
    const float* ap = dZ + q * psz + ot * 16 + n;
    const float* bp = Hin + q * psh + it * 16 + n;
#pragma unroll
    for (int s = 0; s < TR / 4; ++s) acc = mfma4(ap[4 * s * psz], bp[4 * s * psh], acc);
    return acc;
}

constexpr int kSnetMaxTiles = 8;   // column tiles of a layer per wave: widths up to 16 * CG * 8

template <int TR>
__global__ __launch_bounds__(256) void snet_ae_kernel(SnetArgs a) {
    constexpr int RG = TR / 16, CG = 4 / RG;
    extern __shared__ __attribute__((aligned(16))) float sl[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int rg = wave / CG, cg = wave % CG;
    const int q = lane >> 4, n = lane & 15;
    const int L = a.L;
    // ---- stage every weight and bias (zero-padded to 16 x 16 tiles, row stride pin + 4)
    for (int l = 0; l < L; ++l) {
        const SnetLayer& y = a.l[l];
        const float* W = a.params + y.w_off;
        float* dst = sl + y.lw;
        if ((y.in & 3) == 0) {
            const int c4 = y.pin >> 2, in4 = y.in >> 2;
            for (int i = t; i < y.pout * c4; i += 256) {
                const int o = i / c4, c = i - o * c4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (o < y.out && c < in4) v = *reinterpret_cast<const float4*>(W + (int64_t)o * y.in + 4 * c);
                *reinterpret_cast<float4*>(dst + o * y.pws + 4 * c) = v;
            }
        } else {
            for (int i = t; i < y.pout * y.pin; i += 256) {
                const int o = i / y.pin, c = i - o * y.pin;
                dst[o * y.pws + c] = (o < y.out && c < y.in) ? W[(int64_t)o * y.in + c] : 0.f;
            }
        }
        for (int i = t; i < y.pout; i += 256) sl[y.lb + i] = i < y.out ? a.params[y.b_off + i] : 0.f;
    }
    // ---- input tile H_0 (rows past the batch: zeros)
    const int64_t r0 = (int64_t)blockIdx.x * TR;
    {
        const int F = a.l[0].in, p0 = a.l[0].pin, ps0 = a.ps[0];
        float* H0 = sl + a.lh[0];
        for (int i = t; i < TR * p0; i += 256) {
            const int r = i / p0, c = i - r * p0;
            float v = 0.f;
            if (r0 + r < a.R && c < F) v = a.Xn[a.rows.template get<true>(r0 + r) * a.ld + c];
            H0[r * ps0 + c] = v;
        }
    }
    __syncthreads();
    // ---- forward chain
    double sse = 0.0;
    for (int l = 0; l < L; ++l) {
        const SnetLayer& y = a.l[l];
        const float* Hin = sl + a.lh[l];
        float* Hout = sl + a.lh[l + 1];
        const int psin = a.ps[l], pso = a.ps[l + 1];
        const bool last = l == L - 1;
        for (int ct = cg; ct < y.pout / 16; ct += CG) {
            const sv4f acc = snet_fwd_tile(Hin, psin, sl + y.lw, y.pws, rg, ct, y.pin, lane);
            const int col = ct * 16 + n;
            const float bias = sl[y.lb + col];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int row = rg * 16 + 4 * q + v;
                float h = col < y.out ? act_fwd(y.act, acc[v] + bias) : 0.f;
                if (last) {
                    // autoencoder loss on the spot: e = (y - xn) * range ; dY = scale * (y - xn) * range^2 * act'(y)
                    float g = 0.f;
                    if (col < y.out && r0 + row < a.R) {
                        const float x = sl[a.lh[0] + row * a.ps[0] + col];
                        const float rgv = a.range[col];
                        const float ev = (h - x) * rgv;
                        sse += (double)ev * (double)ev;
                        g = a.scale * (h - x) * rgv * rgv * act_grad_from_out(y.act, h);
                    }
                    h = g;   // H_L now holds dZ_L
                }
                Hout[row * pso + col] = h;
            }
        }
        __syncthreads();
    }
    // ---- squared error of the tile -> partial -> (ticket) the step's loss record
    {
        double* red = reinterpret_cast<double*>(sl + a.lred);
        red[t] = sse;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if (t < off) red[t] += red[t + off];
            __syncthreads();
        }
        if (t == 0) handoff_store(a.sse_part + blockIdx.x, red[0]);
    }
    // ---- backward chain: dZ_l lives in the buffer of H_{l+1}; dZ_{l-1} is written over H_l once the weight gradient
    //      of layer l (which reads H_l) has been formed by every wave
    if (a.train) {
        for (int l = L - 1; l >= 0; --l) {
            const SnetLayer& y = a.l[l];
            const float* dZ = sl + a.lh[l + 1];
            float* Hin = sl + a.lh[l];
            const int psz = a.ps[l + 1], psh = a.ps[l];
            // input gradient first, kept in registers
            sv4f dg[kSnetMaxTiles];
            if (l > 0) {
#pragma unroll
                for (int j = 0; j < kSnetMaxTiles; ++j) {
                    const int it = cg + j * CG;
                    if (it < y.pin / 16) dg[j] = snet_dgrad_tile(dZ, psz, sl + y.lw, y.pws, rg, it, y.pout, lane);
                }
            }
            // weight gradient of the tile: the (pout / 16) x (pin / 16) tiles round-robin over the four waves
            {
                const int nto = y.pout / 16, nti = y.pin / 16;
                float* pw = a.part + y.pw_off + (int64_t)blockIdx.x * y.out * y.in;
                for (int tile = wave; tile < nto * nti; tile += 4) {
                    const int ot = tile / nti, it = tile - ot * nti;
                    const sv4f acc = snet_wgrad_tile<TR>(dZ, psz, Hin, psh, ot, it, lane);
                    const int i = it * 16 + n;
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int o = ot * 16 + 4 * q + v;
                        if (o < y.out && i < y.in) pw[(int64_t)o * y.in + i] = acc[v];
                    }
                }
                // bias gradient: column sums of dZ_l over the tile's rows, rows in index order
                for (int o = t; o < y.out; o += 256) {
                    float s = 0.f;
#pragma unroll 8
                    for (int r = 0; r < TR; ++r) s += dZ[r * psz + o];
                    a.part[y.pb_off + (int64_t)blockIdx.x * y.out + o] = s;
                }
            }
            if (l == 0) break;
            __syncthreads();   // every wave is done reading H_l
            const int act_prev = a.l[l - 1].act, out_prev = a.l[l - 1].out;
#pragma unroll
            for (int j = 0; j < kSnetMaxTiles; ++j) {
                const int it = cg + j * CG;
                if (it < y.pin / 16) {
                    const int col = it * 16 + n;
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int row = rg * 16 + 4 * q + v;
                        float* p = Hin + row * psh + col;
                        *p = col < out_prev ? dg[j][v] * act_grad_from_out(act_prev, *p) : 0.f;
                    }
                }
            }
            __syncthreads();
        }
    }
    // ---- last workgroup: total squared error in block order, loss record
    unsigned* flag = reinterpret_cast<unsigned*>(sl + a.lred);
    __syncthreads();
    if (!handoff_arrive_last(a.ticket, gridDim.x, flag)) return;
    if (t < 64) {
        double tot = 0.0;
        for (int b0 = t; b0 < (int)gridDim.x; b0 += 64) tot += handoff_load(a.sse_part + b0);
        for (int off = 32; off > 0; off >>= 1) tot += __shfl_down(tot, off, 64);
        if (t == 0) {
            a.stats[0] = tot;
            const int slot = *a.log_count;
            if (slot < a.log_cap) {
                a.log[(int64_t)slot * a.log_width + 0] = tot / (a.Bg * (double)a.l[0].in);
                a.log[(int64_t)slot * a.log_width + 1] = a.Bg;
            }
            *a.log_count = slot + 1;
        }
    }
}

struct SnetPlan {
    int TR;
    size_t lds_bytes;
    SnetArgs base;       // layer table and LDS map
    float* part;         // gradient partials
    int64_t part_floats; // capacity
    int64_t per_wg;      // floats of one workgroup's partials over all layers (dense: sum out * in + out)
};

static bool snet_disabled() {
    static const bool off = [] { const char* e = getenv("DCV_NO_SNET"); return e && e[0] == '1'; }();
    return off;
}

// Builds the plan once per engine.  Not applicable (returns false): wide layers, dropout, a network that does not fit
// in LDS with at least 16-row tiles.
static bool snet_build(dcv_mlp* m) {
    if (m->desc.model != DCV_MODEL_AE || m->any_drop || snet_disabled()) return false;
    SnetPlan* pl = new (std::nothrow) SnetPlan();
    if (!pl) return false;
    SnetArgs& a = pl->base;
    a.L = m->L;
    int fl = 0;
    int64_t per_wg = 0;
    for (int l = 0; l < m->L; ++l) {
        const LayerPlan& p = m->layers[l];
        SnetLayer& y = a.l[l];
        y.in = p.in; y.out = p.out; y.act = p.act;
        y.pin = (p.in + 15) / 16 * 16;
        y.pout = (p.out + 15) / 16 * 16;
        if (y.pin > 16 * 2 * kSnetMaxTiles || y.pout > 16 * 2 * kSnetMaxTiles) { delete pl; return false; }
        y.w_off = p.w_off; y.b_off = p.b_off;
        y.pws = y.pin + 4;
        y.lw = fl; fl += y.pout * y.pws;
        y.lb = fl; fl += y.pout;
        per_wg += (int64_t)p.out * p.in + p.out;
    }
    pl->per_wg = per_wg;
    const size_t lds_max = 160 * 1024;
    for (int TR : {32, 16}) {
        int f = fl;
        for (int l = 0; l <= m->L; ++l) {
            const int P = ((l == 0 ? m->layers[0].in : m->layers[l - 1].out) + 15) / 16 * 16;
            a.ps[l] = P + 4;
            a.lh[l] = f;
            f += TR * (P + 4);
        }
        f = (f + 3) / 4 * 4;
        a.lred = f;
        f += 512;   // 256 doubles
        if ((size_t)f * sizeof(float) <= lds_max) {
            pl->TR = TR;
            pl->lds_bytes = (size_t)f * sizeof(float);
            pl->part = nullptr;
            pl->part_floats = 0;
            m->snet = pl;
            return true;
        }
    }
    delete pl;
    return false;
}

void snet_free(dcv_mlp* m) {
    SnetPlan* pl = static_cast<SnetPlan*>(m->snet);
    if (!pl) return;
    if (pl->part) (void)hipFree(pl->part);
    delete pl;
    m->snet = nullptr;
}

// One fused step of the autoencoder over `R` rows (train != 0: gradient partials are left for the reduction, whose
// descriptors are filled into `ra`).  Returns 1 when the fused form does not apply (the caller takes the layer-by-layer
// path), DCV_OK when the launch was enqueued.
int snet_ae_step(dcv_mlp* m, const float* Xn_d, int64_t ld, const RowMap& rm, int64_t R, int32_t batch, int train, ReduceArgsView* ra,
                 hipStream_t s) {
    static const int64_t kMaxPartBytes = 96ll << 20;
    if (m->snet == nullptr) {
        if (m->snet_tried || !snet_build(m)) {
            m->snet_tried = true;
            return 1;
        }
        m->snet_tried = true;
    }
    SnetPlan* pl = static_cast<SnetPlan*>(m->snet);
    const int64_t nwg = cdiv(R, pl->TR);
    if (nwg > m->spart_blocks || nwg * pl->per_wg * (int64_t)sizeof(float) > kMaxPartBytes || nwg > 512) return 1;   // large batches: the tiled products are the better engine
    if (train && pl->part_floats < nwg * pl->per_wg) {
        if (pl->part) (void)hipFree(pl->part);
        pl->part = nullptr;
        pl->part_floats = 0;
        if (hipMalloc(reinterpret_cast<void**>(&pl->part), (size_t)(nwg * pl->per_wg) * sizeof(float)) != hipSuccess) {
            (void)hipGetLastError();
            return 1;
        }
        pl->part_floats = nwg * pl->per_wg;
    }
    SnetArgs a = pl->base;
    int64_t off = 0;
    for (int l = 0; l < m->L; ++l) {
        SnetLayer& y = a.l[l];
        y.pw_off = off; off += nwg * (int64_t)y.out * y.in;
        y.pb_off = off; off += nwg * (int64_t)y.out;
        if (ra) {
            ra->slab[l] = pl->part + y.pw_off;
            ra->bpart[l] = pl->part + y.pb_off;
            ra->splits[l] = (int)nwg;
            ra->bblocks[l] = (int)nwg;
        }
    }
    a.params = m->params;
    a.Xn = Xn_d;
    a.ld = ld;
    a.rows = rm;
    a.R = R;
    a.range = m->feat_range;
    a.scale = (float)(2.0 / ((double)batch * (double)m->desc.dims[0]));
    a.train = train;
    a.part = pl->part;
    a.sse_part = m->spart;
    a.ticket = m->ticket;
    a.stats = m->stats;
    a.Bg = (double)batch;
    a.log = m->log;
    a.log_count = m->log_count;
    a.log_cap = m->log_cap;
    a.log_width = m->log_width;
    auto launch = [&](auto kern) -> int {
        static int attr_state[2] = {0, 0};   // 0 unknown, 1 set, -1 refused by the runtime (the fused form is then off)
        const int slot = pl->TR == 32 ? 0 : 1;
        if (attr_state[slot] == 0) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) (void)hipGetLastError();
            attr_state[slot] = e == hipSuccess ? 1 : -1;
        }
        if (attr_state[slot] < 0) return 1;
        hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(256), pl->lds_bytes, s, a);
        DCV_CHECK_LAUNCH();
        return DCV_OK;
    };
    return pl->TR == 32 ? launch(snet_ae_kernel<32>) : launch(snet_ae_kernel<16>);
}

}  // namespace dcv
