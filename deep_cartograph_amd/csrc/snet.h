// Pieces shared by the fused small-network kernels (snet.hip: autoencoder step; snet_dt.hip: Deep-TICA forward / backward):
// the LDS layout of the zero-padded weight images, the v_mfma_f32_16x16x4_f32 tile products with compile-time contraction
// lengths, the activation helpers.  See snet.hip for the design notes.
#pragma once
#include <hip/hip_ext.h>
#include "mlp_state.h"
#include <vector>

namespace dcv {

typedef float sv4f __attribute__((ext_vector_type(4)));

constexpr int kSnetThreads = 512;   // 8 waves: two per SIMD, one hides the other's LDS and MFMA latency
constexpr int kSnetWaves = kSnetThreads / 64;

struct SnetLayer {
    int in, out, pin, pout, act;
    int nk_in, nk_out;          // pin / 16, pout / 16
    int u4_begin, c4_shift;     // staging: first 16-byte unit of this layer's weight image in the flat unit space; log2(pin / 4)
    int64_t w_off, b_off;       // flat parameter buffer
    int lw, lb, pws;            // LDS float offsets of the weight image [pout][pws] and the bias [pout]; pws = pin + 4
    int64_t pw_off, pb_off;     // gradient partials: part + pw_off + wg * pw_stride ; part + pb_off + wg * pb_stride
    int pw_stride, pb_stride;   // out * in and out rounded up to multiples of 4 floats (16-byte loads in the reduction)
};
__device__ __forceinline__ sv4f mfma4(float a, float b, sv4f c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// Contraction lengths are compile-time (NK chunks of 16; the plan pads every width to 16 * 2^j): the fragment reads of
// a tile are issued back to back, the MFMAs run in up to four independent accumulator chains, no branch in between.
// (With run-time trip counts hipcc emitted ds_read -> s_waitcnt lgkmcnt(0) -> 4 dependent MFMAs -> branch per chunk:
// 1.2 us of fixed cost per layer, 51 us per fused step.)
template <int NK>
struct SnetFrags {
    sv4f f[NK];
    __device__ __forceinline__ void load(const float* p) {   // p: this lane's row, offset 4 * q; chunk j at p + 16 j
#pragma unroll
        for (int j = 0; j < NK; ++j) f[j] = *reinterpret_cast<const sv4f*>(p + 16 * j);
    }
};
template <int NK>
__device__ __forceinline__ sv4f snet_chain_sum(sv4f (&acc)[(NK < 4 ? NK : 4)]) {
    constexpr int C = NK < 4 ? NK : 4;
    sv4f r = acc[0];
#pragma unroll
    for (int c = 1; c < C; ++c) r += acc[c];
    return r;
}
// D[r][c] = sum_k A[r][k] W[c][k]  (forward: A = activations of the wave's 16 rows, W row-major [out][in])
template <int NK>
__device__ __forceinline__ sv4f snet_fwd_tile(const SnetFrags<NK>& A, const float* bp) {
    constexpr int C = NK < 4 ? NK : 4;
    SnetFrags<NK> B;
    B.load(bp);
    sv4f acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = sv4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NK; ++j) {
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[j % C] = mfma4(A.f[j][s], B.f[j][s], acc[j % C]);
    }
    return snet_chain_sum<NK>(acc);
}
// D[r][i] = sum_o dZ[r][o] W[o][i]  (input gradient: A = dZ rows, B = a column block of W read down the rows)
template <int NK>
__device__ __forceinline__ sv4f snet_dgrad_tile(const SnetFrags<NK>& A, const float* bp, int pws) {
    constexpr int C = NK < 4 ? NK : 4;
    float b[NK][4];
#pragma unroll
    for (int j = 0; j < NK; ++j)
#pragma unroll
        for (int s = 0; s < 4; ++s) b[j][s] = bp[(16 * j + s) * pws];
    sv4f acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = sv4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NK; ++j) {
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[j % C] = mfma4(A.f[j][s], b[j][s], acc[j % C]);
    }
    return snet_chain_sum<NK>(acc);
}
// D[o][i] = sum_r dZ[r][o] Hin[r][i], r < TR  (weight gradient of the tile)
template <int TR>
__device__ __forceinline__ sv4f snet_wgrad_tile(const float* ap, int psz, const float* bp, int psh) {
    float av[TR / 4], bv[TR / 4];
#pragma unroll
    for (int s = 0; s < TR / 4; ++s) {
        av[s] = ap[4 * s * psz];
        bv[s] = bp[4 * s * psh];
    }
    sv4f acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < TR / 4; s += 2) {
        acc0 = mfma4(av[s], bv[s], acc0);
        acc1 = mfma4(av[s + 1], bv[s + 1], acc1);
    }
    return acc0 + acc1;
}

// activation with the switch outside the element loop
__device__ __forceinline__ sv4f snet_act4(int act, sv4f z) {
    sv4f h;
    switch (act) {
        case DCV_ACT_NONE: h = z; break;
        case DCV_ACT_LEAKY_RELU:
#pragma unroll
            for (int v = 0; v < 4; ++v) h[v] = z[v] > 0.f ? z[v] : 0.01f * z[v];
            break;
        case DCV_ACT_RELU:
#pragma unroll
            for (int v = 0; v < 4; ++v) h[v] = z[v] > 0.f ? z[v] : 0.f;
            break;
        default:
#pragma unroll
            for (int v = 0; v < 4; ++v) h[v] = act_fwd(act, z[v]);
            break;
    }
    return h;
}
__device__ __forceinline__ sv4f snet_actgrad4(int act, sv4f h) {
    sv4f g;
    switch (act) {
        case DCV_ACT_NONE: g = sv4f{1.f, 1.f, 1.f, 1.f}; break;
        case DCV_ACT_LEAKY_RELU:
#pragma unroll
            for (int v = 0; v < 4; ++v) g[v] = h[v] > 0.f ? 1.f : 0.01f;
            break;
        case DCV_ACT_RELU:
#pragma unroll
            for (int v = 0; v < 4; ++v) g[v] = h[v] > 0.f ? 1.f : 0.f;
            break;
        default:
#pragma unroll
            for (int v = 0; v < 4; ++v) g[v] = act_grad_from_out(act, h[v]);
            break;
    }
    return g;
}

constexpr int kSnetMaxTiles = 8;   // column tiles of a layer per wave (input gradients are held in registers across a barrier)

#define SNET_NK_SWITCH(nk, CALL)          \
    switch (nk) {                         \
        case 1: { CALL(1) } break;        \
        case 2: { CALL(2) } break;        \
        case 4: { CALL(4) } break;        \
        case 8: { CALL(8) } break;        \
        default: { CALL(16) } break;      \
    }


// Layer table + staging table of a network whose weight images all live in one CU's LDS.  Every width is padded to
// 16 * 2^j (compile-time contraction lengths); weight image [pout][pin + 4] (conflict-free 128-bit fragment reads), then the
// bias [pout].  tab: per 16-byte unit of the LDS image {source element offset into params or -1, LDS float offset |
// valid elements << 20 | 16-byte load legal << 24}; tab_begin[l] = first entry of layer l.  Returns false when a width
// does not qualify.  fl: floats of LDS the images take; per_wg: floats of one workgroup's gradient partials (dense).
inline bool snet_layout(const dcv_mlp* m, SnetLayer* ly, std::vector<int2>& tab, int* tab_begin, int& fl, int64_t& per_wg,
                        std::vector<int>* img_idx = nullptr) {
    if (img_idx) img_idx->assign((size_t)m->n_params, -1);
    fl = 0;
    per_wg = 0;
    int u4 = 0;
    for (int l = 0; l < m->L; ++l) {
        const LayerPlan& p = m->layers[l];
        SnetLayer& y = ly[l];
        y.in = p.in; y.out = p.out; y.act = p.act;
        auto pad = [](int w) { int k = 1; while (16 * k < w) k *= 2; return 16 * k; };
        y.pin = pad(p.in);
        y.pout = pad(p.out);
        if (y.pin > 256 || y.pout > 256) return false;
        y.nk_in = y.pin / 16;
        y.nk_out = y.pout / 16;
        y.u4_begin = u4;
        u4 += y.pout * (y.pin / 4);
        y.c4_shift = 0;
        while ((1 << y.c4_shift) < y.pin / 4) ++y.c4_shift;
        y.w_off = p.w_off; y.b_off = p.b_off;
        y.pws = y.pin + 4;
        y.lw = fl; fl += y.pout * y.pws;
        y.lb = fl; fl += y.pout;
        y.pw_off = y.pb_off = 0;
        y.pw_stride = (p.out * p.in + 3) / 4 * 4;
        y.pb_stride = (p.out + 3) / 4 * 4;
        per_wg += (int64_t)((p.out * p.in + 3) / 4 * 4) + (p.out + 3) / 4 * 4;
        if (tab_begin) tab_begin[l] = (int)tab.size();
        if (img_idx) {   // image float offset of every parameter of this layer
            for (int o = 0; o < p.out; ++o) {
                for (int i = 0; i < p.in; ++i) (*img_idx)[(size_t)(p.w_off + (int64_t)o * p.in + i)] = y.lw + o * y.pws + i;
                (*img_idx)[(size_t)(p.b_off + o)] = y.lb + o;
            }
        }
        const bool vec = (p.in % 4 == 0) && (p.w_off % 4 == 0);
        for (int o = 0; o < y.pout; ++o)
            for (int c = 0; c < y.pin / 4; ++c) {
                const int nv = o < p.out ? (p.in - 4 * c >= 4 ? 4 : (p.in - 4 * c > 0 ? p.in - 4 * c : 0)) : 0;
                tab.push_back(make_int2(nv > 0 ? (int)(p.w_off + (int64_t)o * p.in + 4 * c) : -1,
                                        (y.lw + o * y.pws + 4 * c) | (nv << 20) | ((vec && nv == 4 ? 1 : 0) << 24)));
            }
        for (int c = 0; c < y.pout / 4; ++c) {
            const int nv = p.out - 4 * c >= 4 ? 4 : (p.out - 4 * c > 0 ? p.out - 4 * c : 0);
            tab.push_back(make_int2(nv > 0 ? (int)(p.b_off + 4 * c) : -1, (y.lb + 4 * c) | (nv << 20) | ((nv == 4 && p.b_off % 4 == 0 ? 1 : 0) << 24)));
        }
    }
    return fl < (1 << 20) && m->n_params < (1ll << 31);
}

// Stages floats [f0, f1) of the global weight image (same layout as the LDS image; f0, f1 multiples of 256 floats are not
// required) into LDS with global_load_lds: no table, no VGPR, every copy of the workgroup in flight at once -- ONE round trip
// where the table-driven staging took two per pass of 12 units.  Completion: the caller's s_waitcnt vmcnt(0) + barrier.
template <int NT>
__device__ __forceinline__ void snet_stage_image(const float* __restrict__ img, float* sl, int f0, int f1, int t) {
    const int u0 = f0 >> 2, u1 = (f1 + 3) >> 2;
    for (int ub = u0 + (t & ~63); ub < u1; ub += NT) {   // a wave-instruction covers 64 consecutive 16-byte units
        const int u = ub + (t & 63);
        const unsigned ldsw = lds_addr_uniform(sl + 4 * ub);
        if (u < u1) glds16(img + 4 * u, ldsw);
    }
}

inline bool snet_disabled() {
    static const bool off = [] { const char* e = getenv("DCV_NO_SNET"); return e && e[0] == '1'; }();
    return off;
}

}  // namespace dcv
