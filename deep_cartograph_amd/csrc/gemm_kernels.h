// __global__ wrappers, epilogues and launch helpers around the block-tile engine of gemm.h.
#pragma once
#include <type_traits>
#include <stdlib.h>
#include <hip/hip_ext.h>
#include "gemm.h"

namespace dcv {

// ------------------------------------------------------------------ activations
constexpr float kSoftplus0 = 0.6931471824645996f;   // F.softplus(torch.zeros(1)).item(): float32 log1p(exp(0))
__device__ __forceinline__ float act_fwd(int act, float z) {
    switch (act) {
        case DCV_ACT_LEAKY_RELU: return z > 0.f ? z : 0.01f * z;
        case DCV_ACT_RELU: return z > 0.f ? z : 0.f;
        case DCV_ACT_TANH: return tanhf(z);
        case DCV_ACT_ELU: return z > 0.f ? z : expm1f(z);
        case DCV_ACT_SOFTPLUS: return z > 20.f ? z : log1pf(expf(z));
        case DCV_ACT_SHIFTED_SOFTPLUS: return (z > 20.f ? z : log1pf(expf(z))) - kSoftplus0;   // mlcolvar Shifted_Softplus
        case DCV_ACT_CUSTOM_SIGMOID: return 1.f / (1.f + expf(-3.f * z));                       // mlcolvar Custom_Sigmoid(p=3)
        default: return z;
    }
}
// derivative expressed through the stored post-activation value h = act(z)
__device__ __forceinline__ float act_grad_from_out(int act, float h) {
    switch (act) {
        case DCV_ACT_LEAKY_RELU: return h > 0.f ? 1.f : 0.01f;
        case DCV_ACT_RELU: return h > 0.f ? 1.f : 0.f;
        case DCV_ACT_TANH: return 1.f - h * h;
        case DCV_ACT_ELU: return h > 0.f ? 1.f : h + 1.f;
        case DCV_ACT_SOFTPLUS: return h > 20.f ? 1.f : 1.f - expf(-h);
        case DCV_ACT_SHIFTED_SOFTPLUS: return 1.f - expf(-(h + kSoftplus0));   // sigmoid(z) = 1 - exp(-softplus(z))
        case DCV_ACT_CUSTOM_SIGMOID: return 3.f * h * (1.f - h);
        default: return 1.f;
    }
}

// ------------------------------------------------------------------ dropout
// torch.nn.Dropout(p) in training mode: y = x * keep / (1 - p), keep ~ Bernoulli(1 - p) per element.  The keep bit
// of element (row, col) of layer `layer` at training step `step` is bit-reproducible from a counter-based generator
// (Philox-4x32, 7 rounds; one call covers the 4 columns of a 16-byte row segment): nothing is stored, the backward
// pass recomputes the mask of the forward.  (The reference draws its masks from torch's CPU generator -- a different
// stream; parity is exact GIVEN the mask, which dcv_mlp_dropout_mask hands to the tests.)
__device__ __forceinline__ uint4 philox4x32_7(uint4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 7; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
        c = make_uint4(hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}
struct DropCfg {
    uint32_t thr;     // keep iff draw >= thr ; thr = p * 2^32 ; 0 = dropout off
    float scale;      // 1 / (1 - p)
    uint32_t k0, k1;  // seed
    uint32_t layer, step;
    // keep * scale multipliers of columns col .. col + 3 (col % 4 == 0) of row `row`
    __device__ __forceinline__ float4 mult(int64_t row, int64_t col) const {
        const uint4 u = philox4x32_7(make_uint4((uint32_t)row, (uint32_t)(col >> 2), layer, step), k0, k1);
        return make_float4(u.x >= thr ? scale : 0.f, u.y >= thr ? scale : 0.f, u.z >= thr ? scale : 0.f, u.w >= thr ? scale : 0.f);
    }
    template <int N>
    __device__ __forceinline__ void apply(float4 (&v)[N], int64_t row0, int row_step, int64_t col) const {
#pragma unroll
        for (int q = 0; q < N; ++q) {
            const float4 k = mult(row0 + (int64_t)q * row_step, col);
            v[q].x *= k.x; v[q].y *= k.y; v[q].z *= k.z; v[q].w *= k.w;
        }
    }
};
constexpr DropCfg kNoDrop{0u, 1.f, 0u, 0u, 0u, 0u};

// ------------------------------------------------------------------ epilogues
// The activation switch is hoisted out of the element loops: one uniform branch per workgroup.
template <int N, class F>
__device__ __forceinline__ void map_quads(float4 (&v)[N], F f) {
#pragma unroll
    for (int q = 0; q < N; ++q) {
        v[q].x = f(v[q].x, 0, q);
        v[q].y = f(v[q].y, 1, q);
        v[q].z = f(v[q].z, 2, q);
        v[q].w = f(v[q].w, 3, q);
    }
}
__device__ __forceinline__ float f4c(const float4& v, int c) { return c == 0 ? v.x : (c == 1 ? v.y : (c == 2 ? v.z : v.w)); }

struct EpiStore {  // C = acc
    static constexpr bool kDrop = false;
    static constexpr bool kColSum = false;
    static constexpr bool kSide = false;
    static constexpr bool kHead = false;
    static constexpr bool kMaskOut = false;
    static constexpr bool kMaskIn = false;
    float* C;
    int64_t ldc;
    bool vec;
    __device__ __forceinline__ float4 colconst(int64_t, int) const { return make_float4(0.f, 0.f, 0.f, 0.f); }
    template <int N>
    __device__ __forceinline__ void transform(float4 (&)[N], const float4 (&)[1], const float4&) const {}
    __device__ __forceinline__ float* out_ptr(int, int64_t r, int64_t c) const { return C + r * ldc + c; }
    __device__ __forceinline__ void one(int, int64_t r, int64_t c, float v) const { C[r * ldc + c] = v; }
};
struct EpiBiasAct {  // H = dropout(act(acc + bias[col]))
    static constexpr bool kDrop = true;
    static constexpr bool kColSum = false;
    static constexpr bool kSide = false;
    static constexpr bool kHead = false;
    static constexpr bool kMaskOut = true;
    static constexpr bool kMaskIn = false;
    float* C;
    int64_t ldc;
    const float* bias;
    int act;
    bool vec;
    unsigned long long* mask = nullptr;   // optional sign mask of H (ReLU family), see store_sign_mask
    DropCfg drop = kNoDrop;
    __device__ __forceinline__ float4 colconst(int64_t c, int nvalid) const {
        return bias ? load_quad(bias + c, nvalid, vec) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    template <int N>
    __device__ __forceinline__ void transform(float4 (&v)[N], const float4 (&)[1], const float4& b) const {
        switch (act) {
            case DCV_ACT_LEAKY_RELU: map_quads<N>(v, [&](float x, int c, int) { const float z = x + f4c(b, c); return z > 0.f ? z : 0.01f * z; }); break;
            case DCV_ACT_RELU: map_quads<N>(v, [&](float x, int c, int) { const float z = x + f4c(b, c); return z > 0.f ? z : 0.f; }); break;
            case DCV_ACT_NONE: map_quads<N>(v, [&](float x, int c, int) { return x + f4c(b, c); }); break;
            default: map_quads<N>(v, [&](float x, int c, int) { return act_fwd(act, x + f4c(b, c)); }); break;
        }
    }
    __device__ __forceinline__ float* out_ptr(int, int64_t r, int64_t c) const { return C + r * ldc + c; }
    __device__ __forceinline__ void one(int, int64_t r, int64_t c, float v) const {
        C[r * ldc + c] = act_fwd(act, v + (bias ? bias[c] : 0.f));
    }
};
// Butterfly reduce-scatter of CUR values over the C4 consecutive lanes that share the rows of a thread
// group: while the count is even each step halves it (a lane keeps the half selected by its lane bit and
// adds the partner's copy of that half), odd counts fall back to an all-reduce step.  Afterwards x[i],
// i < the returned count, holds the full sum of original entry base + i; lanes that differ only in the
// bits of `dup` hold copies.
template <int CUR, int OFF, int MAXN, class T = float>
__device__ __forceinline__ int butterfly_sum(T (&x)[MAXN], int lane, int& base, int& dup) {
    if constexpr (OFF >= 1) {
        if constexpr (CUR % 2 == 0) {
            constexpr int H = CUR / 2;
            const bool up = (lane & OFF) != 0;
#pragma unroll
            for (int i = 0; i < H; ++i) {
                const T send = up ? x[i] : x[i + H];
                const T keep = up ? x[i + H] : x[i];
                x[i] = keep + __shfl_xor(send, OFF, 64);
            }
            base += up ? H : 0;
            return butterfly_sum<H, OFF / 2, MAXN, T>(x, lane, base, dup);
        } else {
#pragma unroll
            for (int i = 0; i < CUR; ++i) x[i] += __shfl_xor(x[i], OFF, 64);
            dup |= OFF;
            return butterfly_sum<CUR, OFF / 2, MAXN, T>(x, lane, base, dup);
        }
    } else {
        return CUR;
    }
}

// H = act(acc + bias) as EpiBiasAct, and -- fused -- the narrow Linear that follows it:
// Z[r][j] = act2(sum_c H[r][c] W2[j][c] + b2[j]), j < D <= DMAX, for a product whose N columns fit one
// column tile (the whole row of H is in this workgroup).  Saves streaming H once more for 2*D flop per
// element: a thread's 4-column partial dot products are summed over the lanes that share its rows.
template <int DMAX>
struct EpiBiasActHead {
    static constexpr bool kDrop = true;
    static constexpr bool kColSum = false;
    static constexpr bool kSide = false;
    static constexpr bool kHead = true;
    static constexpr bool kMaskOut = false;
    static constexpr bool kMaskIn = false;
    float* C;
    int64_t ldc;
    const float* bias;
    int act;
    bool vec;
    const float* W2;   // [D][ldw]
    int64_t ldw;
    const float* b2;   // [D] or null
    int D, act2;
    float* Z;          // [M][ldz]
    int64_t ldz;
    DropCfg drop = kNoDrop;
    __device__ __forceinline__ float4 colconst(int64_t c, int nvalid) const {
        return bias ? load_quad(bias + c, nvalid, vec) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    template <int N>
    __device__ __forceinline__ void transform(float4 (&v)[N], const float4 (&)[1], const float4& b) const {
        switch (act) {
            case DCV_ACT_LEAKY_RELU: map_quads<N>(v, [&](float x, int c, int) { const float z = x + f4c(b, c); return z > 0.f ? z : 0.01f * z; }); break;
            case DCV_ACT_RELU: map_quads<N>(v, [&](float x, int c, int) { const float z = x + f4c(b, c); return z > 0.f ? z : 0.f; }); break;
            case DCV_ACT_NONE: map_quads<N>(v, [&](float x, int c, int) { return x + f4c(b, c); }); break;
            default: map_quads<N>(v, [&](float x, int c, int) { return act_fwd(act, x + f4c(b, c)); }); break;
        }
    }
    // v[q]: columns col..col+3 (nvalid of them in range) of row row0 + q * row_step, after transform()
    template <int N, int C4>
    __device__ __forceinline__ void head(const float4 (&v)[N], int64_t row0, int row_step, int64_t col, int nvalid, int64_t M,
                                         int lane) const {
        static_assert((C4 & (C4 - 1)) == 0 && C4 <= 32, "lanes per row group");
        float x[N * DMAX];
#pragma unroll
        for (int j = 0; j < DMAX; ++j) {
            const float4 w = (j < D && nvalid > 0) ? load_quad(W2 + (int64_t)j * ldw + col, nvalid, vec) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int q = 0; q < N; ++q) x[q * DMAX + j] = fmaf(v[q].x, w.x, fmaf(v[q].y, w.y, fmaf(v[q].z, w.z, v[q].w * w.w)));
        }
        int base = 0, dup = 0;
        const int cnt = butterfly_sum<N * DMAX, C4 / 2, N * DMAX>(x, lane, base, dup);
        if ((lane & dup) != 0) return;
#pragma unroll
        for (int i = 0; i < N * DMAX; ++i) {
            if (i < cnt) {
                const int f = base + i, q = f / DMAX, j = f - q * DMAX;
                const int64_t row = row0 + (int64_t)q * row_step;
                if (j < D && row < M) Z[row * ldz + j] = act_fwd(act2, x[i] + (b2 ? b2[j] : 0.f));
            }
        }
    }
    __device__ __forceinline__ float* out_ptr(int, int64_t r, int64_t c) const { return C + r * ldc + c; }
    __device__ __forceinline__ void one(int, int64_t, int64_t, float) const {}
};
struct EpiActGrad {  // dZ = acc * act'(H) [* dropout mask of H] ; column sums of dZ per row tile -> bias-gradient partials
    static constexpr bool kDrop = true;
    static constexpr bool kColSum = true;
    static constexpr bool kSide = true;
    static constexpr bool kHead = false;
    static constexpr bool kMaskOut = false;
    static constexpr bool kMaskIn = true;
    float* C;
    int64_t ldc;
    const float* H;
    int64_t ldh;
    int act;
    float* bpart;  // [tiles_m][n]
    int64_t n;
    bool vec;
    const unsigned long long* mask = nullptr;   // sign mask of H written by the forward of the same tile shape: H is not read
    float slope = 0.f;                          // derivative where H <= 0 (0.01 leaky ReLU, 0 ReLU)
    DropCfg drop = kNoDrop;                     // dropout of the layer that produced H: the stored H is act(z) * keep / (1 - p)
    float hscale = 1.f;                         // 1 - p: stored kept value -> act(z)
    template <int N>
    __device__ __forceinline__ void apply_mask(float4 (&v)[N], int64_t seg0, int wave, int lane) const {
#pragma unroll
        for (int q = 0; q < N; ++q) {
            const unsigned long long* w = mask + ((seg0 + q) * 4 + wave) * 4;
            if (!((w[0] >> lane) & 1ull)) v[q].x *= slope;
            if (!((w[1] >> lane) & 1ull)) v[q].y *= slope;
            if (!((w[2] >> lane) & 1ull)) v[q].z *= slope;
            if (!((w[3] >> lane) & 1ull)) v[q].w *= slope;
        }
    }
    __device__ __forceinline__ float4 colconst(int64_t, int) const { return make_float4(0.f, 0.f, 0.f, 0.f); }
    __device__ __forceinline__ const float* side_ptr(int64_t r, int64_t c) const { return H + r * ldh + c; }
    template <int N>
    __device__ __forceinline__ void transform(float4 (&v)[N], const float4 (&h)[N], const float4&) const {
        switch (act) {
            case DCV_ACT_LEAKY_RELU: map_quads<N>(v, [&](float x, int c, int q) { return f4c(h[q], c) > 0.f ? x : 0.01f * x; }); break;
            case DCV_ACT_RELU: map_quads<N>(v, [&](float x, int c, int q) { return f4c(h[q], c) > 0.f ? x : 0.f; }); break;
            case DCV_ACT_NONE: break;
            default: map_quads<N>(v, [&](float x, int c, int q) { return x * act_grad_from_out(act, f4c(h[q], c) * hscale); }); break;
        }
    }
    __device__ __forceinline__ float* out_ptr(int, int64_t r, int64_t c) const { return C + r * ldc + c; }
    __device__ __forceinline__ void one(int, int64_t, int64_t, float) const {}
    __device__ __forceinline__ void colsum(int tile_m, int64_t c, float v) const { bpart[(int64_t)tile_m * n + c] = v; }
};
struct EpiSlab {  // split-K partials: slab[z][which][M][N]
    static constexpr bool kDrop = false;
    static constexpr bool kColSum = false;
    static constexpr bool kSide = false;
    static constexpr bool kHead = false;
    static constexpr bool kMaskOut = false;
    static constexpr bool kMaskIn = false;
    float* slab;
    int64_t M, N;
    int nb;
    int64_t z;
    bool vec;
    int64_t cap = 0;   // slabs of nb * M * N floats the buffer holds: checked against the split count at launch
    __device__ __forceinline__ float4 colconst(int64_t, int) const { return make_float4(0.f, 0.f, 0.f, 0.f); }
    template <int NN>
    __device__ __forceinline__ void transform(float4 (&)[NN], const float4 (&)[1], const float4&) const {}
    __device__ __forceinline__ float* out_ptr(int which, int64_t r, int64_t c) const { return slab + ((z * nb + which) * M + r) * N + c; }
    __device__ __forceinline__ void one(int which, int64_t r, int64_t c, float v) const {
        slab[((z * nb + which) * M + r) * N + c] = v;
    }
};

// EpiSlab + the column sums of the A operand per contraction chunk: asum[z][M] (gemm.h: kASum)
struct EpiSlabSum : EpiSlab {
    static constexpr bool kASum = true;
    float* asum = nullptr;
};

inline bool quad_ok(const void* p, int64_t ld) { return (ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(p) & 15) == 0); }

// ------------------------------------------------------------------ kernels
// XCD-aware block -> (tile, split) map.  Blocks b and b + 8 are observed to share an XCD (and its
// L2); the workgroups that read the same rows -- the column tiles of one row tile (NT / NN), the
// output tiles of one contraction chunk (TN) -- are therefore given ids that are congruent mod 8.
// Placement is a speed heuristic only: any dispatch order computes the same result.
// `lin` is the linear workgroup index of the product: blockIdx.x for the row-tiled forms (NT / NN, tail chunks
// behind the regular tiles), blockIdx.z * tiles + blockIdx.x for the split-K form (TN).
struct BlockMap {
    int tile_m, tile_n, tail_chunk, split;
    int64_t k_begin, k_end;
};
template <int MODE>
__device__ __forceinline__ BlockMap map_block(const GemmDims& d, int lin) {
    BlockMap bm;
    bm.tail_chunk = -1;
    bm.split = 0;
    bm.k_begin = 0;
    bm.k_end = d.K;
    if constexpr (MODE == kTN) {
        const int T = d.tiles_m * d.tiles_n;
        int tile = lin % T, split = lin / T;
        if (d.xcd_remap) {
            const int xcd = lin & 7, q = lin >> 3;
            tile = q % T;
            split = (q / T) * 8 + xcd;
        }
        bm.tile_m = tile / d.tiles_n;
        bm.tile_n = tile - bm.tile_m * d.tiles_n;
        bm.split = split;
        bm.k_begin = (int64_t)split * d.k_chunk;
        bm.k_end = bm.k_begin + d.k_chunk < d.K ? bm.k_begin + d.k_chunk : d.K;
    } else {
        const int tile = lin;
        const int regular = d.tail_split > 0 ? (d.tiles_m - 1) * d.tiles_n : d.tiles_m * d.tiles_n;
        if (tile >= regular) {   // a contraction chunk of the ragged last row tile
            const int j = tile - regular;
            bm.tile_m = d.tiles_m - 1;
            bm.tile_n = j % d.tiles_n;
            bm.tail_chunk = j / d.tiles_n;
            bm.k_begin = (int64_t)bm.tail_chunk * d.k_chunk;
            bm.k_end = bm.k_begin + d.k_chunk < d.K ? bm.k_begin + d.k_chunk : d.K;
        } else if (d.xcd_remap) {
            const int xcd = tile & 7, q = tile >> 3;
            bm.tile_n = q % d.tiles_n;
            bm.tile_m = (q / d.tiles_n) * 8 + xcd;
        } else {
            bm.tile_m = tile / d.tiles_n;
            bm.tile_n = tile - bm.tile_m * d.tiles_n;
        }
    }
    return bm;
}

// grid.x = tiles_m * tiles_n (tile_n fastest; + tail chunks), grid.z = k splits (TN only)
template <int MODE, class Cfg, int NB, bool VEC, bool GATHER, class Epi>
__global__ __launch_bounds__(256, 2) void gemm_kernel(Operand A, Operand B, int64_t lag2, GemmDims d, Epi epi) {
    extern __shared__ __attribute__((aligned(16))) float lds_f[];
    const int lin = MODE == kTN ? (int)(blockIdx.z * (unsigned)(d.tiles_m * d.tiles_n) + blockIdx.x) : (int)blockIdx.x;
    const BlockMap bm = map_block<MODE>(d, lin);
    if constexpr (MODE == kTN && std::is_base_of<EpiSlab, Epi>::value) epi.z = bm.split;
    gemm_block<MODE, Cfg, NB, VEC, GATHER, Epi>(A, B, lag2, d, bm.tile_m, bm.tile_n, bm.k_begin, bm.k_end, lds_f, epi, bm.tail_chunk);
}

// Two independent products in ONE launch (blocks [0, blocks1) run the first, the rest the second): the weight gradient
// (TN, split-K slabs) and the input gradient (NN) of a layer both read the same dZ and neither reads the other's
// output; as two launches of a small-batch step each leaves part of the chip idle (256 + 516 workgroups at 8202 rows)
// and pays its own launch ramp and drain.
template <class Cfg1, class Cfg2>
__global__ __launch_bounds__(256, 2) void wgrad_dgrad_kernel(Operand A1, Operand B1, GemmDims d1, EpiSlab e1, int blocks1, Operand A2, Operand B2,
                                                           GemmDims d2, EpiActGrad e2) {
    extern __shared__ __attribute__((aligned(16))) float lds_f[];
    if ((int)blockIdx.x < blocks1) {
        const BlockMap bm = map_block<kTN>(d1, (int)blockIdx.x);
        e1.z = bm.split;
        gemm_block<kTN, Cfg1, 1, true, false, EpiSlab>(A1, B1, 0, d1, bm.tile_m, bm.tile_n, bm.k_begin, bm.k_end, lds_f, e1, -1);
    } else {
        const BlockMap bm = map_block<kNN>(d2, (int)blockIdx.x - blocks1);
        gemm_block<kNN, Cfg2, 1, true, false, EpiActGrad>(A2, B2, 0, d2, bm.tile_m, bm.tile_n, bm.k_begin, bm.k_end, lds_f, e2, bm.tail_chunk);
    }
}

// Tile configurations, each in two arithmetic flavours (template argument S): the FP32-input MFMA
// (S = false) and FP32-accurate split products on the BF16 matrix pipe (S = true, gemm.h: split3 / mfma16);
// gemm_split() picks at launch time (dcv_set_gemm_mode / DCV_GEMM_MODE).
template <bool S> using CfgBigT = TileCfg<2, 2, 2, 2, 32, 2, S>;      // 128 x 128
template <bool S> using CfgHalfMT = TileCfg<2, 2, 1, 2, 32, 2, S>;    // 64 x 128: twice the workgroups when the row count is small
template <bool S> using CfgQuarterT = TileCfg<2, 2, 1, 1, 32, 2, S>;   // 64 x 64: split-K products whose 128 x 128 grid would leave most CUs idle
template <bool S> using CfgNarrowNT = TileCfg<4, 1, 1, 1, 32, 2, S>;  // 128 x 32
template <bool S> using CfgNarrowMT = TileCfg<1, 4, 1, 1, 32, 2, S>;  // 32 x 128
template <bool S> using CfgCovT = TileCfg<2, 2, 2, 2, 16, 2, S>;      // 128 x 128, two B operands, 48 KiB LDS
#ifdef DCV_BIGCFG
using CfgBig = DCV_BIGCFG;   // diagnostic override (tools/gemm_bench)
#else
using CfgBig = CfgBigT<false>;
#endif

template <int MODE, class Cfg, int NB, bool VEC, bool GATHER, class Epi>
static int launch_gemm_vec(const Operand& A, const Operand& B, int64_t lag2, const GemmDims& d, int64_t splits,
                           const Epi& epi, hipStream_t s) {
#ifdef DCV_FORCE_LDS
    constexpr size_t lds = DCV_FORCE_LDS;   // diagnostic: limit co-residency through the LDS request
#else
    constexpr size_t lds = gemm_lds_bytes<Cfg, NB>();
#endif
    auto kern = gemm_kernel<MODE, Cfg, NB, VEC, GATHER, Epi>;
    if (lds > 64 * 1024) {
        static bool attr_set = false;  // per instantiation
        if (!attr_set) {
            DCV_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_set = true;
        }
    }
    const int64_t blocks = d.tail_split > 0 ? (int64_t)(d.tiles_m - 1 + d.tail_split) * d.tiles_n : (int64_t)d.tiles_m * d.tiles_n;
    if (g_launch_ev.start != nullptr) {   // a profiled launch: the events carry the kernel's own begin / end (common.h)
        const LaunchEvents ev = g_launch_ev;
        g_launch_ev = LaunchEvents{};
        g_launch_taken = ev.start;
        hipExtLaunchKernelGGL(kern, dim3((unsigned)blocks, 1, (unsigned)splits), dim3(256), (uint32_t)lds, s, ev.start, ev.stop, 0u, A, B, lag2, d, epi);
    } else {
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks, 1, (unsigned)splits), dim3(256), lds, s, A, B, lag2, d, epi);
    }
    DCV_CHECK_LAUNCH();
    return DCV_OK;
}

// Workspace of the contraction-split tail tile (GemmDims::tail_split): owned by the caller, one per stream of launches
struct TailWs {
    float* ws = nullptr;       // cap floats
    unsigned* cnt = nullptr;   // max_tiles_n zero-initialised tickets
    int64_t cap = 0;
    int max_tiles_n = 0;
};
#ifndef DCV_TAIL_MAXSPLIT
#define DCV_TAIL_MAXSPLIT 4   // measured at 8202 x 256 x 512: 2 chunks 26.5 us, 3: 25.8, 4: 25.5, 6: 25.4, 8: 26-28, 16: 34
#endif
constexpr int kTailMaxSplit = DCV_TAIL_MAXSPLIT;
// Workspace of the hand-off between the chunk workgroups of a tail tile (handoff.h): ordinary device memory, the tickets
// zeroed once here (the last arriver of every launch leaves them zero).  On failure the TailWs stays empty (= tail cut disabled).
inline bool alloc_tail_ws(TailWs* tw, int max_tiles_n) {
    *tw = TailWs{};
    const int64_t cap = (int64_t)max_tiles_n * kTailMaxSplit * 64 * 256;   // a 128 x 128 tile's accumulators per chunk
    void *ws = nullptr, *cnt = nullptr;
    if (hipMalloc(&ws, (size_t)cap * sizeof(float)) != hipSuccess ||
        hipMalloc(&cnt, (size_t)(max_tiles_n + 3) / 4 * 4 * sizeof(unsigned)) != hipSuccess ||
        hipMemset(cnt, 0, (size_t)max_tiles_n * sizeof(unsigned)) != hipSuccess) {
        (void)hipGetLastError();
        if (ws) (void)hipFree(ws);
        if (cnt) (void)hipFree(cnt);
        return false;
    }
    tw->ws = static_cast<float*>(ws);
    tw->cnt = static_cast<unsigned*>(cnt);
    tw->cap = cap;
    tw->max_tiles_n = max_tiles_n;
    return true;
}
#ifndef DCV_TAIL_STAGES_MIN
#define DCV_TAIL_STAGES_MIN 8   // contraction stages below which the tail tile stays whole
#endif
#ifndef DCV_TAIL_MINSTAGES
#define DCV_TAIL_MINSTAGES 2   // stages per tail chunk
#endif
inline bool tail_split_enabled() {
    static const bool on = [] { const char* e = getenv("DCV_TAIL_KSPLIT"); return !(e && e[0] == '0'); }();
    return on;
}

// Launch geometry of one product: tile counts, XCD map, tail cut, split count, loader flavour.
// tiles_m_out (optional) receives the number of row tiles (= bias partial blocks of EpiActGrad)
struct GemmPlan {
    GemmDims d;
    int64_t splits;
    bool vec, gather;
};
template <int MODE, class Cfg, int NB, class Epi>
static int prepare_gemm(const Operand& A, const Operand& B, int64_t M, int64_t N, int64_t K, int64_t k_chunk, const Epi& epi,
                        int* tiles_m_out, const TailWs* tw, GemmPlan* plan) {
    GemmDims& d = plan->d;
    d = GemmDims{};
    d.M = M;
    d.N = N;
    d.K = K;
    d.k_chunk = k_chunk > 0 ? k_chunk : K;
    d.tiles_m = (int)cdiv(M, Cfg::TM);
    d.tiles_n = (int)cdiv(N, Cfg::TN);
    if (tiles_m_out) *tiles_m_out = d.tiles_m;
    {
        static int env = -1;
        if (env < 0) { const char* e = getenv("DCV_XCD_REMAP"); env = e ? atoi(e) : 1; }
        const int64_t nsplit = (MODE == kTN) ? cdiv(K, d.k_chunk) : 1;
        // Outputs up to 256 MB leave the kernel as write-through stores: the next launch (another set of XCDs) reads them
        // from memory either way, and a launch that ends without dirty lines skips the L2 write-back in front of its
        // successor (contract batch: 105.0 -> 101.9 us per step; neutral at the large batch and on the covariance slabs,
        // which stay write-back).  DCV_WT=0 restores plain stores.
        static const int wt_env = [] { const char* e = getenv("DCV_WT"); return e ? atoi(e) : 1; }();
        const int64_t out_bytes = nsplit * M * N * (int64_t)sizeof(float) * NB;
        d.wt = (wt_env != 0 && out_bytes <= (256ll << 20)) ? 1 : 0;
        // bijective only when the remapped index is a multiple of 8
        d.xcd_remap = env && ((MODE == kTN) ? (nsplit % 8 == 0) : (d.tiles_m % 8 == 0 && d.tiles_n > 1));
        if constexpr (MODE != kTN && NB == 1) {
            // Ragged last row tile: worth cutting along the contraction when it is what pushes the workgroup count past a
            // multiple of the CU count on a small grid (8202 rows in 64-row tiles x 2 column tiles = 258 workgroups on
            // 256 CUs: two CUs would run two full-length workgroups on the same SIMDs, 1.4 x the launch time).
            const int64_t stages = K / Cfg::KB, ncu = num_cus();
            const int64_t all = (int64_t)d.tiles_m * d.tiles_n, regular = all - d.tiles_n;
            if (tw && tw->ws && tail_split_enabled() && M % Cfg::TM != 0 && d.tiles_m >= 2 && K % Cfg::KB == 0 && stages >= DCV_TAIL_STAGES_MIN &&   // shorter contractions: measured slower
                all <= 4 * ncu && cdiv(all, ncu) > cdiv(regular, ncu) && d.tiles_n <= tw->max_tiles_n) {
                const int64_t want = stages / DCV_TAIL_MINSTAGES < kTailMaxSplit ? stages / DCV_TAIL_MINSTAGES : kTailMaxSplit;
                const int64_t kc = cdiv(stages, want) * Cfg::KB;
                const int64_t S = cdiv(K, kc);
                if (S >= 2 && (int64_t)d.tiles_n * S * Cfg::FM * Cfg::FN * 16 * 256 <= tw->cap) {
                    d.tail_split = (int)S;
                    d.k_chunk = kc;
                    d.tail_ws = tw->ws;
                    d.tail_cnt = tw->cnt;
                    d.xcd_remap = env && ((d.tiles_m - 1) % 8 == 0 && d.tiles_n > 1);
                }
            }
        }
    }
    plan->splits = (MODE == kTN) ? cdiv(K, d.k_chunk) : 1;
    const int64_t tiles = (int64_t)d.tiles_m * d.tiles_n;
    DCV_REQUIRE(tiles > 0 && tiles < (1ll << 31) && plan->splits > 0 && plan->splits < 65536, "gemm: grid out of range (tiles=%lld splits=%lld)",
                (long long)tiles, (long long)plan->splits);
    DCV_REQUIRE(!(Cfg::SPLIT && (A.shift || B.shift)), "gemm: a column shift needs the FP32-input MFMA flavour");
    if constexpr (std::is_base_of<EpiSlab, Epi>::value) {
        // every split writes its own slab: refuse the launch instead of writing past the caller's buffer
        if (plan->splits > epi.cap) {
            set_error("gemm: %lld split-K slabs needed, the slab buffer holds %lld", (long long)plan->splits, (long long)epi.cap);
            return DCV_ENOMEM;
        }
    }
    // 16-byte loads: aligned operands; contraction-contiguous (MMAJOR) operands also need K % 4 == 0
    constexpr bool A_MM = (MODE == kNT || MODE == kNN), B_MM = (MODE == kNT);
    plan->vec = A.vec_ok && B.vec_ok && (!A_MM || K % 4 == 0) && (!B_MM || K % 4 == 0);
    plan->gather = A.rows.idx != nullptr || B.rows.idx != nullptr;
    return DCV_OK;
}
template <int MODE, class Cfg, int NB, class Epi>
static int launch_gemm_cfg(const Operand& A, const Operand& B, int64_t lag2, int64_t M, int64_t N, int64_t K,
                           int64_t k_chunk, const Epi& epi, hipStream_t s, int* tiles_m_out = nullptr, const TailWs* tw = nullptr) {
    GemmPlan pl;
    const int rc = prepare_gemm<MODE, Cfg, NB, Epi>(A, B, M, N, K, k_chunk, epi, tiles_m_out, tw, &pl);
    if (rc) return rc;
    if (pl.vec && !pl.gather) return launch_gemm_vec<MODE, Cfg, NB, true, false, Epi>(A, B, lag2, pl.d, pl.splits, epi, s);
    if (pl.vec) return launch_gemm_vec<MODE, Cfg, NB, true, true, Epi>(A, B, lag2, pl.d, pl.splits, epi, s);
    return launch_gemm_vec<MODE, Cfg, NB, false, true, Epi>(A, B, lag2, pl.d, pl.splits, epi, s);   // scalar loads: gather-capable form
}

// Picks the tile shape from the output extents.  Row-parallel products (NT / NN) fall back to
// shorter tiles when 128-row tiles would leave CUs without two resident workgroups (small per-GPU
// batches of a multi-GPU run); TN products get their parallelism from the split count instead.
enum CfgPick { kPickNarrowN, kPickNarrowM, kPickHalfM, kPickQuarter, kPickBig };
template <int MODE, bool HEAD>
static CfgPick pick_cfg(int64_t M, int64_t N, int64_t K, int64_t k_chunk) {
#ifdef DCV_FORCE_BIG   // diagnostic (tools/gemm_bench): every product takes the DCV_BIGCFG tile, whatever its extents
    return kPickBig;
#endif
    if (N <= 32) return kPickNarrowN;
    if (M <= 32) return kPickNarrowM;
    if constexpr (MODE != kTN) {
        const int64_t want = 2 * (int64_t)num_cus();
        const int64_t tn = cdiv(N, 128);
        if (cdiv(M, 128) * tn < want) {
            if (cdiv(M, 64) * tn >= want || M <= 64 * 4) return kPickHalfM;
            if (cdiv(M, 64) * tn < want / 2) return kPickNarrowM;
            // between one and two 64 x 128 workgroups per CU (8202 rows x 256 columns: 258): 64 x 64 tiles put two waves on
            // every SIMD, which overlap each other's split and MFMA phases (measured 22.8 -> 20.8 us at 8192 x 256 x 512)
            if constexpr (!HEAD) {
                static const bool quarter_off = [] { const char* e = getenv("DCV_NO_QUARTER_NT"); return e && e[0] == '1'; }();
                if (!quarter_off && N % 64 == 0) return kPickQuarter;
            }
            return kPickHalfM;
        }
    }
    if constexpr (MODE == kTN) {
        // split-K product on a small grid (weight gradients of a small batch: the split count is bounded by the rows):
        // four times the workgroups with 64 x 64 tiles (measured on the 128 x 256 x 8202 product: 17.2 -> 10.6 us)
        const int64_t nsplit = cdiv(K, k_chunk > 0 ? k_chunk : K);
        // (measured on the 256 x 512 x 8202 product: 16 chunks, 128 workgroups of 128 x 128 32.5 us, 512 of 64 x 64 21.9 us)
        if (cdiv(M, 128) * cdiv(N, 128) * nsplit <= (int64_t)num_cus() / 2) return kPickQuarter;
    }
    return kPickBig;
}
// Split-from-LDS configurations (TileCfg::PL bit 4): NT products with 16-byte rows, no gathered rows and K a multiple of the
// stage depth; returns 1 when the product does not qualify (the caller takes the in-register split).
template <class Cfg, class Epi>
static int launch_gemm_ls(const Operand& A, const Operand& B, int64_t M, int64_t N, int64_t K, const Epi& epi, hipStream_t s,
                          int* tiles_m_out, const TailWs* tw) {
    if (K % Cfg::KB != 0 || A.shift || B.shift) return 1;
    GemmPlan pl;
    int rc = prepare_gemm<kNT, Cfg, 1, Epi>(A, B, M, N, K, 0, epi, tiles_m_out, tw, &pl);
    if (rc) return rc;
    if (!pl.vec || pl.gather) return 1;
    return launch_gemm_vec<kNT, Cfg, 1, true, false, Epi>(A, B, 0, pl.d, pl.splits, epi, s);
}

template <int MODE, bool S, class Epi>
static int launch_gemm_mode(const Operand& A, const Operand& B, int64_t M, int64_t N, int64_t K, int64_t k_chunk,
                            const Epi& epi, hipStream_t s, int* tiles_m_out, const TailWs* tw = nullptr) {
#ifdef DCV_BIGCFG
    using Big = CfgBig;
#else
    using Big = CfgBigT<S>;
#endif
    const CfgPick pick = pick_cfg<MODE, Epi::kHead>(M, N, K, k_chunk);
#ifdef DCV_PS_EXPERIMENT   // tools/planes_bench only: measured SLOWER than the LDS-DMA ring + in-register split
    if constexpr (S && MODE != kTN) {
        // Small grids of the split flavour (the row-parallel products of a small batch: one or two workgroups per CU) are
        // bound by the vector ALU's operand splitting: stage through registers and split once per workgroup on the way to
        // LDS (TileCfg::PL bits 2 / 3) -- both operands of an NT product, the A operand of an NN product.  Bit-identical
        // results, half the splitting work -- and 25.2 vs 22.4 us at 8192 x 256 x 512, 12.0 vs 10.5 us at 8202 x 128 x 256:
        // the register-staged loop exposes the load latency the DMA ring hides.
        constexpr int PS = MODE == kNT ? 12 : 4;
        static const bool ps_on = [] { const char* e = getenv("DCV_SPLIT_AT_STORE"); return !(e && e[0] == '0'); }();
        if (ps_on && K % 32 == 0) {
            if (pick == kPickHalfM || (pick == kPickQuarter && Epi::kHead))
                return launch_gemm_cfg<MODE, TileCfg<2, 2, 1, 2, 32, 2, true, PS>, 1, Epi>(A, B, 0, M, N, K, k_chunk, epi, s, tiles_m_out, tw);
            if (pick == kPickNarrowM) return launch_gemm_cfg<MODE, TileCfg<1, 4, 1, 1, 32, 2, true, PS>, 1, Epi>(A, B, 0, M, N, K, k_chunk, epi, s, tiles_m_out, tw);
            if constexpr (!Epi::kHead) {
                if (pick == kPickQuarter) return launch_gemm_cfg<MODE, TileCfg<2, 2, 1, 1, 32, 2, true, PS>, 1, Epi>(A, B, 0, M, N, K, k_chunk, epi, s, tiles_m_out, tw);
            }
        }
    }
#endif
#ifdef DCV_PS_EXPERIMENT   // tools/planes_bench only (-DDCV_PS_EXPERIMENT, DCV_LS=1): bit-identical and measured SLOWER (25.7 vs 22.8 us at 8192 x 256 x 512)
    if constexpr (S && MODE == kNT && !Epi::kHead) {
        static const bool ls_on = [] { const char* e = getenv("DCV_LS"); return e && e[0] == '1'; }();
        if (ls_on && pick == kPickQuarter) {
            const int rc = launch_gemm_ls<TileCfg<2, 2, 1, 1, 32, 2, true, 28>, Epi>(A, B, M, N, K, epi, s, tiles_m_out, tw);
            if (rc != 1) return rc;
        }
    }
#endif
    switch (pick) {
        case kPickNarrowN: return launch_gemm_cfg<MODE, CfgNarrowNT<S>, 1, Epi>(A, B, 0, M, N, K, k_chunk, epi, s, tiles_m_out, tw);
        case kPickNarrowM: return launch_gemm_cfg<MODE, CfgNarrowMT<S>, 1, Epi>(A, B, 0, M, N, K, k_chunk, epi, s, tiles_m_out, tw);
        case kPickHalfM: return launch_gemm_cfg<MODE, CfgHalfMT<S>, 1, Epi>(A, B, 0, M, N, K, k_chunk, epi, s, tiles_m_out, tw);
        case kPickQuarter:
            if constexpr (!Epi::kHead) return launch_gemm_cfg<MODE, CfgQuarterT<S>, 1, Epi>(A, B, 0, M, N, K, k_chunk, epi, s, tiles_m_out, tw);
            else return launch_gemm_cfg<MODE, CfgHalfMT<S>, 1, Epi>(A, B, 0, M, N, K, k_chunk, epi, s, tiles_m_out, tw);
        default: return launch_gemm_cfg<MODE, Big, 1, Epi>(A, B, 0, M, N, K, k_chunk, epi, s, tiles_m_out, tw);
    }
}

template <int MODE, class Epi>
static int launch_gemm(const Operand& A, const Operand& B, int64_t M, int64_t N, int64_t K, int64_t k_chunk,
                       const Epi& epi, hipStream_t s, int* tiles_m_out = nullptr, const TailWs* tw = nullptr) {
#ifdef DCV_BIGCFG
    return launch_gemm_mode<MODE, false, Epi>(A, B, M, N, K, k_chunk, epi, s, tiles_m_out, tw);
#else
    if (gemm_split()) return launch_gemm_mode<MODE, true, Epi>(A, B, M, N, K, k_chunk, epi, s, tiles_m_out, tw);
    return launch_gemm_mode<MODE, false, Epi>(A, B, M, N, K, k_chunk, epi, s, tiles_m_out, tw);
#endif
}

// The engine's products as plain functions, one translation unit per epilogue (inst_*.hip) so that the build compiles
// them in parallel; each is launch_gemm<MODE, Epi> of the named epilogue.
int gemm_nt_bias_act(const Operand& A, const Operand& B, int64_t M, int64_t N, int64_t K, const EpiBiasAct& epi, hipStream_t s, const TailWs* tw);
int gemm_nt_head4(const Operand& A, const Operand& B, int64_t M, int64_t N, int64_t K, const EpiBiasActHead<4>& epi, hipStream_t s, const TailWs* tw);
int gemm_nt_head8(const Operand& A, const Operand& B, int64_t M, int64_t N, int64_t K, const EpiBiasActHead<8>& epi, hipStream_t s, const TailWs* tw);
int gemm_tn_slab(const Operand& A, const Operand& B, int64_t M, int64_t N, int64_t K, int64_t k_chunk, const EpiSlab& epi, hipStream_t s);
int gemm_nn_act_grad(const Operand& A, const Operand& B, int64_t M, int64_t N, int64_t K, const EpiActGrad& epi, hipStream_t s,
                     int* tiles_m_out, const TailWs* tw);
int gemm_nn_store(const Operand& A, const Operand& B, int64_t M, int64_t N, int64_t K, const EpiStore& epi, hipStream_t s);

// Weight gradient (TN, slabs) and input gradient (NN) of one layer as ONE launch (wgrad_dgrad_kernel; defined in
// pair.hip).  Returns 1 when the pair form does not apply (the caller then launches the two products one after the
// other), DCV_OK when both were enqueued, < 0 on error.  tiles_m_out2: row tiles of the NN product (bias partials).
int launch_wgrad_dgrad(const Operand& A1, const Operand& B1, int64_t M1, int64_t N1, int64_t K1, int64_t k_chunk1, const EpiSlab& e1,
                       const Operand& A2, const Operand& B2, int64_t M2, int64_t N2, int64_t K2, const EpiActGrad& e2, int* tiles_m_out2,
                       const TailWs* tw, hipStream_t s);

// ------------------------------------------------------------------ plane operands (TileCfg::PL)
// Plane form of an fp32 matrix [rows][cols]: row r is [plane 1 | plane 2 | plane 3], each Kp = round_up(cols, 32) bf16
// (zero padded), the pieces of split3 -- x = p1 + p2 + p3 exactly, truncation split.  In float units the row pitch is
// 3 * Kp / 2 and the plane stride Kp / 2.  The engine keeps the training matrix and the weights in this form next to
// the fp32 originals: the split is then paid once per matrix instead of once per use inside the product's main loop.
__host__ __device__ inline int64_t planes_kp(int64_t cols) { return (cols + 31) / 32 * 32; }
__host__ __device__ inline int64_t planes_ld(int64_t cols) { return 3 * planes_kp(cols) / 2; }        // floats
__host__ __device__ inline int64_t planes_pstride(int64_t cols) { return planes_kp(cols) / 2; }       // floats
inline size_t planes_bytes(int64_t rows, int64_t cols) { return (size_t)rows * (size_t)planes_ld(cols) * sizeof(float); }
// TRANS: dst row = src column (the planes of the transpose; small matrices only -- the loads are strided)
template <bool TRANS>
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ src, int64_t ld, int64_t rows, int64_t cols,
                                                           float* __restrict__ dst, int64_t rows_out) {
    const int64_t orows = TRANS ? cols : rows, ocols = TRANS ? rows : cols;   // extents of the matrix being written
    const int64_t kp8 = planes_kp(ocols) / 8, ldp = planes_ld(ocols), ps = planes_pstride(ocols);
    const int64_t total = rows_out * kp8;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / kp8, g = i - r * kp8;
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int64_t k = 8 * g + e;
            x[e] = (r < orows && k < ocols) ? (TRANS ? src[k * ld + r] : src[r * ld + k]) : 0.f;
        }
        u32x4 p1, p2, p3;
        split3(x, p1, p2, p3);
        float* o = dst + r * ldp + g * 4;
        *reinterpret_cast<u32x4*>(o) = p1;
        *reinterpret_cast<u32x4*>(o + ps) = p2;
        *reinterpret_cast<u32x4*>(o + 2 * ps) = p3;
    }
}
// rows_out >= (TRANS ? cols : rows): rows past the matrix are written as zeros
template <bool TRANS>
static int launch_split_planes(const float* src, int64_t ld, int64_t rows, int64_t cols, float* dst, int64_t rows_out, hipStream_t s) {
    const int64_t ocols = TRANS ? rows : cols;
    const int64_t total = rows_out * (planes_kp(ocols) / 8);
    if (total <= 0) return DCV_OK;
    int64_t blocks = cdiv(total, 256);
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(split_planes_kernel<TRANS>, dim3((unsigned)blocks), dim3(256), 0, s, src, ld, rows, cols, dst, rows_out);
    DCV_CHECK_LAUNCH();
    return DCV_OK;
}
inline Operand make_plane_operand(const float* planes, int64_t cols, const RowMap& rows = RowMap{nullptr, 0, 0, 0}) {
    Operand o;
    o.p = planes;
    o.ld = planes_ld(cols);
    o.rows = rows;
    o.shift = nullptr;
    o.vec_ok = (reinterpret_cast<uintptr_t>(planes) & 15) == 0;
    o.planes = 1;
    o.pstride = planes_pstride(cols);
    return o;
}

// Tile shapes of the plane kernels (LDS per workgroup in brackets):
//   PL = 3 (both operands pre-split, no vector-ALU work in the main loop):
//          128 x 128, 16-deep stages, ring of 3 [72 KiB] ; 64 x 128, 32-deep stages [72 KiB]
//   PL = 2 (B pre-split, A split in registers): 128 x 128 [80 KiB] ; 64 x 128 [64 KiB]
#ifdef DCV_PL3_BIG   // diagnostic override (tools/planes_bench): KB, NBUF of the PL = 3 128 x 128 tile
template <int PL> using CfgPlBig = TileCfg<2, 2, 2, 2, PL == 3 ? DCV_PL3_BIG : 32, 2, true, PL>;
#else
template <int PL> using CfgPlBig = TileCfg<2, 2, 2, 2, PL == 3 ? 16 : 32, PL == 3 ? 3 : 2, true, PL>;
#endif
#ifndef DCV_PLH_NBUF
#define DCV_PLH_NBUF 2
#endif
template <int PL> using CfgPlHalf = TileCfg<2, 2, 1, 2, 32, DCV_PLH_NBUF, true, PL>;
#ifndef DCV_PLQ_NBUF
#define DCV_PLQ_NBUF 2
#endif
template <int PL> using CfgPlQuarter = TileCfg<2, 2, 1, 1, 32, DCV_PLQ_NBUF, true, PL>;   // 64 x 64: two workgroups per CU on a small grid

// NT product with plane operands: C[M,N] = A[M,K] . B[N,K]^T.  Returns DCV_EINVAL-free "not applicable" (1) when the
// shape or the operands do not qualify -- the caller then takes the fp32-operand kernel -- and a DCV_E* (< 0) on error.
template <int PL, class Epi>
static int launch_gemm_planes(const Operand& A, const Operand& B, int64_t M, int64_t N, int64_t K, const Epi& epi, hipStream_t s,
                              int* tiles_m_out = nullptr) {
    static_assert(PL >= 1 && PL <= 3, "PL");
    const bool a_ok = (PL & 1) ? (A.planes == 1 && A.pstride % 4 == 0) : (A.planes == 0);
    const bool b_ok = (PL & 2) ? (B.planes == 1 && B.pstride % 4 == 0) : (B.planes == 0);
    if (!a_ok || !b_ok || !A.vec_ok || !B.vec_ok || A.ld % 4 != 0 || B.ld % 4 != 0) return 1;
    if (K % 32 != 0 || K < 64 || N <= 32 || M <= 32) return 1;
    if (A.rows.idx != nullptr || B.rows.idx != nullptr || A.shift || B.shift) return 1;
    if (A.ld >= kMaxAffineLd || B.ld >= kMaxAffineLd) return 1;
    auto go = [&](auto cfg) -> int {
        using Cfg = decltype(cfg);
        GemmDims d;
        d.M = M; d.N = N; d.K = K; d.k_chunk = K;
        d.tiles_m = (int)cdiv(M, Cfg::TM);
        d.tiles_n = (int)cdiv(N, Cfg::TN);
        if (tiles_m_out) *tiles_m_out = d.tiles_m;
        static int env = -1;
        if (env < 0) { const char* e = getenv("DCV_XCD_REMAP"); env = e ? atoi(e) : 1; }
        d.xcd_remap = env && d.tiles_m % 8 == 0 && d.tiles_n > 1;
        const int64_t tiles = (int64_t)d.tiles_m * d.tiles_n;
        DCV_REQUIRE(tiles > 0 && tiles < (1ll << 31), "gemm: grid out of range (tiles=%lld)", (long long)tiles);
        return launch_gemm_vec<kNT, Cfg, 1, true, false, Epi>(A, B, 0, d, 1, epi, s);
    };
    const int64_t want = 2 * (int64_t)num_cus();
#ifdef DCV_PL_FORCE_BIG
    return go(CfgPlBig<PL>{});
#endif
    if (cdiv(M, 128) * cdiv(N, 128) < want) {
        // between one and two 64 x 128 workgroups per CU: 64 x 64 tiles, as the fp32-operand kernels choose (pick_cfg)
#ifndef DCV_NO_PLQ
        if (cdiv(M, 64) * cdiv(N, 128) < want && cdiv(M, 64) * cdiv(N, 128) >= want / 2 && N % 64 == 0) return go(CfgPlQuarter<PL>{});
#endif
        return go(CfgPlHalf<PL>{});
    }
    return go(CfgPlBig<PL>{});
}

inline Operand make_operand(const float* p, int64_t ld, int64_t inner_extent, const RowMap& rows = RowMap{nullptr, 0, 0, 0},
                            const float* shift = nullptr) {
    Operand o;
    o.p = p;
    o.ld = ld;
    o.rows = rows;
    o.shift = shift;
    (void)inner_extent;
    o.vec_ok = (ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(p) & 15) == 0) &&
               (shift == nullptr || (reinterpret_cast<uintptr_t>(shift) & 15) == 0);
    return o;
}

}  // namespace dcv
