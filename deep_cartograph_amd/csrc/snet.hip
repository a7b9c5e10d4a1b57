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
#include "snet.h"
#include <new>

namespace dcv {

struct SnetArgs {
    SnetLayer l[DCV_MAX_LAYERS];
    int L;
    const int2* stage_tab;        // staging table: per 16-byte unit of the LDS image {source element offset into params or -1,
                                  // LDS float offset | valid elements << 20 | 16-byte load legal << 24}
    int stage_n;
    int lh[DCV_MAX_LAYERS + 1];   // LDS float offset of H_l [TR][ps_l]   (H_0 = the input tile)
    int ps[DCV_MAX_LAYERS + 1];   // row stride of H_l = padded width + 4
    int lred;                     // LDS float offset of the reduction scratch (kSnetThreads doubles)
    const float* params;
    const float* img;             // global weight image in the LDS layout (snet_image_build), or null: table-driven staging
    int img_floats;
    const float* Xn;
    int64_t ld;
    RowMap rows;
    int64_t R;                    // rows of one batch
    int nb, wgpb;                 // batches of this launch (> 1: evaluation only, snet_ae_eval_batches) and workgroups per batch
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
    unsigned long long* stamps;   // diagnostic (dcv_debug_snet_stamps): s_memrealtime of workgroup 0 at the phase boundaries, or null
};
#define SNET_STAMP(k)                                                                          \
    do {                                                                                        \
        if (a.stamps != nullptr && blockIdx.x == 0 && threadIdx.x == 0) a.stamps[k] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)

template <int TR>
__global__ __launch_bounds__(kSnetThreads) void snet_ae_kernel(SnetArgs a) {
    constexpr int NT = kSnetThreads;
    constexpr int RG = TR / 16, CG = kSnetWaves / RG;
    extern __shared__ __attribute__((aligned(16))) float sl[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    // wave -> (row group, column group): consecutive waves (= the four SIMDs) take different column groups first, so a narrow
    // layer with one or two column tiles keeps one wave on each SIMD instead of two waves on half of them
    const int rg = wave % RG, cg = wave / RG;
    const int q = lane >> 4, n = lane & 15;
    const int L = a.L;
    // the kernel arguments are 2 KB (one table entry per layer and phase): touch every 64-byte line with a scalar load now, so
    // that the first use of a layer's entry in the forward chain is a scalar-cache hit (measured with the new wave map:
    // 21.2 -> 19.8 us per evaluation step, 48.2 -> 46.7 per training step on the same box)
    unsigned ka_touch = 0;
    {
        const __attribute__((address_space(4))) unsigned* kp = (const __attribute__((address_space(4))) unsigned*)__builtin_amdgcn_kernarg_segment_ptr();
#pragma unroll
        for (int off = 0; off < (int)sizeof(SnetArgs); off += 64) ka_touch ^= kp[off / 4];
    }
    SNET_STAMP(0);
    // ---- input tile H_0 (rows past the batch: zeros): with 16-byte loads the whole tile is at most four units per thread
    //      (TR * pin / 4 <= 4 * NT); they are issued inside the weight staging, behind its data loads, and written to LDS
    //      after it
    // workgroup -> (batch of the launch, tile of the batch): one batch unless this is a batched evaluation (nb > 1), whose
    // batch j covers the logical rows [j * R, (j + 1) * R) of the row map
    const int bj = a.nb > 1 ? (int)blockIdx.x / a.wgpb : 0;
    const int tile0 = (int)blockIdx.x - bj * a.wgpb;
    const int64_t r0 = (int64_t)tile0 * TR;
    RowMap rows = a.rows;
    if (bj != 0) {
        if (rows.idx != nullptr) rows.idx += (int64_t)bj * a.R;
        rows.row0 += (int64_t)bj * a.R;
    }
    const int F0 = a.l[0].in, p0 = a.l[0].pin, ps0 = a.ps[0];
    float* H0 = sl + a.lh[0];
    const bool x_vec = (F0 & 3) == 0 && (a.ld & 3) == 0 && (reinterpret_cast<uintptr_t>(a.Xn) & 15) == 0;
    const int x_sh = a.l[0].c4_shift, x_tot = TR << x_sh;
    float4 xv[4];
    bool x_issued = false;
    // ---- stage every weight image and bias (zero-padded, row stride pin + 4) through the plan's staging table: one flat
    //      space of 16-byte units over all layers, twelve independent loads in flight per thread and pass -- two dependent
    //      round trips in all (table entry, then data).  Per-layer loops cost one L2 round trip per pass (8-11 us).
    if (a.img != nullptr) {
        // one contiguous LDS-DMA copy of the whole weight image (kept current by the optimiser: OptArgs::img); the input rows
        // are requested right behind it
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
            if (x_vec && !x_issued) {   // behind the first pass's data loads (loads return in order: issued earlier, the cold
                x_issued = true;        // rows of X would hold up the table entries; here they ride along the data round trip)
                const int f4 = F0 >> 2;
    #pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = t + NT * u, r = i >> x_sh, c = i - (r << x_sh);
                    xv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (i < x_tot && r0 + r < a.R && c < f4) xv[u] = *reinterpret_cast<const float4*>(a.Xn + rows.template get<true>(r0 + r) * a.ld + 4 * c);
                }
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
    if (x_vec && !x_issued) {   // (a plan without staging units: not reachable, kept for the invariant xv is loaded)
        const int f4 = F0 >> 2;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = t + NT * u, r = i >> x_sh, c = i - (r << x_sh);
            xv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < x_tot && r0 + r < a.R && c < f4) xv[u] = *reinterpret_cast<const float4*>(a.Xn + rows.template get<true>(r0 + r) * a.ld + 4 * c);
        }
    }
    asm volatile("" ::"s"(ka_touch));   // the touches have landed
    if (a.img != nullptr) vm_wait<0>();   // the image copies of this wave have landed (the barrier below covers the other waves)
    SNET_STAMP(1);
    if (x_vec) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = t + NT * u, r = i >> x_sh, c = i - (r << x_sh);
            if (i < x_tot) *reinterpret_cast<float4*>(H0 + r * ps0 + 4 * c) = xv[u];
        }
    } else {
        for (int i = t; i < TR * p0; i += NT) {
            const int r = i / p0, c = i - r * p0;
            float v = 0.f;
            if (r0 + r < a.R && c < F0) v = a.Xn[rows.template get<true>(r0 + r) * a.ld + c];
            H0[r * ps0 + c] = v;
        }
    }
    __syncthreads();
    SNET_STAMP(2);
    // ---- forward chain
    double sse = 0.0;
    for (int l = 0; l < L; ++l) {
        const SnetLayer& y = a.l[l];
        const float* Hin = sl + a.lh[l];
        float* Hout = sl + a.lh[l + 1];
        const int psin = a.ps[l], pso = a.ps[l + 1];
        const bool last = l == L - 1;
        const float* ap = Hin + (rg * 16 + n) * psin + 4 * q;
        const float* W = sl + y.lw + n * y.pws + 4 * q;
        const float* H0row = sl + a.lh[0] + (rg * 16 + 4 * q) * a.ps[0];
#define SNET_FWD(NK)                                                                                                        \
        SnetFrags<NK> A;                                                                                                     \
        A.load(ap);                                                                                                          \
        for (int ct = cg; ct < y.nk_out; ct += CG) {                                                                         \
            const sv4f acc = snet_fwd_tile<NK>(A, W + ct * 16 * y.pws);                                                      \
            const int col = ct * 16 + n;                                                                                     \
            const float bias = sl[y.lb + col];                                                                               \
            sv4f h = snet_act4(y.act, acc + bias);                                                                           \
            if (col >= y.out) h = sv4f{0.f, 0.f, 0.f, 0.f};                                                                  \
            if (last) {   /* autoencoder loss on the spot: e = (y - xn) * range ; dY = scale * (y - xn) * range^2 * act'(y) */ \
                const float rgv = col < y.out ? a.range[col] : 0.f;                                                          \
                const sv4f dh = snet_actgrad4(y.act, h);                                                                     \
                _Pragma("unroll") for (int v = 0; v < 4; ++v) {                                                              \
                    float g = 0.f;                                                                                           \
                    if (col < y.out && r0 + rg * 16 + 4 * q + v < a.R) {                                                     \
                        const float x = H0row[v * a.ps[0] + col];                                                            \
                        const float ev = (h[v] - x) * rgv;                                                                   \
                        sse += (double)ev * (double)ev;                                                                      \
                        g = a.scale * (h[v] - x) * rgv * rgv * dh[v];                                                        \
                    }                                                                                                        \
                    h[v] = g;   /* H_L now holds dZ_L */                                                                     \
                }                                                                                                            \
            }                                                                                                                \
            _Pragma("unroll") for (int v = 0; v < 4; ++v) Hout[(rg * 16 + 4 * q + v) * pso + col] = h[v];                    \
        }
        SNET_NK_SWITCH(y.nk_in, SNET_FWD)
#undef SNET_FWD
        __syncthreads();
        SNET_STAMP(3 + l);
    }
    // ---- squared error of the tile -> partial -> (ticket) the step's loss record
    {
        // waves by shuffles, the eight wave sums in wave order by one thread (a fixed order: deterministic)
        double* red = reinterpret_cast<double*>(sl + a.lred);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sse += __shfl_down(sse, off, 64);
        if (lane == 0) red[wave] = sse;
        __syncthreads();
        if (t == 0) {
            double tot = red[0];
#pragma unroll
            for (int w = 1; w < kSnetWaves; ++w) tot += red[w];
            handoff_store(a.sse_part + blockIdx.x, tot);
        }
    }
    SNET_STAMP(20);
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
                const float* ap = dZ + (rg * 16 + n) * psz + 4 * q;
                const float* W = sl + y.lw + (4 * q) * y.pws + n;
#define SNET_DGRAD(NK)                                                                                   \
                SnetFrags<NK> A;                                                                         \
                A.load(ap);                                                                              \
                _Pragma("unroll") for (int j = 0; j < kSnetMaxTiles; ++j) {                              \
                    const int it = cg + j * CG;                                                          \
                    if (it < y.nk_in) dg[j] = snet_dgrad_tile<NK>(A, W + it * 16, y.pws);                \
                }
                SNET_NK_SWITCH(y.nk_out, SNET_DGRAD)
#undef SNET_DGRAD
            }
            SNET_STAMP(40 + l);
            // weight gradient of the tile: the nk_out x nk_in tiles round-robin over the waves
            {
                const int nti = y.nk_in, ntot = y.nk_out * nti;
                float* pw = a.part + y.pw_off + (int64_t)blockIdx.x * y.pw_stride;
                const bool vec_ok = (y.in & 3) == 0 && ((y.pw_off + (int64_t)blockIdx.x * y.pw_stride) & 3) == 0;   // a.part is a hipMalloc base
                int ot = 0, it = wave;
                while (it >= nti) { it -= nti; ++ot; }
#pragma unroll 2
                for (int tile = wave; tile < ntot; tile += kSnetWaves) {
                    // operands swapped (rows of the MFMA tile = input columns): a lane ends up with four CONSECUTIVE inputs
                    // i of one output o = its 16 bytes of the partial's row -- one global_store_dwordx4 per tile and lane
                    // instead of four 4-byte stores (the partial stores of the two wide layers were what these phases waited on)
                    const sv4f acc = snet_wgrad_tile<TR>(Hin + q * psh + it * 16 + n, psh, dZ + q * psz + ot * 16 + n, psz);
                    const int o = ot * 16 + n, i0 = it * 16 + 4 * q;
                    if (o < y.out) {
                        float* dst = pw + (int64_t)o * y.in + i0;
                        if (vec_ok && i0 + 4 <= y.in) {
                            handoff_store16(dst, acc);   // write-through: the 10 MB of partials are not left for the write-back at the launch's end (44.3 -> 42.8 us per step)
                        } else {
#pragma unroll
                            for (int v = 0; v < 4; ++v)
                                if (i0 + v < y.in) dst[v] = acc[v];
                        }
                    }
                    it += kSnetWaves;
                    while (it >= nti) { it -= nti; ++ot; }
                }
                SNET_STAMP(48 + l);
                // bias gradient: column sums of dZ_l over the tile's rows, rows in index order
                // (four threads per column, a quarter of the rows each, combined by two shuffles: a fixed order)
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
            SNET_STAMP(21 + 2 * l);
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
    SNET_STAMP(60);
    // ---- last workgroup: total squared error in block order, loss record
    unsigned* flag = reinterpret_cast<unsigned*>(sl + a.lred);
    __syncthreads();
    const unsigned wgpb = a.nb > 1 ? (unsigned)a.wgpb : gridDim.x;
    if (!handoff_arrive_last(a.ticket + bj, wgpb, flag)) return;
    SNET_STAMP(61);
    if (t < 64) {
        double tot = 0.0;
        const double* sp = a.sse_part + (int64_t)bj * wgpb;
        for (int b0 = t; b0 < (int)wgpb; b0 += 64) tot += handoff_load(sp + b0);
        for (int off = 32; off > 0; off >>= 1) tot += __shfl_down(tot, off, 64);
        if (t == 0) {
            if (a.nb <= 1) a.stats[0] = tot;
            if (a.log != nullptr) {   // (a data-parallel step logs after the all-reduce of the sum: ae_log_kernel)
                // a batched evaluation appends its records in batch order: every batch's last arriver reads the counter, the last
                // of those (a second ticket, taken behind the read) moves it
                const int slot0 = *a.log_count;
                const int slot = slot0 + bj;
                if (slot < a.log_cap) {
                    a.log[(int64_t)slot * a.log_width + 0] = tot / (a.Bg * (double)a.l[0].in);
                    a.log[(int64_t)slot * a.log_width + 1] = a.Bg;
                }
                if (a.nb <= 1) {
                    *a.log_count = slot + 1;
                } else {
                    asm volatile("s_waitcnt vmcnt(0) ; the counter has been read" ::: "memory");
                    const unsigned prev = __hip_atomic_fetch_add(a.ticket + a.nb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (prev == (unsigned)a.nb - 1u) {
                        __hip_atomic_store(a.ticket + a.nb, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        *a.log_count = slot0 + a.nb;
                    }
                }
            }
        }
    }
}

struct SnetPlan {
    int tr_max;          // largest tile (32 or 16 rows) whose activation map fits in LDS behind the weight images; 16 always fits when 32 does
    int fl;              // LDS floats of the weight images
    SnetArgs base;       // layer table (the activation map is laid out per launch: it depends on the tile rows)
    float* part;         // gradient partials
    int64_t part_floats; // capacity
    int64_t per_wg;      // floats of one workgroup's partials over all layers (dense: sum out * in + out)
    unsigned long long* stamps;   // 64 words, or null (DCV_SNET_STAMPS=1)
    int2* stage_tab;     // device copy of the staging table
    double* ev_sse;      // batched evaluation: squared-error partials [batches][workgroups per batch] ...
    int64_t ev_sse_n;
    unsigned* ev_ticket; // ... and one ticket per batch + the one that moves the log counter (zero between launches)
    int64_t ev_ticket_n;
};

// Activation map of a TR-row tile behind the weight images (H_0 .. H_L, the reduction scratch); returns the LDS bytes, 0 when
// the tile does not fit
static size_t snet_ae_map(SnetArgs& a, int L, int fl, int TR) {
    int f = fl;
    for (int l = 0; l <= L; ++l) {
        const int P = l == 0 ? a.l[0].pin : a.l[l - 1].pout;
        if (l > 0 && l < L && a.l[l].pin != P) return 0;   // (always equal: both pad the same width)
        a.ps[l] = P + 4;
        a.lh[l] = f;
        f += TR * (P + 4);
    }
    f = (f + 3) / 4 * 4;
    a.lred = f;
    f += 2 * kSnetThreads;   // kSnetThreads doubles
    return (size_t)f * sizeof(float) <= (size_t)160 * 1024 ? (size_t)f * sizeof(float) : 0;
}
// Rows per workgroup for batches of R rows.  16-row tiles shorten the latency chain of a workgroup; measured on one box
// (tools/dbg/ae_tr_probe.py, contiguous batches, us per training step, 16 | 32 rows): 54-16-8-2 autoencoder 27.0 | 28.5 at batch
// 128, 27.0 | 29.4 at 512, 28.2 | 29.8 at 2048, 30.8 | 30.3 at 4096; the C2 network 128-64-32-2: 31.1 | 35.7 at 128, 31.1 | 36.9
// at 512, 33.1 | 37.3 at 2048 (21 MB of partials and still ahead), 37.4 | 37.9 at 4096.  So: 16 rows up to 2048 rows (128
// workgroups), the larger tile beyond.  The choice depends on the rows of ONE batch only: a batched validation pass tiles its
// batches as the single steps would, so its records stay bit-equal to theirs.
static int snet_ae_pick_tr(const SnetPlan* pl, int64_t R) {
    static const int tr_env = [] { const char* e = getenv("DCV_SNET_TR"); return e ? atoi(e) : 0; }();   // 16 | 32: force the rows per workgroup
    if ((tr_env == 16 || tr_env == 32) && tr_env <= pl->tr_max) return tr_env;
    if (pl->tr_max == 32) {
        if (cdiv(R, 16) <= 128) return 16;
    }
    return pl->tr_max;
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
    std::vector<int2> tab;
    if (!snet_layout(m, a.l, tab, nullptr, fl, per_wg)) { delete pl; return false; }
    pl->per_wg = per_wg;
    (void)snet_image_build(m);   // on failure the kernels keep the table-driven staging
    pl->stage_tab = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&pl->stage_tab), tab.size() * sizeof(int2)) != hipSuccess ||
        hipMemcpy(pl->stage_tab, tab.data(), tab.size() * sizeof(int2), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipGetLastError();
        if (pl->stage_tab) (void)hipFree(pl->stage_tab);
        delete pl;
        return false;
    }
    a.stage_tab = pl->stage_tab;
    a.stage_n = (int)tab.size();
    pl->fl = fl;
    pl->tr_max = 0;
    for (int TR : {32, 16}) {
        SnetArgs tmp = a;
        if (snet_ae_map(tmp, m->L, fl, TR) != 0) {
            pl->tr_max = TR;
            break;
        }
    }
    if (pl->tr_max != 0) {
        pl->part = nullptr;
        pl->part_floats = 0;
        pl->stamps = nullptr;
        pl->ev_sse = nullptr;
        pl->ev_sse_n = 0;
        pl->ev_ticket = nullptr;
        pl->ev_ticket_n = 0;
        // batched validation passes: partials and tickets for the bounds of dcv_mlp_eval_steps (33 KB); without them the
        // passes go batch by batch
        if (hipMalloc(reinterpret_cast<void**>(&pl->ev_sse), (size_t)kEvalWorkgroupsPerLaunch * sizeof(double)) == hipSuccess &&
            hipMalloc(reinterpret_cast<void**>(&pl->ev_ticket), (size_t)(kEvalBatchesPerLaunch + 1) * sizeof(unsigned)) == hipSuccess &&
            hipMemset(pl->ev_ticket, 0, (size_t)(kEvalBatchesPerLaunch + 1) * sizeof(unsigned)) == hipSuccess) {
            pl->ev_sse_n = kEvalWorkgroupsPerLaunch;
            pl->ev_ticket_n = kEvalBatchesPerLaunch + 1;
        } else {
            (void)hipGetLastError();
        }
        {
            const char* e = getenv("DCV_SNET_STAMPS");
            if (e && e[0] == '1' && hipMalloc(reinterpret_cast<void**>(&pl->stamps), 64 * sizeof(unsigned long long)) == hipSuccess)
                (void)hipMemset(pl->stamps, 0, 64 * sizeof(unsigned long long));
        }
        m->snet = pl;
        return true;
    }
    (void)hipFree(pl->stage_tab);
    delete pl;
    return false;
}

// ---- the global weight image: every weight / bias at its LDS-image offset, zero padding included, kept current by the
// optimiser kernels (OptArgs::img / img_idx) and rebuilt by dcv_mlp_set_params
__global__ __launch_bounds__(256) void snet_pack_kernel(const float* __restrict__ params, const int* __restrict__ idx, int64_t n, float* __restrict__ img) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int j = idx[i];
        if (j >= 0) img[j] = params[i];
    }
}
bool snet_image_build(dcv_mlp* m) {
    static const bool off = [] { const char* e = getenv("DCV_SNET_IMG"); return e && e[0] == '0'; }();
    if (off) return false;
    if (m->snet_img != nullptr) return true;
    SnetLayer ly[DCV_MAX_LAYERS];
    std::vector<int2> tab;
    std::vector<int> idx;
    int fl = 0;
    int64_t per_wg = 0;
    if (!snet_layout(m, ly, tab, nullptr, fl, per_wg, &idx)) return false;
    for (int l = 0; l < m->L; ++l)   // 16-byte copies: every image row must start on a 16-byte boundary (pws and pout are multiples of 4)
        if ((ly[l].lw & 3) || (ly[l].lb & 3)) return false;
    float* img = nullptr;
    int* didx = nullptr;
    const size_t fpad = ((size_t)fl + 255) / 256 * 256;
    if (hipMalloc(reinterpret_cast<void**>(&img), fpad * sizeof(float)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&didx), idx.size() * sizeof(int)) != hipSuccess ||
        hipMemset(img, 0, fpad * sizeof(float)) != hipSuccess ||
        hipMemcpy(didx, idx.data(), idx.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipGetLastError();
        if (img) (void)hipFree(img);
        if (didx) (void)hipFree(didx);
        return false;
    }
    m->snet_img = img;
    m->snet_img_idx = didx;
    m->snet_img_floats = fl;
    if (snet_image_repack(m, nullptr) != DCV_OK || hipDeviceSynchronize() != hipSuccess) {
        (void)hipGetLastError();
        snet_image_free(m);
        return false;
    }
    return true;
}
int snet_image_repack(dcv_mlp* m, hipStream_t s) {
    if (m->snet_img == nullptr) return DCV_OK;
    int64_t blocks = cdiv(m->n_params, 256);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(snet_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)m->params, (const int*)m->snet_img_idx, m->n_params, m->snet_img);
    DCV_CHECK_LAUNCH();
    return DCV_OK;
}
void snet_image_free(dcv_mlp* m) {
    if (m->snet_img) (void)hipFree(m->snet_img);
    if (m->snet_img_idx) (void)hipFree(m->snet_img_idx);
    m->snet_img = nullptr;
    m->snet_img_idx = nullptr;
    m->snet_img_floats = 0;
}

void snet_free(dcv_mlp* m) {
    SnetPlan* pl = static_cast<SnetPlan*>(m->snet);
    if (!pl) return;
    if (pl->part) (void)hipFree(pl->part);
    if (pl->stamps) (void)hipFree(pl->stamps);
    if (pl->stage_tab) (void)hipFree(pl->stage_tab);
    if (pl->ev_sse) (void)hipFree(pl->ev_sse);
    if (pl->ev_ticket) (void)hipFree(pl->ev_ticket);
    delete pl;
    m->snet = nullptr;
}

// Rows per workgroup the fused autoencoder kernel takes for batches of R rows (R = 0: its largest tile); the plan is built on
// first use; 0: the fused form does not apply.
int snet_ae_tile_rows(dcv_mlp* m, int64_t R) {
    if (m->snet == nullptr) {
        if (m->snet_tried || !snet_build(m)) {
            m->snet_tried = true;
            return 0;
        }
        m->snet_tried = true;
    }
    const SnetPlan* pl = static_cast<SnetPlan*>(m->snet);
    return R > 0 ? snet_ae_pick_tr(pl, R) : pl->tr_max;
}

// One fused step of the autoencoder over `R` rows (train != 0: gradient partials are left for the reduction, whose
// descriptors are filled into `ra`).  Returns 1 when the fused form does not apply (the caller takes the layer-by-layer
// path), DCV_OK when the launch was enqueued.
// nb > 1 (evaluation only): nb batches of R rows each in the one launch -- batch j = the logical rows [j * R, (j + 1) * R) of
// `rm` -- with one loss record per batch, appended in batch order.
int snet_ae_step(dcv_mlp* m, const float* Xn_d, int64_t ld, const RowMap& rm, int64_t R, int64_t batch, int train, ReduceArgsView* ra,
                 hipStream_t s, bool write_log, int nb) {
    static const int64_t kMaxPartBytes = 96ll << 20;
    if (snet_ae_tile_rows(m) == 0) return 1;
    SnetPlan* pl = static_cast<SnetPlan*>(m->snet);
    const int TR = snet_ae_pick_tr(pl, R);
    const int64_t wgpb = cdiv(R, TR);
    if (wgpb > m->spart_blocks || wgpb * pl->per_wg * (int64_t)sizeof(float) > kMaxPartBytes || wgpb > 512) return 1;   // large batches: the tiled products are the better engine
    if (nb < 1 || (nb > 1 && (train || !write_log))) return 1;
    const int64_t nwg = wgpb * nb;
    if (nb > 1 && (nwg > pl->ev_sse_n || nb + 1 > pl->ev_ticket_n)) return 1;   // (sized by snet_build for the bounds of dcv_mlp_eval_steps)
    const int64_t part_need = nwg * pl->per_wg + 8 * (int64_t)m->L;   // + the alignment padding of the items
    if (train && pl->part_floats < part_need) {
        if (pl->part) (void)hipFree(pl->part);
        pl->part = nullptr;
        pl->part_floats = 0;
        if (hipMalloc(reinterpret_cast<void**>(&pl->part), (size_t)part_need * sizeof(float)) != hipSuccess) {
            (void)hipGetLastError();
            return 1;
        }
        pl->part_floats = part_need;
    }
    SnetArgs a = pl->base;
    const size_t lds_bytes = snet_ae_map(a, m->L, pl->fl, TR);
    if (lds_bytes == 0) return 1;
    int64_t off = 0;
    for (int l = 0; l < m->L; ++l) {
        SnetLayer& y = a.l[l];
        y.pw_off = off; off += nwg * (int64_t)y.pw_stride;   // 16-byte aligned items (vector stores here, vector loads in the reduction)
        y.pb_off = off; off += nwg * (int64_t)y.pb_stride;
        if (ra) {
            ra->slab[l] = pl->part + y.pw_off;
            ra->bpart[l] = pl->part + y.pb_off;
            ra->splits[l] = (int)nwg;
            ra->bblocks[l] = (int)nwg;
            ra->wstride[l] = y.pw_stride;
            ra->bstride[l] = y.pb_stride;
        }
    }
    a.params = m->params;
    a.img = m->snet_img;
    a.img_floats = m->snet_img_floats;
    a.Xn = Xn_d;
    a.ld = ld;
    a.rows = rm;
    a.R = R;
    a.nb = nb;
    a.wgpb = (int)wgpb;
    a.range = m->feat_range;
    a.scale = (float)(2.0 / ((double)batch * (double)m->desc.dims[0]));
    a.train = train;
    a.part = pl->part;
    a.sse_part = nb > 1 ? pl->ev_sse : m->spart;
    a.ticket = nb > 1 ? pl->ev_ticket : m->ticket;
    a.stats = m->stats;
    a.Bg = (double)batch;
    a.log = write_log ? m->log : nullptr;
    a.log_count = m->log_count;
    a.log_cap = m->log_cap;
    a.log_width = m->log_width;
    a.stamps = pl->stamps;
    auto launch = [&](auto kern) -> int {
        static int attr_state[2] = {0, 0};   // 0 unknown, 1 set, -1 refused by the runtime (the fused form is then off)
        const int slot = TR == 32 ? 0 : 1;
        if (attr_state[slot] == 0) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
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
    };
    return TR == 32 ? launch(snet_ae_kernel<32>) : launch(snet_ae_kernel<16>);
}

}  // namespace dcv

// diagnostic (tools/dbg/snet_probe.py; not part of include/dcv.h): the 64 phase stamps of the last fused launch
// (s_memrealtime ticks of 10 ns), 0 where not taken; needs DCV_SNET_STAMPS=1 at engine creation
extern "C" int dcv_debug_snet_stamps(dcv_mlp* m, unsigned long long* out_h) {
    using namespace dcv;
    if (!m || !out_h) return DCV_EINVAL;
    SnetPlan* pl = static_cast<SnetPlan*>(m->snet);
    if (!pl || !pl->stamps) return 1;
    DCV_CHECK_HIP(hipDeviceSynchronize());
    DCV_CHECK_HIP(hipMemcpy(out_h, pl->stamps, 64 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return DCV_OK;
}
