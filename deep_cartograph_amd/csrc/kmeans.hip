// k-means on the projected CVs (float64, d <= 16, k <= 64): one HBM-bound pass per Lloyd
// iteration computing labels, per-cluster sums / counts, inertia and the number of changed
// labels; centroid-nearest-sample search; 1-NN label transfer.
//
// Determinism: every wave owns a private LDS accumulator, waves are combined in wave order,
// blocks in block order, and each thread walks its points in increasing index order.
#include "gemm_kernels.h"   // butterfly_sum
#include <stdlib.h>

namespace dcv {

constexpr int kKmThreads = 256;
constexpr int kKmMaxD = 16;
constexpr int kKmMaxK = 64;
constexpr int kKmMaxBlocks = 1024;

static int km_blocks(int64_t n) {
    int64_t b = cdiv(n, (int64_t)kKmThreads * 4);
    const int64_t cap = (int64_t)num_cus() * 4;
    if (b > cap) b = cap;
    if (b > kKmMaxBlocks) b = kKmMaxBlocks;
    if (b < 1) b = 1;
    return (int)b;
}

__device__ __forceinline__ void load_point(const double* __restrict__ P, int64_t i, int d, double* x) {
    const double* p = P + i * d;
    if ((d & 1) == 0) {
#pragma unroll
        for (int c = 0; c < kKmMaxD; c += 2) {
            if (c < d) {
                const double2 v = *reinterpret_cast<const double2*>(p + c);
                x[c] = v.x;
                x[c + 1] = v.y;
            }
        }
    } else {
#pragma unroll
        for (int c = 0; c < kKmMaxD; ++c)
            if (c < d) x[c] = p[c];
    }
}


// ---- D = 4 float64 coordinates = 32 bytes per point.  A lane that reads its own point issues two 16-byte loads at a
// 32-byte lane stride: every wave-instruction touches 2 KB of lines and uses half of them (measured 3.9 - 4.2 TB/s where the
// 16-byte-per-lane contiguous streams of this library reach 5.3).  Instead two wave-instructions each read 1 KB contiguously
// -- lane l takes 16-byte unit l: half (l & 1) of point l >> 1 of the first, resp. second, 32 points of a 64-point chunk --
// the two lanes of a pair swap halves (row DPP, quad_perm [1, 0, 3, 2]: no LDS) and every lane ends up with ONE full point:
// even lanes the point of the first instruction, odd lanes that of the second.
struct PairUnits {
    double2 a, b;
};
__device__ __forceinline__ int64_t pair_point(int64_t chunk_base, int lane) { return chunk_base + (lane >> 1) + ((lane & 1) ? 32 : 0); }
__device__ __forceinline__ void pair_issue(const double* __restrict__ P, int64_t chunk_base, int64_t end, int lane, PairUnits& r) {
    // UNCONDITIONAL loads from clamped indices (points past `end` are masked where they are used): a load inside a branch
    // makes hipcc's wait-count analysis give up -- it put s_waitcnt vmcnt(0) right behind the NEXT batch's loads, so nothing
    // was ever in flight during the compute (round 4: both passes sat at 3.9 TB/s whatever their arithmetic)
    int64_t pa = chunk_base + (lane >> 1), pb = pa + 32;
    pa = pa < end ? pa : end - 1;
    pb = pb < end ? pb : end - 1;
    const int h = 2 * (lane & 1);
    r.a = *reinterpret_cast<const double2*>(P + pa * 4 + h);
    r.b = *reinterpret_cast<const double2*>(P + pb * 4 + h);
}
__device__ __forceinline__ double dpp_swap_pair(double v) {
    const long long u = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_mov_dpp((int)(u & 0xFFFFFFFFll), 0xB1, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp((int)(u >> 32), 0xB1, 0xF, 0xF, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ void pair_finish(const PairUnits& r, int lane, double (&x)[4]) {
    const double sax = dpp_swap_pair(r.a.x), say = dpp_swap_pair(r.a.y);   // the partner's unit of the first instruction
    const double sbx = dpp_swap_pair(r.b.x), sby = dpp_swap_pair(r.b.y);
    const bool odd = (lane & 1) != 0;
    x[0] = odd ? sbx : r.a.x;
    x[1] = odd ? sby : r.a.y;
    x[2] = odd ? r.b.x : sax;
    x[3] = odd ? r.b.y : say;
}

// ---- per-wave LDS-DMA ring for the point stream (D = 2 or 4 coordinates).  What bounded both streaming passes of this file
// at ~4 TB/s (round 4) was the number of bytes a CU keeps in flight: the points of a batch sit in VGPRs from issue to use, the
// k-means pass holds ~100 accumulator registers besides, two waves per SIMD x two points per lane = 37 KB per CU against the
// ~50-70 KB that 6 TB/s x 2 us of loaded latency ask for.  Here every wave owns kRingSlots slots of 64 points in LDS and keeps
// kRingSlots - 1 chunks requested ahead with global_load_lds (no VGPR between memory and LDS: 16 waves x 3 slots x 2 KB = 96 KB
// per CU in flight); a lane then reads ITS point from the slot (the pair swap of the register path is not needed).  Only the
// issuing wave reads its slots: the covering s_waitcnt vmcnt(N) is all the synchronisation there is (MI355X_MICROARCH.md, "Two
// waves per SIMD", item 7).  vmcnt counts loads, LDS-DMA and stores together in issue order; the loop below issues nothing
// but its own DMAs and (k-means) ONE label store per chunk, so the number of younger operations is known exactly.
constexpr int kRingSlots = 4;   // (k-means: 110 VGPRs -> four blocks per CU fit beside 4 x 36 KB of rings: 16 waves x 3 slots x 2.25 KB = 108 KB in flight)
__device__ __forceinline__ void glds4(const void* gsrc, unsigned lds_wave_addr) {   // 4 bytes per lane: LDS[m0 + 4 * lane]
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dword %1, off nt\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_wave_addr)
        : "memory");
}
// 16 bytes per lane, non-temporal: the points are streamed once (MI355X_MICROARCH.md, nt-weights: issue -> landed 18 % sooner
// for once-read streams).  DCV_KM_NT=0 at build time restores the default policy.
#ifndef DCV_KM_NT
#define DCV_KM_NT 1
#endif
__device__ __forceinline__ void glds16_stream(const float* gsrc, unsigned lds_wave_addr) {
#if DCV_KM_NT
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off nt\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_wave_addr)
        : "memory");
#else
    glds16(gsrc, lds_wave_addr);
#endif
}
__device__ __forceinline__ void vm_wait_dyn(int n) {   // s_waitcnt vmcnt(min(n, 31)): fewer allowed in flight only waits longer
    n = __builtin_amdgcn_readfirstlane(n);   // wave-uniform by construction: a scalar branch, not an exec-masked tree
    switch (n) {
#define DCV_VMW(N) case N: vm_wait<N>(); break;
        DCV_VMW(0) DCV_VMW(1) DCV_VMW(2) DCV_VMW(3) DCV_VMW(4) DCV_VMW(5) DCV_VMW(6) DCV_VMW(7) DCV_VMW(8) DCV_VMW(9) DCV_VMW(10)
        DCV_VMW(11) DCV_VMW(12) DCV_VMW(13) DCV_VMW(14) DCV_VMW(15) DCV_VMW(16) DCV_VMW(17) DCV_VMW(18) DCV_VMW(19) DCV_VMW(20)
        DCV_VMW(21) DCV_VMW(22) DCV_VMW(23) DCV_VMW(24) DCV_VMW(25) DCV_VMW(26) DCV_VMW(27) DCV_VMW(28) DCV_VMW(29) DCV_VMW(30)
#undef DCV_VMW
        default: vm_wait<31>(); break;
    }
}
// Streams the points [begin, end) of a block through the calling wave's ring: chunk kk of the wave is the 64 points from
// begin + (wave + 4 kk) * 64; f(x, i, old) is called for every valid point by its lane (old: the point's previous label when
// LABELS, else 0).  stores_per_chunk: vector-memory stores f issues per chunk (they count in vmcnt).
template <int D, bool LABELS, class F>
__device__ __forceinline__ void ring_stream(const double* __restrict__ P, const int32_t* __restrict__ labels, int64_t begin, int64_t end, char* ring,
                                            int wave, int lane, int stores_per_chunk, F&& f) {
    static_assert(D == 2 || D == 4, "ring: 16-byte units");
    constexpr int UNITS = D / 2, OPS = UNITS + (LABELS ? 1 : 0), SLOT = UNITS * 1024 + (LABELS ? 256 : 0);
    char* my = ring + wave * kRingSlots * SLOT;
    const unsigned my_lds = lds_addr_uniform(reinterpret_cast<const float*>(my));
    const int64_t nchunk = (end - begin + 63) / 64;
    const int64_t nw = nchunk > wave ? (nchunk - wave + 3) / 4 : 0;
    const char* Pb = reinterpret_cast<const char*>(P);
    const int64_t max_off = end * (int64_t)(D * 8) - 16;
    auto issue = [&](int64_t kk) {
        const int64_t cb = begin + (wave + 4 * kk) * 64;
        const unsigned slot = my_lds + (unsigned)(kk % kRingSlots) * SLOT;
#pragma unroll
        for (int u = 0; u < UNITS; ++u) {
            int64_t off = cb * (int64_t)(D * 8) + u * 1024 + lane * 16;
            off = off < max_off ? off : max_off;   // units past the block's range re-read its last unit (never used)
            glds16_stream(reinterpret_cast<const float*>(Pb + off), slot + u * 1024);
        }
        if constexpr (LABELS) {
            int64_t li = cb + lane;
            li = li < end ? li : end - 1;
            glds4(labels + li, slot + UNITS * 1024);
        }
    };
    for (int64_t kk = 0; kk < kRingSlots - 1 && kk < nw; ++kk) issue(kk);
    for (int64_t kk = 0; kk < nw; ++kk) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the slot about to be refilled (chunk kk - 1) has been read
        if (kk + kRingSlots - 1 < nw) issue(kk + kRingSlots - 1);
        // operations younger than chunk kk's DMAs: the DMAs of the chunks requested after it and the stores of the chunks
        // processed since it was requested
        const int64_t ahead = nw - 1 - kk < kRingSlots - 1 ? nw - 1 - kk : kRingSlots - 1;
        const int64_t since = kk < kRingSlots - 1 ? kk : kRingSlots - 1;
        vm_wait_dyn((int)(ahead * OPS + since * stores_per_chunk));
        const char* slot = my + (kk % kRingSlots) * SLOT;
        double x[D];
#pragma unroll
        for (int c = 0; c < D; c += 2) {
            const double2 v = *reinterpret_cast<const double2*>(slot + lane * (D * 8) + c * 8);
            x[c] = v.x;
            x[c + 1] = v.y;
        }
        int32_t old = 0;
        if constexpr (LABELS) old = *reinterpret_cast<const int32_t*>(slot + UNITS * 1024 + lane * 4);
        const int64_t i = begin + (wave + 4 * kk) * 64 + lane;
        if (i < end) f(x, i, old);
    }
    vm_wait<0>();
}
template <int D, bool LABELS>
constexpr size_t ring_lds_bytes() { return (size_t)4 * kRingSlots * ((D / 2) * 1024 + (LABELS ? 256 : 0)); }

// acc layout per block: [sums k*d | counts k | inertia | changed]
__global__ __launch_bounds__(kKmThreads) void kmeans_step_kernel(const double* __restrict__ P, int64_t n, int d,
                                                                 const double* __restrict__ offset,
                                                                 const double* __restrict__ centers, int k,
                                                                 int32_t* __restrict__ labels,
                                                                 double* __restrict__ mindist,
                                                                 double* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int W = k * d + k;                       // per-wave accumulator width
    double* s_c = reinterpret_cast<double*>(smem);  // [k][d]
    double* s_cn = s_c + k * d;                     // [k]
    double* s_acc = s_cn + k;                       // [4 waves][W]
    double* s_misc = s_acc + 4 * W;                 // [256] inertia, then [256] changed
    double off[kKmMaxD];
#pragma unroll
    for (int c = 0; c < kKmMaxD; ++c) off[c] = (offset && c < d) ? offset[c] : 0.0;
    const int t = threadIdx.x;
    const int wave = t >> 6;
    for (int i = t; i < k * d; i += kKmThreads) s_c[i] = centers[i];
    for (int i = t; i < 4 * W; i += kKmThreads) s_acc[i] = 0.0;
    __syncthreads();
    for (int j = t; j < k; j += kKmThreads) {
        double s = 0.0;
        for (int c = 0; c < d; ++c) s += s_c[j * d + c] * s_c[j * d + c];
        s_cn[j] = s;
    }
    __syncthreads();

    double* my_acc = s_acc + wave * W;
    double inertia = 0.0;
    double changed = 0.0;
    // contiguous block of points per workgroup, thread-strided inside
    const int64_t per_block = (n + gridDim.x - 1) / gridDim.x;
    const int64_t begin = (int64_t)blockIdx.x * per_block;
    const int64_t end = begin + per_block < n ? begin + per_block : n;
    for (int64_t i = begin + t; i < end; i += kKmThreads) {
        double x[kKmMaxD];
        load_point(P, i, d, x);
        if (offset) {
#pragma unroll
            for (int c = 0; c < kKmMaxD; ++c)
                if (c < d) x[c] -= off[c];  // X -= X_mean, as KMeans.fit does
        }
        // label = argmin_j (|c_j|^2 - 2 x.c_j), first minimum wins (sklearn lloyd_iter_chunked_dense)
        int best = 0;
        double bestv = INFINITY;
        for (int j = 0; j < k; ++j) {
            double dot = 0.0;
#pragma unroll
            for (int c = 0; c < kKmMaxD; ++c)
                if (c < d) dot += x[c] * s_c[j * d + c];
            const double v = s_cn[j] - 2.0 * dot;
            if (v < bestv) {
                bestv = v;
                best = j;
            }
        }
        double dist = 0.0;
#pragma unroll
        for (int c = 0; c < kKmMaxD; ++c)
            if (c < d) {
                const double df = x[c] - s_c[best * d + c];
                dist += df * df;
            }
        inertia += dist;
        if (mindist) mindist[i] = dist;
        if (labels[i] != best) changed += 1.0;
        labels[i] = best;
#pragma unroll
        for (int c = 0; c < kKmMaxD; ++c)
            if (c < d) atomicAdd(&my_acc[best * d + c], x[c]);
        atomicAdd(&my_acc[k * d + best], 1.0);
    }
    s_misc[t] = inertia;
    s_misc[kKmThreads + t] = changed;
    __syncthreads();
    double* my_part = part + (int64_t)blockIdx.x * (W + 2);
    for (int i = t; i < W; i += kKmThreads) my_part[i] = ((s_acc[i] + s_acc[W + i]) + s_acc[2 * W + i]) + s_acc[3 * W + i];
    if (t < 2) {
        double s = 0.0;
        for (int q = 0; q < kKmThreads; ++q) s += s_misc[t * kKmThreads + q];
        my_part[W + t] = s;
    }
}

// The same pass for the common small shapes (D <= 4 coordinates, k <= KMAX <= 16 clusters) without a single atomic:
// every thread keeps the sums and counts of all clusters in registers (a predicated add per cluster and coordinate;
// the LDS float64 atomics of the general kernel serialise on the few hot addresses and held it to 1.2 TB/s), waves
// combine with a butterfly reduce-scatter, then wave order, then block order.  Streams 8 D + 8 bytes per point.
template <int D, int KMAX, bool RING = false>
__global__ __launch_bounds__(kKmThreads, (KMAX * (D + 1) > 48 ? 1 : 2)) void kmeans_step_reg_kernel(const double* __restrict__ P, int64_t n,
                                                                     const double* __restrict__ offset,
                                                                     const double* __restrict__ centers, int k,
                                                                     int32_t* __restrict__ labels, double* __restrict__ mindist,
                                                                     double* __restrict__ part) {
    constexpr int WR = KMAX * D + KMAX + 2;   // sums | counts | inertia | changed
    __shared__ double s_c[KMAX * D];
    __shared__ double s_cn[KMAX];
    __shared__ double s_red[4][WR];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    double off[D];
#pragma unroll
    for (int c = 0; c < D; ++c) off[c] = offset ? offset[c] : 0.0;
    for (int i = t; i < KMAX * D; i += kKmThreads) s_c[i] = i < k * D ? centers[i] : 0.0;
    __syncthreads();
    if (t < KMAX) {
        double s = 0.0;
#pragma unroll
        for (int c = 0; c < D; ++c) s += s_c[t * D + c] * s_c[t * D + c];
        s_cn[t] = s;
    }
    __syncthreads();
    double acc[WR];
#pragma unroll
    for (int i = 0; i < WR; ++i) acc[i] = 0.0;
    const int64_t per_block = (n + gridDim.x - 1) / gridDim.x;
    const int64_t begin = (int64_t)blockIdx.x * per_block;
    const int64_t end = begin + per_block < n ? begin + per_block : n;
    // U points per thread are requested before any of them is used: with one point per thread in flight a wave keeps
    // 2 KB outstanding and the pass is bound by load latency (1.4 TB/s), not by HBM
    constexpr int U = 2;   // points per thread and batch; two batches alternate (below): four points in flight, two waves per SIMD
    // one point: label, inertia, changed count, sums (the arithmetic is the same whichever way the point was loaded)
    auto take_point = [&](const double (&xr)[D], int64_t i, int32_t old) {
        double x[D];
#pragma unroll
        for (int c = 0; c < D; ++c) x[c] = xr[c] - off[c];   // X -= X_mean, as KMeans.fit does (a zero offset changes nothing)
        int best = 0;
        double bestv = INFINITY;
#pragma unroll
        for (int j = 0; j < KMAX; ++j) {
            if (KMAX <= 8 || j < k) {   // KMAX <= 8: the kernel is instantiated for the exact count
                double dot = 0.0;
#pragma unroll
                for (int c = 0; c < D; ++c) dot += x[c] * s_c[j * D + c];
                const double v = s_cn[j] - 2.0 * dot;
                if (v < bestv) {   // first minimum wins (sklearn lloyd_iter_chunked_dense)
                    bestv = v;
                    best = j;
                }
            }
        }
        double dist = 0.0;
#pragma unroll
        for (int c = 0; c < D; ++c) {
            const double df = x[c] - s_c[best * D + c];
            dist += df * df;
        }
        acc[KMAX * D + KMAX] += dist;
        if (mindist) mindist[i] = dist;
        if (old != best) acc[KMAX * D + KMAX + 1] += 1.0;
        labels[i] = best;
        // acc[j] += (best == j) ? x : 0 as ONE fused multiply-add per value with a 0 / 1 multiplier: fma(1, x, acc) is the
        // correctly rounded acc + x and fma(0, x, acc) is acc (finite x) -- the sums a select + add gives
#pragma unroll
        for (int j = 0; j < KMAX; ++j) {
            if (KMAX <= 8 || j < k) {
                // the 0 / 1 multiplier as ONE select of its high word, opaque to the optimiser (which otherwise turns
                // fma(select(c, 1, 0), x, acc) back into select(c, acc + x, acc): two v_cndmask and an add per value)
                unsigned mhi = best == j ? 0x3FF00000u : 0u;
                asm volatile("" : "+v"(mhi));
                const double mj = __hiloint2double((int)mhi, 0);
#pragma unroll
                for (int c = 0; c < D; ++c) acc[j * D + c] = fma(mj, x[c], acc[j * D + c]);
                acc[KMAX * D + j] += mj;
            }
        }
    };
    // A batch = U points per thread, requested together.  TWO batches alternate: the loads of the next one are issued before the
    // current one is computed, so a wave has memory requests in flight while it computes (one batch at a time left the pass at
    // the SUM of its load latency and its ~150 float64 instructions per point: 3.8 TB/s, round 3).
    struct Batch {
        PairUnits pu[U];      // D == 4: contiguous 16-byte units, halves swapped between lane pairs (pair_issue / pair_finish)
        double xs[U][D];      // other D: the lane's own points
        int32_t olds[U];
    };
    const int64_t nchunk = (end - begin + 63) / 64;                                  // D == 4: 64-point chunks, wave w takes w, w + 4, ...
    const int64_t step = D == 4 ? 4 * U : (int64_t)kKmThreads * U;                    // batch stride in chunks / points
    const int64_t first = D == 4 ? wave : begin + t, last = D == 4 ? nchunk : end;   // batch positions of this wave / thread
    auto issue = [&](Batch& b, int64_t pos) {
        if constexpr (D == 4) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t cb = begin + (pos + 4 * u) * 64;
                pair_issue(P, cb, end, lane, b.pu[u]);
                const int64_t i = pair_point(cb, lane);
                b.olds[u] = labels[i < end ? i : end - 1];
            }
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t i = pos + (int64_t)u * kKmThreads;
                const int64_t ic = i < end ? i : end - 1;   // unconditional loads from a clamped index (see pair_issue)
                const double* p = P + ic * D;
                if constexpr (D % 2 == 0) {
#pragma unroll
                    for (int c = 0; c < D; c += 2) {
                        const double2 v = *reinterpret_cast<const double2*>(p + c);
                        b.xs[u][c] = v.x;
                        b.xs[u][c + 1] = v.y;
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < D; ++c) b.xs[u][c] = p[c];
                }
                b.olds[u] = labels[ic];
            }
        }
    };
    auto process = [&](Batch& b, int64_t pos) {
        if constexpr (D == 4) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t cb = begin + (pos + 4 * u) * 64;
                double xr[D];
                pair_finish(b.pu[u], lane, xr);   // every lane of the wave takes part in the swap
                const int64_t i = pair_point(cb, lane);
                if (i < end) take_point(xr, i, b.olds[u]);
            }
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t i = pos + (int64_t)u * kKmThreads;
                if (i < end) take_point(b.xs[u], i, b.olds[u]);
            }
        }
    };
    if constexpr (RING && (D == 2 || D == 4)) {
        // the point stream through the wave's LDS-DMA ring (ring_stream above): launched when mindist == nullptr, so the only
        // vector-memory store per chunk is the label store of take_point
        extern __shared__ __attribute__((aligned(16))) char s_ring[];
        ring_stream<D, true>(P, labels, begin, end, s_ring, wave, lane, 1, [&](const double (&xr)[D], int64_t i, int32_t old) { take_point(xr, i, old); });
    } else {
        Batch ba, bb;
        int64_t pos = first;
        if (pos < last) issue(ba, pos);
        for (; pos < last; pos += 2 * step) {
            if (pos + step < last) issue(bb, pos + step);
            process(ba, pos);
            if (pos + 2 * step < last) issue(ba, pos + 2 * step);
            if (pos + step < last) process(bb, pos + step);
        }
    }
    {
        int base = 0, dup = 0;
        const int cnt = butterfly_sum<WR, 32, WR, double>(acc, lane, base, dup);
        if ((lane & dup) == 0) {
#pragma unroll
            for (int i = 0; i < WR; ++i)
                if (i < cnt) s_red[wave][base + i] = acc[i];
        }
    }
    __syncthreads();
    // output layout of the general kernel: [sums k*d | counts k | inertia | changed]
    double* my_part = part + (int64_t)blockIdx.x * (k * D + k + 2);
    for (int i = t; i < k * D + k + 2; i += kKmThreads) {
        const int src = i < k * D ? i : (i < k * D + k ? KMAX * D + (i - k * D) : KMAX * D + KMAX + (i - k * D - k));
        my_part[i] = ((s_red[0][src] + s_red[1][src]) + s_red[2][src]) + s_red[3][src];
    }
}
typedef void (*km_reg_fn_t)(const double*, int64_t, const double*, const double*, int, int32_t*, double*, double*);
template <int KM>
static km_reg_fn_t km_reg_fn_d(int d, bool ring) {
    if (ring && KM <= 8) {   // (9 .. 16 clusters: the register form; its accumulators leave one wave per SIMD anyway)
        if (d == 2) return kmeans_step_reg_kernel<2, KM, true>;
        if (d == 4) return kmeans_step_reg_kernel<4, KM, true>;
    }
    switch (d) {
        case 1: return kmeans_step_reg_kernel<1, KM>;
        case 2: return kmeans_step_reg_kernel<2, KM>;
        case 3: return kmeans_step_reg_kernel<3, KM>;
        case 4: return kmeans_step_reg_kernel<4, KM>;
        default: return nullptr;
    }
}
// k <= 8: an instantiation for the exact cluster count (the `j < k` tests of a padded count were 158 scalar branches in the
// point loop of the D = 4, KMAX = 8 kernel: every unrolled centroid its own basic block); 9 .. 16 clusters share KMAX = 16
static km_reg_fn_t km_reg_fn(int d, int k, bool ring) {
    switch (k) {
        case 1: return km_reg_fn_d<1>(d, ring);
        case 2: return km_reg_fn_d<2>(d, ring);
        case 3: return km_reg_fn_d<3>(d, ring);
        case 4: return km_reg_fn_d<4>(d, ring);
        case 5: return km_reg_fn_d<5>(d, ring);
        case 6: return km_reg_fn_d<6>(d, ring);
        case 7: return km_reg_fn_d<7>(d, ring);
        case 8: return km_reg_fn_d<8>(d, ring);
        default: return k <= 16 ? km_reg_fn_d<16>(d, false) : nullptr;
    }
}
static bool km_ring_enabled() {
    static const bool on = [] { const char* e = getenv("DCV_KM_RING"); return !(e && e[0] == '0'); }();
    return on;
}

// acc[i] = sum over the blocks of part[b][i]: one workgroup per output, its threads take b = t, t + 256, ... (all loads
// in flight at once; one thread walking 1024 partials with dependent loads cost 0.4 ms -- most of the pass) and a fixed
// LDS tree adds them up.
__global__ __launch_bounds__(256) void kmeans_final_kernel(const double* __restrict__ part, int nblocks, int width, double* __restrict__ acc) {
    __shared__ double s_red[256];
    const int i = blockIdx.x, t = threadIdx.x;
    double s = 0.0;
    for (int b = t; b < nblocks; b += 256) s += part[(int64_t)b * width + i];
    s_red[t] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) s_red[t] += s_red[t + o];
        __syncthreads();
    }
    if (t == 0) acc[i] = s_red[0];
}

// ------------------------------------------------------------------ k-means++ seeding passes
// sklearn _kmeans_plusplus keeps, per point, the squared distance to the closest centre
// chosen so far and, for every new centre, scores a few candidates by the potential sum_i min(closest_i, dist(x_i, cand)).
// Both are streaming passes over the resident points; only the sequential float64 cumsum + searchsorted that turns the
// random draws into candidate indices stays on the host (a parallel prefix sum rounds differently).
// dist = -2 x.c + |c|^2 + |x|^2, clipped at 0 (sklearn euclidean_distances(squared=True) with precomputed norms), every
// step separately rounded.
constexpr int kPpMaxTrials = 8;
template <int D>
__device__ __forceinline__ double pp_dist(const double (&x)[D], const double* __restrict__ c, double cc) {
    double dot = 0.0, xx = 0.0;
#pragma unroll
    for (int q = 0; q < D; ++q) {
        dot = fma(x[q], c[q], dot);
        xx = fma(x[q], x[q], xx);
    }
    const double v = __dadd_rn(__dadd_rn(__dmul_rn(-2.0, dot), cc), xx);
    return v > 0.0 ? v : 0.0;
}
template <int D>
__device__ __forceinline__ void pp_load(const double* __restrict__ P, int64_t i, const double* off, double (&x)[D]) {
    const double* p = P + i * D;
    if constexpr (D % 2 == 0) {
#pragma unroll
        for (int c = 0; c < D; c += 2) {
            const double2 v = *reinterpret_cast<const double2*>(p + c);
            x[c] = v.x - off[c];
            x[c + 1] = v.y - off[c + 1];
        }
    } else {
#pragma unroll
        for (int c = 0; c < D; ++c) x[c] = p[c] - off[c];
    }
}
// part[block][t] = sum over the block's points of min(closest_i, dist(x_i, cand_t))
template <int D>
__global__ __launch_bounds__(kKmThreads) void kmeanspp_potentials_kernel(const double* __restrict__ P, int64_t n,
                                                                         const double* __restrict__ offset,
                                                                         const double* __restrict__ cand, int T,
                                                                         const double* __restrict__ closest, double* __restrict__ part) {
    __shared__ double s_c[kPpMaxTrials * D];
    __shared__ double s_cc[kPpMaxTrials];
    __shared__ double s_red[kKmThreads];
    const int t = threadIdx.x;
    double off[D];
#pragma unroll
    for (int c = 0; c < D; ++c) off[c] = offset ? offset[c] : 0.0;
    if (t < T * D) s_c[t] = cand[t];
    __syncthreads();
    if (t < T) {
        double s = 0.0;
#pragma unroll
        for (int c = 0; c < D; ++c) s = fma(s_c[t * D + c], s_c[t * D + c], s);
        s_cc[t] = s;
    }
    __syncthreads();
    double acc[kPpMaxTrials];
#pragma unroll
    for (int q = 0; q < kPpMaxTrials; ++q) acc[q] = 0.0;
    const int64_t per_block = (n + gridDim.x - 1) / gridDim.x;
    const int64_t begin = (int64_t)blockIdx.x * per_block;
    const int64_t end = begin + per_block < n ? begin + per_block : n;
    for (int64_t i = begin + t; i < end; i += kKmThreads) {
        double x[D];
        pp_load<D>(P, i, off, x);
        const double cl = closest[i];
#pragma unroll
        for (int q = 0; q < kPpMaxTrials; ++q)
            if (q < T) {
                const double v = pp_dist<D>(x, s_c + q * D, s_cc[q]);
                acc[q] += v < cl ? v : cl;
            }
    }
    for (int q = 0; q < T; ++q) {
        s_red[t] = acc[q];
        __syncthreads();
        for (int o = kKmThreads / 2; o > 0; o >>= 1) {
            if (t < o) s_red[t] += s_red[t + o];
            __syncthreads();
        }
        if (t == 0) part[(int64_t)blockIdx.x * T + q] = s_red[0];
        __syncthreads();
    }
}
// closest_i = first ? dist(x_i, c) : min(closest_i, dist(x_i, c)) ; part[block] = sum of the new values
template <int D>
__global__ __launch_bounds__(kKmThreads) void kmeanspp_update_kernel(const double* __restrict__ P, int64_t n,
                                                                     const double* __restrict__ offset,
                                                                     const double* __restrict__ centre, int first,
                                                                     double* __restrict__ closest, double* __restrict__ part) {
    __shared__ double s_red[kKmThreads];
    const int t = threadIdx.x;
    double off[D], c[D];
    double cc = 0.0;
#pragma unroll
    for (int q = 0; q < D; ++q) {
        off[q] = offset ? offset[q] : 0.0;
        c[q] = centre[q];
        cc = fma(c[q], c[q], cc);
    }
    double acc = 0.0;
    const int64_t per_block = (n + gridDim.x - 1) / gridDim.x;
    const int64_t begin = (int64_t)blockIdx.x * per_block;
    const int64_t end = begin + per_block < n ? begin + per_block : n;
    for (int64_t i = begin + t; i < end; i += kKmThreads) {
        double x[D];
        pp_load<D>(P, i, off, x);
        double v = pp_dist<D>(x, c, cc);
        if (!first) {
            const double cl = closest[i];
            v = v < cl ? v : cl;
        }
        closest[i] = v;
        acc += v;
    }
    s_red[t] = acc;
    __syncthreads();
    for (int o = kKmThreads / 2; o > 0; o >>= 1) {
        if (t < o) s_red[t] += s_red[t + o];
        __syncthreads();
    }
    if (t == 0) part[blockIdx.x] = s_red[0];
}

// ------------------------------------------------------------------ nearest sample per centroid
// numpy: sqrt(add.reduce((x - c)**2, axis=1)); pairwise summation degenerates to a sequential
// sum for d < 8 and to 8 interleaved partial sums for 8 <= d <= 128.
__device__ __forceinline__ double np_norm(const double* x, const double* c, int d) {
    // separately rounded multiply / add, as NumPy's ufunc loops do.  HIP's __dmul_rn / __dadd_rn are plain operators: without
    // the pragma hipcc contracts df * df + res into one fma (seen in round 4: a returned distance one ulp off numpy's)
#pragma clang fp contract(off)
    double sq[kKmMaxD];
#pragma unroll
    for (int q = 0; q < kKmMaxD; ++q)
        if (q < d) {
            const double df = x[q] - c[q];
            sq[q] = __dmul_rn(df, df);
        }
    double res;
    if (d < 8) {
        res = 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (q < d) res = __dadd_rn(res, sq[q]);
    } else {
        double r[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) r[q] = sq[q];
        const int full = d - (d % 8);
#pragma unroll
        for (int q = 8; q < kKmMaxD; q += 8)
            if (q < full) {
#pragma unroll
                for (int u = 0; u < 8; ++u) r[u] = __dadd_rn(r[u], sq[q + u]);
            }
        res = __dadd_rn(__dadd_rn(__dadd_rn(r[0], r[1]), __dadd_rn(r[2], r[3])),
                        __dadd_rn(__dadd_rn(r[4], r[5]), __dadd_rn(r[6], r[7])));
#pragma unroll
        for (int q = 8; q < kKmMaxD; ++q)
            if (q >= full && q < d) res = __dadd_rn(res, sq[q]);
    }
    return __dsqrt_rn(res);   // numpy's sqrt is correctly rounded; the plain device sqrt is allowed 1 ulp (seen: one distance in eleven off by an ulp)
}

// grid.x = point blocks, grid.y = centroid.  part[(j*nblocks + b)] = (dist, row)
__global__ __launch_bounds__(kKmThreads) void nearest_rows_kernel(const double* __restrict__ P, int64_t n, int d,
                                                                  const double* __restrict__ centers,
                                                                  double* __restrict__ pdist, int64_t* __restrict__ prow) {
    __shared__ double s_d[kKmThreads];
    __shared__ int64_t s_i[kKmThreads];
    const int j = blockIdx.y;
    const int t = threadIdx.x;
    double c[kKmMaxD];
#pragma unroll
    for (int q = 0; q < kKmMaxD; ++q) c[q] = q < d ? centers[j * d + q] : 0.0;
    const int64_t per_block = (n + gridDim.x - 1) / gridDim.x;
    const int64_t begin = (int64_t)blockIdx.x * per_block;
    const int64_t end = begin + per_block < n ? begin + per_block : n;
    double best = INFINITY;
    int64_t besti = INT64_MAX;
    for (int64_t i = begin + t; i < end; i += kKmThreads) {
        double x[kKmMaxD];
        load_point(P, i, d, x);
        const double v = np_norm(x, c, d);
        if (v < best) {  // increasing i: strict '<' keeps the first index
            best = v;
            besti = i;
        }
    }
    s_d[t] = best;
    s_i[t] = besti;
    __syncthreads();
    for (int off = kKmThreads / 2; off > 0; off >>= 1) {
        if (t < off) {
            const double od = s_d[t + off];
            const int64_t oi = s_i[t + off];
            if (od < s_d[t] || (od == s_d[t] && oi < s_i[t])) {
                s_d[t] = od;
                s_i[t] = oi;
            }
        }
        __syncthreads();
    }
    if (t == 0) {
        pdist[(int64_t)j * gridDim.x + blockIdx.x] = s_d[0];
        prow[(int64_t)j * gridDim.x + blockIdx.x] = s_i[0];
    }
}

// ONE pass over the points for all centroids of a chunk (D <= 4 coordinates: the CV spaces of the pipeline): a thread keeps
// the running minimum (distance, row) of up to KC centroids in registers and streams its points once, four in flight.  The
// per-centroid form above re-reads every point k times (k = 6, 20M x 4: 3.84 GB moved for 0.64 GB of points, 597 us);
// centroids beyond KC take further chunks on grid.y.  Same arithmetic (np_norm), same order per thread (increasing row,
// strict '<'), same lexicographic (distance, row) combination: bit-identical rows.
constexpr int kNearKC = 8;
// KX > 0: the kernel is instantiated for exactly KX centroids per chunk (no per-centroid `j < kc` test in the point loop)
template <int D, bool RING = false, int KX = 0>
__global__ __launch_bounds__(kKmThreads) void nearest_rows_multi_kernel(const double* __restrict__ P, int64_t n, const double* __restrict__ centers,
                                                                        int k, double* __restrict__ pdist, int64_t* __restrict__ prow) {
    constexpr int KC = KX > 0 ? KX : kNearKC, U = 4;
    __shared__ double s_c[KC][D];
    __shared__ double s_d[4][KC];
    __shared__ int64_t s_i[4][KC];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int j0 = blockIdx.y * KC;
    const int kc = KX > 0 ? KX : (k - j0 < KC ? k - j0 : KC);
    if (t < KC * D) {
        const int j = t / D, q = t - j * D;
        s_c[j][q] = j < kc ? centers[(int64_t)(j0 + j) * D + q] : 0.0;
    }
    __syncthreads();
    double best[KC], bthr[KC];   // smallest sum of squares so far, and the value below which a root is certainly smaller
    int64_t besti[KC];
#pragma unroll
    for (int j = 0; j < KC; ++j) {
        best[j] = INFINITY;
        bthr[j] = INFINITY;
        besti[j] = INT64_MAX;
    }
    const int64_t per_block = (n + gridDim.x - 1) / gridDim.x;
    const int64_t begin = (int64_t)blockIdx.x * per_block;
    const int64_t end = begin + per_block < n ? begin + per_block : n;
    // one point against the chunk's centroids
    auto take_point = [&](const double (&xr)[D], int64_t i) {
#pragma unroll
        for (int j = 0; j < KC; ++j) {
            if (KX > 0 || j < kc) {
                // np_norm for D < 8: sequentially added, separately rounded squares (no fma contraction: see np_norm)
                double res = 0.0;
                {
#pragma clang fp contract(off)
#pragma unroll
                    for (int q = 0; q < D; ++q) {
                        const double df = xr[q] - s_c[j][q];
                        const double sq = df * df;
                        res = res + sq;
                    }
                }
                // numpy takes argmin over the ROUNDED distances sqrt(res): the first index wins among equal ones.  The
                // square root (a ~30-instruction float64 sequence, 120 M of them at 20M x 6: what bounded this pass) is
                // needed only where two sums of squares are so close that their roots could round to the same double:
                // below best * (1 - 2^-50) the rounded root is strictly smaller, at or above best it is not smaller.
                if (res < best[j]) {
                    if (res < bthr[j] || __dsqrt_rn(res) < __dsqrt_rn(best[j])) {
                        best[j] = res;
                        bthr[j] = res * (1.0 - 0x1p-50);
                        besti[j] = i;
                    }
                }
            }
        }
    };
    // two alternating batches of U points per thread: the next one's loads are in flight while this one is compared (see
    // kmeans_step_reg_kernel); D == 4 reads contiguous 16-byte units and swaps halves between lane pairs
    struct Batch {
        PairUnits pu[U];
        double xs[U][D];
    };
    const int64_t nchunk = (end - begin + 63) / 64;
    const int64_t step = D == 4 ? 4 * U : (int64_t)kKmThreads * U;
    const int64_t first = D == 4 ? wave : begin + t, last = D == 4 ? nchunk : end;
    auto issue = [&](Batch& b, int64_t pos) {
        if constexpr (D == 4) {
#pragma unroll
            for (int u = 0; u < U; ++u) pair_issue(P, begin + (pos + 4 * u) * 64, end, lane, b.pu[u]);
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t i = pos + (int64_t)u * kKmThreads;
                const double* p = P + (i < end ? i : end - 1) * D;
                if constexpr (D % 2 == 0) {
#pragma unroll
                    for (int q = 0; q < D; q += 2) {
                        const double2 v = *reinterpret_cast<const double2*>(p + q);
                        b.xs[u][q] = v.x;
                        b.xs[u][q + 1] = v.y;
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < D; ++q) b.xs[u][q] = p[q];
                }
            }
        }
    };
    auto process = [&](Batch& b, int64_t pos) {
        if constexpr (D == 4) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                double xr[D];
                pair_finish(b.pu[u], lane, xr);
                const int64_t i = pair_point(begin + (pos + 4 * u) * 64, lane);
                if (i < end) take_point(xr, i);
            }
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t i = pos + (int64_t)u * kKmThreads;
                if (i < end) take_point(b.xs[u], i);
            }
        }
    };
    if constexpr (RING && (D == 2 || D == 4)) {   // the point stream through the wave's LDS-DMA ring (ring_stream): no store in the loop
        extern __shared__ __attribute__((aligned(16))) char s_ring[];
        ring_stream<D, false>(P, nullptr, begin, end, s_ring, wave, lane, 0, [&](const double (&xr)[D], int64_t i, int32_t) { take_point(xr, i); });
    } else {
        Batch ba, bb;
        int64_t pos = first;
        if (pos < last) issue(ba, pos);
        for (; pos < last; pos += 2 * step) {
            if (pos + step < last) issue(bb, pos + step);
            process(ba, pos);
            if (pos + 2 * step < last) issue(ba, pos + 2 * step);
            if (pos + step < last) process(bb, pos + step);
        }
    }
#pragma unroll
    for (int j = 0; j < KC; ++j) {
        double b = __dsqrt_rn(best[j]);   // from here on the rounded distances, as numpy's argmin sees them
        int64_t bi = besti[j];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double v2 = __shfl_down(b, off, 64);
            const int64_t i2 = __shfl_down(bi, off, 64);
            if (v2 < b || (v2 == b && i2 < bi)) {
                b = v2;
                bi = i2;
            }
        }
        if (lane == 0) {
            s_d[wave][j] = b;
            s_i[wave][j] = bi;
        }
    }
    __syncthreads();
    if (t < kc) {
        double b = s_d[0][t];
        int64_t bi = s_i[0][t];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const double v2 = s_d[w][t];
            const int64_t i2 = s_i[w][t];
            if (v2 < b || (v2 == b && i2 < bi)) {
                b = v2;
                bi = i2;
            }
        }
        pdist[(int64_t)(j0 + t) * gridDim.x + blockIdx.x] = b;
        prow[(int64_t)(j0 + t) * gridDim.x + blockIdx.x] = bi;
    }
}

// One wave per centroid: the lanes take the block partials b = lane, lane + 64, ... with four loads in flight each, then
// a shuffle tree keeps the lexicographic minimum (distance, row) -- the same answer as a serial walk in block order (ties go
// to the smaller row).  (One THREAD per centroid walked ~512 partials with dependent loads: 167 us for a 6-centroid final.)
__global__ __launch_bounds__(64) void nearest_rows_final(const double* __restrict__ pdist, const int64_t* __restrict__ prow, int nblocks, int k,
                                                         int64_t row_offset, double* __restrict__ dist, int64_t* __restrict__ rows) {
    const int j = blockIdx.x, lane = threadIdx.x;
    if (j >= k) return;
    double best = INFINITY;
    int64_t besti = INT64_MAX;
    const double* pd = pdist + (int64_t)j * nblocks;
    const int64_t* pr = prow + (int64_t)j * nblocks;
    for (int b0 = lane; b0 < nblocks; b0 += 4 * 64) {
        double v[4];
        int64_t i[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int b = b0 + 64 * u;
            v[u] = b < nblocks ? pd[b] : INFINITY;
            i[u] = b < nblocks ? pr[b] : INT64_MAX;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (v[u] < best || (v[u] == best && i[u] < besti)) {
                best = v[u];
                besti = i[u];
            }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const double v2 = __shfl_down(best, off, 64);
        const int64_t i2 = __shfl_down(besti, off, 64);
        if (v2 < best || (v2 == best && i2 < besti)) {
            best = v2;
            besti = i2;
        }
    }
    if (lane == 0) {
        dist[j] = best;
        rows[j] = besti == INT64_MAX ? -1 : besti + row_offset;
    }
}

// ------------------------------------------------------------------ 1-NN of supplementary points
__global__ __launch_bounds__(kKmThreads) void nearest_point_kernel(const double* __restrict__ train, int64_t n_train,
                                                                   const double* __restrict__ sup, int64_t n_sup, int d,
                                                                   int64_t* __restrict__ nn) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* tile = reinterpret_cast<double*>(smem);  // [256][d]
    const int t = threadIdx.x;
    const int64_t i = (int64_t)blockIdx.x * kKmThreads + t;
    double x[kKmMaxD];
    if (i < n_sup) load_point(sup, i, d, x);
    double best = INFINITY;
    int64_t besti = -1;
    for (int64_t base = 0; base < n_train; base += kKmThreads) {
        const int64_t cnt = n_train - base < kKmThreads ? n_train - base : kKmThreads;
        __syncthreads();
        for (int64_t q = t; q < cnt * d; q += kKmThreads) tile[q] = train[base * d + q];
        __syncthreads();
        if (i < n_sup) {
            for (int q = 0; q < (int)cnt; ++q) {
                double s = 0.0;
#pragma unroll
                for (int c = 0; c < kKmMaxD; ++c)
                    if (c < d) {
                        const double df = x[c] - tile[q * d + c];
                        s += df * df;
                    }
                if (s < best) {
                    best = s;
                    besti = base + q;
                }
            }
        }
    }
    if (i < n_sup) nn[i] = besti;
}

}  // namespace dcv

using namespace dcv;

extern "C" size_t dcv_kmeans_workspace(int64_t n, int32_t d, int32_t k) {
    if (n <= 0 || d <= 0 || k <= 0) return 0;
    return (size_t)kKmMaxBlocks * ((size_t)k * d + k + 2) * sizeof(double);
}

extern "C" int dcv_kmeans_step(const double* P_d, int64_t n, int32_t d, const double* offset_d,
                               const double* centers_d, int32_t k, int32_t* labels_d, double* acc_d, double* mindist_d, void* ws_d, size_t ws_bytes,
                               void* stream) {
    DCV_REQUIRE(P_d && centers_d && labels_d && acc_d && n > 0, "dcv_kmeans_step: bad arguments");
    DCV_REQUIRE(d >= 1 && d <= kKmMaxD && k >= 1 && k <= kKmMaxK, "dcv_kmeans_step: d=%d (1..16) k=%d (1..64) unsupported", d, k);
    DCV_REQUIRE(ws_d && ws_bytes >= dcv_kmeans_workspace(n, d, k), "dcv_kmeans_step: workspace too small");
    hipStream_t s = as_stream(stream);
    const int nb = km_blocks(n);
    const int W = k * d + k;
    const size_t lds = ((size_t)k * d + k + 4 * W + 2 * kKmThreads) * sizeof(double);
    double* part = static_cast<double*>(ws_d);
    static const bool no_reg = [] { const char* e = getenv("DCV_KMEANS_ATOMIC"); return e && e[0] == '1'; }();   // diagnostic: general kernel
    const bool ring = km_ring_enabled() && mindist_d == nullptr && (d == 2 || d == 4) && k <= 8;
    const size_t ring_lds = ring ? (d == 4 ? ring_lds_bytes<4, true>() : ring_lds_bytes<2, true>()) : 0;
    if (km_reg_fn_t reg = no_reg ? nullptr : km_reg_fn(d, k, ring)) {
        // One round of the chip: the register kernel holds ~150 VGPRs (3 blocks per CU), and 1024 equal blocks on 768
        // slots is two rounds with the second a third full (3.5 TB/s).  Blocks = CUs x resident blocks per CU.
        int nbr = nb;
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(reg), kKmThreads, ring_lds) == hipSuccess && per_cu > 0) {
            const int64_t one_round = (int64_t)num_cus() * per_cu;
            if (nbr > one_round) nbr = (int)one_round;
        } else {
            (void)hipGetLastError();
        }
        hipLaunchKernelGGL(reg, dim3(nbr), dim3(kKmThreads), ring_lds, s, P_d, n, offset_d, centers_d, (int)k, labels_d, mindist_d, part);
        DCV_CHECK_LAUNCH();
        hipLaunchKernelGGL(kmeans_final_kernel, dim3(W + 2), dim3(256), 0, s, part, nbr, W + 2, acc_d);
        DCV_CHECK_LAUNCH();
        return DCV_OK;
    } else
        hipLaunchKernelGGL(kmeans_step_kernel, dim3(nb), dim3(kKmThreads), lds, s, P_d, n, d, offset_d, centers_d, k, labels_d, mindist_d, part);
    DCV_CHECK_LAUNCH();
    hipLaunchKernelGGL(kmeans_final_kernel, dim3(W + 2), dim3(256), 0, s, part, nb, W + 2, acc_d);
    DCV_CHECK_LAUNCH();
    return DCV_OK;
}

extern "C" size_t dcv_nearest_rows_workspace(int64_t n, int32_t d, int32_t k) {
    (void)d;
    if (n <= 0 || k <= 0) return 0;
    return (size_t)kKmMaxBlocks * k * (sizeof(double) + sizeof(int64_t));
}

extern "C" int dcv_nearest_rows(const double* P_d, int64_t n, int32_t d, const double* centers_d, int32_t k,
                                int64_t row_offset, double* dist_d, int64_t* rows_d, void* ws_d, size_t ws_bytes,
                                void* stream) {
    DCV_REQUIRE(P_d && centers_d && dist_d && rows_d && n > 0, "dcv_nearest_rows: bad arguments");
    DCV_REQUIRE(d >= 1 && d <= kKmMaxD && k >= 1 && k <= 1024, "dcv_nearest_rows: d=%d k=%d unsupported", d, k);
    DCV_REQUIRE(ws_d && ws_bytes >= dcv_nearest_rows_workspace(n, d, k), "dcv_nearest_rows: workspace too small");
    hipStream_t s = as_stream(stream);
    const int nb = km_blocks(n);
    double* pdist = static_cast<double*>(ws_d);
    int64_t* prow = reinterpret_cast<int64_t*>(pdist + (size_t)kKmMaxBlocks * k);
    static const bool multi_off = [] { const char* e = getenv("DCV_NEAREST_PER_CENTROID"); return e && e[0] == '1'; }();
    const dim3 gm(nb, (unsigned)cdiv(k, kNearKC));
    if (d <= 4 && !multi_off) {   // one pass over the points per chunk of kNearKC centroids
        const bool ring = km_ring_enabled();
        const size_t rl2 = ring_lds_bytes<2, false>(), rl4 = ring_lds_bytes<4, false>();
        switch (d) {
            case 1: hipLaunchKernelGGL(nearest_rows_multi_kernel<1>, gm, dim3(kKmThreads), 0, s, P_d, n, centers_d, (int)k, pdist, prow); break;
            case 2:
                if (ring) hipLaunchKernelGGL((nearest_rows_multi_kernel<2, true>), gm, dim3(kKmThreads), rl2, s, P_d, n, centers_d, (int)k, pdist, prow);
                else hipLaunchKernelGGL(nearest_rows_multi_kernel<2>, gm, dim3(kKmThreads), 0, s, P_d, n, centers_d, (int)k, pdist, prow);
                break;
            case 3: hipLaunchKernelGGL(nearest_rows_multi_kernel<3>, gm, dim3(kKmThreads), 0, s, P_d, n, centers_d, (int)k, pdist, prow); break;
            default:
                if (ring && k <= kNearKC) {   // one chunk of exactly k centroids
                    switch (k) {
#define DCV_NR4(K) case K: hipLaunchKernelGGL((nearest_rows_multi_kernel<4, true, K>), gm, dim3(kKmThreads), rl4, s, P_d, n, centers_d, (int)k, pdist, prow); break;
                        DCV_NR4(1) DCV_NR4(2) DCV_NR4(3) DCV_NR4(4) DCV_NR4(5) DCV_NR4(6) DCV_NR4(7) DCV_NR4(8)
#undef DCV_NR4
                    }
                } else if (ring) hipLaunchKernelGGL((nearest_rows_multi_kernel<4, true>), gm, dim3(kKmThreads), rl4, s, P_d, n, centers_d, (int)k, pdist, prow);
                else hipLaunchKernelGGL(nearest_rows_multi_kernel<4>, gm, dim3(kKmThreads), 0, s, P_d, n, centers_d, (int)k, pdist, prow);
                break;
        }
    } else {
        hipLaunchKernelGGL(nearest_rows_kernel, dim3(nb, k), dim3(kKmThreads), 0, s, P_d, n, d, centers_d, pdist, prow);
    }
    DCV_CHECK_LAUNCH();
    hipLaunchKernelGGL(nearest_rows_final, dim3((unsigned)k), dim3(64), 0, s, pdist, prow, nb, k, row_offset, dist_d, rows_d);
    DCV_CHECK_LAUNCH();
    return DCV_OK;
}

extern "C" int dcv_nearest_point(const double* train_d, int64_t n_train, const double* sup_d, int64_t n_sup, int32_t d,
                                 int64_t* nn_d, void* stream) {
    DCV_REQUIRE(train_d && sup_d && nn_d && n_train > 0 && n_sup > 0, "dcv_nearest_point: bad arguments");
    DCV_REQUIRE(d >= 1 && d <= kKmMaxD, "dcv_nearest_point: d=%d unsupported", d);
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(nearest_point_kernel, dim3((unsigned)cdiv(n_sup, kKmThreads)), dim3(kKmThreads),
                       (size_t)kKmThreads * d * sizeof(double), s, train_d, n_train, sup_d, n_sup, d, nn_d);
    DCV_CHECK_LAUNCH();
    return DCV_OK;
}

// ------------------------------------------------------------------ k-means++ seeding (C-ABI)
#define DCV_PP_DISPATCH(KERNEL, ...)                                                                   \
    switch (d) {                                                                                      \
        case 1: hipLaunchKernelGGL(KERNEL<1>, __VA_ARGS__); break;                                    \
        case 2: hipLaunchKernelGGL(KERNEL<2>, __VA_ARGS__); break;                                    \
        case 3: hipLaunchKernelGGL(KERNEL<3>, __VA_ARGS__); break;                                    \
        case 4: hipLaunchKernelGGL(KERNEL<4>, __VA_ARGS__); break;                                    \
        case 5: hipLaunchKernelGGL(KERNEL<5>, __VA_ARGS__); break;                                    \
        case 6: hipLaunchKernelGGL(KERNEL<6>, __VA_ARGS__); break;                                    \
        case 7: hipLaunchKernelGGL(KERNEL<7>, __VA_ARGS__); break;                                    \
        case 8: hipLaunchKernelGGL(KERNEL<8>, __VA_ARGS__); break;                                    \
        case 9: hipLaunchKernelGGL(KERNEL<9>, __VA_ARGS__); break;                                    \
        case 10: hipLaunchKernelGGL(KERNEL<10>, __VA_ARGS__); break;                                  \
        case 11: hipLaunchKernelGGL(KERNEL<11>, __VA_ARGS__); break;                                  \
        case 12: hipLaunchKernelGGL(KERNEL<12>, __VA_ARGS__); break;                                  \
        case 13: hipLaunchKernelGGL(KERNEL<13>, __VA_ARGS__); break;                                  \
        case 14: hipLaunchKernelGGL(KERNEL<14>, __VA_ARGS__); break;                                  \
        case 15: hipLaunchKernelGGL(KERNEL<15>, __VA_ARGS__); break;                                  \
        case 16: hipLaunchKernelGGL(KERNEL<16>, __VA_ARGS__); break;                                  \
        default: set_error("k-means++ passes: d=%d (1..16) unsupported", d); return DCV_EINVAL;       \
    }

extern "C" size_t dcv_kmeanspp_workspace(int64_t n, int32_t trials) {
    if (n <= 0 || trials <= 0) return 0;
    return (size_t)kKmMaxBlocks * (size_t)(trials > 1 ? trials : 1) * sizeof(double);
}

extern "C" int dcv_kmeanspp_potentials(const double* P_d, int64_t n, int32_t d, const double* offset_d, const double* cand_d,
                                       int32_t trials, const double* closest_d, double* pot_d, void* ws_d, size_t ws_bytes, void* stream) {
    DCV_REQUIRE(P_d && cand_d && closest_d && pot_d && n > 0, "dcv_kmeanspp_potentials: bad arguments");
    DCV_REQUIRE(trials >= 1 && trials <= kPpMaxTrials, "dcv_kmeanspp_potentials: trials=%d (1..%d)", trials, kPpMaxTrials);
    DCV_REQUIRE(ws_d && ws_bytes >= dcv_kmeanspp_workspace(n, trials), "dcv_kmeanspp_potentials: workspace too small");
    hipStream_t s = as_stream(stream);
    const int nb = km_blocks(n);
    double* part = static_cast<double*>(ws_d);
    DCV_PP_DISPATCH(kmeanspp_potentials_kernel, dim3(nb), dim3(kKmThreads), 0, s, P_d, n, offset_d, cand_d, (int)trials, closest_d, part)
    DCV_CHECK_LAUNCH();
    hipLaunchKernelGGL(kmeans_final_kernel, dim3((unsigned)trials), dim3(256), 0, s, part, nb, (int)trials, pot_d);
    DCV_CHECK_LAUNCH();
    return DCV_OK;
}

extern "C" int dcv_kmeanspp_update(const double* P_d, int64_t n, int32_t d, const double* offset_d, const double* centre_d,
                                   int32_t first, double* closest_d, double* pot_d, void* ws_d, size_t ws_bytes, void* stream) {
    DCV_REQUIRE(P_d && centre_d && closest_d && pot_d && n > 0, "dcv_kmeanspp_update: bad arguments");
    DCV_REQUIRE(ws_d && ws_bytes >= dcv_kmeanspp_workspace(n, 1), "dcv_kmeanspp_update: workspace too small");
    hipStream_t s = as_stream(stream);
    const int nb = km_blocks(n);
    double* part = static_cast<double*>(ws_d);
    DCV_PP_DISPATCH(kmeanspp_update_kernel, dim3(nb), dim3(kKmThreads), 0, s, P_d, n, offset_d, centre_d, (int)first, closest_d, part)
    DCV_CHECK_LAUNCH();
    hipLaunchKernelGGL(kmeans_final_kernel, dim3(1), dim3(256), 0, s, part, nb, 1, pot_d);
    DCV_CHECK_LAUNCH();
    return DCV_OK;
}
