// FP32 MFMA (v_mfma_f32_32x32x2_f32) block-tile engine for gfx950.
//
// One 256-thread workgroup (4 wave64) owns a TM x TN output tile and walks the contraction
// axis in KB-deep stages: global -> registers -> LDS (double buffered, one barrier per stage),
// LDS -> MFMA fragments.  Two operand storage kinds cover every product on the CV-fit path:
//
//   MMAJOR  element (m, k) at base[row(m)*ld + k]   (contraction index contiguous)
//           LDS image [T][KB] with the 16-byte chunks of a row XOR-swizzled so that the
//           ds_read_b128 of a 32-row fragment is bank-conflict free; one 128-bit read feeds
//           four MFMA k-steps.
//   KMAJOR  element (k, m) at base[row(k)*ld + m]   (contraction index = row index)
//           LDS image [KB][T], a plain copy of the row segments; fragments are conflict-free
//           ds_read_b32 (32 consecutive lanes -> 32 consecutive banks).
//
// The MFMA k-slot assignment is free as long as A and B agree: lane half h = lane>>5 and
// step s = 0..3 of k-group g use k = 8g + 4h + s.
//
//   NT  C[M,N] = A[M,K] . B[N,K]^T      A, B MMAJOR   (forward:  H = X W^T)
//   NN  C[M,N] = A[M,K] . B[K,N]        A MMAJOR, B KMAJOR (dgrad: dX = dZ W)
//   TN  C[M,N] = A[K,M]^T . B[K,N]      A, B KMAJOR   (wgrad / covariance: X^T Y), split-K
//
// Out-of-range rows / columns / k are zero-filled on load and masked on store, so any
// M, N, K works.  VEC (compile time) selects 16-byte global loads; it requires 16-byte aligned
// bases, ld % 4 == 0 and inner extents that are multiples of 4, otherwise the scalar loaders
// are used.
#pragma once
#include <type_traits>
#include "common.h"
#include "handoff.h"

namespace dcv {

#ifdef DCV_STAMP
// diagnostic build only (tools/gemm_bench): per-workgroup cycle stamps of the kernel phases
__device__ unsigned long long g_stamp[8 * 8192];
#define DCV_STAMP_AT(slot)                                                                      \
    do {                                                                                        \
        if (threadIdx.x == 0 && blockIdx.x < 8192 && blockIdx.z == 0)                           \
            g_stamp[blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memtime();                   \
    } while (0)
#define DCV_STAMP_RT(slot)                                                                      \
    do {                                                                                        \
        if (threadIdx.x == 0 && blockIdx.x < 8192 && blockIdx.z == 0)                           \
            g_stamp[blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memrealtime();               \
    } while (0)
#else
#define DCV_STAMP_AT(slot) do {} while (0)
#define DCV_STAMP_RT(slot) do {} while (0)
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// ------------------------------------------------------------------ FP32-accurate products on the BF16 pipe
// x = x1 + x2 + x3 with three bf16 pieces (truncation split: 8 + 8 + 8 mantissa bits, every residual exact
// in fp32), x*y ~ x1y1 + x1y2 + x2y1 + x2y2 + x1y3 + x3y1: each bf16 product is exact in the fp32
// accumulator and the dropped terms (x2y3, x3y2, x3y3) are <= 2^-23 |xy| -- the size of an fp32 product's own
// rounding.  Six v_mfma_f32_32x32x16_bf16 (16 passes each would be 8 fp32-input MFMAs of 64 cycles: 512 cycles;
// the six bf16 ones take 192) per 32x32x16 block; measured error against float64 is at or below that of the
// FP32-input MFMA path (tools/split_bench.hip, tests/test_kernels_gpu.py).
// split3: the three planes of 8 floats, plane p as 4 dwords of 2 bf16 (element 2i in the low half).
__device__ __forceinline__ void split3(const float (&x)[8], u32x4& p1, u32x4& p2, u32x4& p3) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned a = __float_as_uint(x[2 * i]), b = __float_as_uint(x[2 * i + 1]);
        p1[i] = __builtin_amdgcn_perm(b, a, 0x07060302);   // [hi16(b) : hi16(a)]
        const float ra = x[2 * i] - __uint_as_float(a & 0xFFFF0000u);
        const float rb = x[2 * i + 1] - __uint_as_float(b & 0xFFFF0000u);
        const unsigned a2 = __float_as_uint(ra), b2 = __float_as_uint(rb);
        p2[i] = __builtin_amdgcn_perm(b2, a2, 0x07060302);
        const float sa = ra - __uint_as_float(a2 & 0xFFFF0000u);
        const float sb = rb - __uint_as_float(b2 & 0xFFFF0000u);
        p3[i] = __builtin_amdgcn_perm(__float_as_uint(sb), __float_as_uint(sa), 0x07060302);
    }
}
__device__ __forceinline__ bf16x8 as_bf16x8(const u32x4& v) { return __builtin_bit_cast(bf16x8, v); }

enum GemmMode { kNT = 0, kNN = 1, kTN = 2 };

// epilogues that also want the column sums of the (shifted) A operand of a TN product declare `kASum = true`
template <class E, class = void> struct epi_has_asum : std::false_type {};
template <class E> struct epi_has_asum<E, std::void_t<decltype(E::kASum)>> : std::bool_constant<E::kASum> {};

// Logical row -> row of the underlying matrix.  Batch matrices of the MLP engine are never
// materialised: logical row r < half is sample r (x_t), r >= half is sample r-half shifted by
// `lag` rows (x_lag); a sample's row is idx[j] or row0 + j.
struct RowMap {
    const int64_t* idx;
    int64_t row0;
    int32_t half;
    int32_t lag;
    // GATHER is a compile-time property of the kernel instantiation: without it no index load (and
    // no wait on the vector-memory counter) is generated at all
    template <bool GATHER>
    __device__ __forceinline__ int64_t get(int64_t r) const {
        const bool second = half > 0 && r >= half;
        const int64_t j = second ? r - half : r;
        int64_t base;
        if constexpr (GATHER) base = idx ? idx[j] : row0 + j;
        else base = row0 + j;
        return second ? base + lag : base;
    }
};
inline RowMap identity_rows() { return RowMap{nullptr, 0, 0, 0}; }

struct Operand {
    const float* p;
    int64_t ld;
    RowMap rows;
    const float* shift;  // per-(non-contraction)-column value subtracted on load (KMAJOR only), or null
    int vec_ok;          // 16-byte loads legal (alignment part; extents are checked at launch)
    // Pre-split operand (contraction-contiguous kinds only; see "plane operands" below): p points at the three bf16
    // planes of the matrix, ld and pstride are in float units (2 bf16 each): plane q of (row r, contraction index k)
    // is the bf16 at ((const __bf16*)(p + r * ld + q * pstride))[k].
    int planes = 0;
    int64_t pstride = 0;
};

struct GemmDims {
    int64_t M, N, K;
    int64_t k_chunk;  // TN: contraction rows per blockIdx.z (K for the others)
    int tiles_m, tiles_n;
    int xcd_remap;    // XCD-aware block -> tile map enabled (launch-time decision)
    int wt;           // output rows leave as write-through (sc1) stores (launch-time decision: prepare_gemm)
    // Ragged last row tile of a row-parallel product (NT / NN) whose workgroup count just exceeds the CU count: that
    // tile's contraction is cut into tail_split chunks of k_chunk, one extra (short-lived) workgroup each, instead of
    // one full-length workgroup that would share a CU's SIMDs with a regular one for the whole launch.  Every chunk
    // leaves its raw accumulators in tail_ws; the last to arrive (ticket in tail_cnt) adds them up in chunk order and
    // runs the epilogue.  0 = off.  The hand-off between the chunk workgroups is handoff.h's (ordinary device memory).
    int tail_split = 0;
    float* tail_ws = nullptr;
    unsigned* tail_cnt = nullptr;
};

// NBUF: LDS stage buffers of the pure-DMA main loop (a ring: NBUF-1 stages in flight ahead of the
// one being multiplied); the register-staged loop always double-buffers in the first two.
// SPLIT: FP32-accurate products on the BF16 matrix pipe (see split3 / mfma16 below) instead of the
// FP32-input MFMA.
// PL (SPLIT): bit 0 / bit 1 (NT products only) = the A / B operand arrives pre-split into its three bf16 planes
// (Operand::planes), so its share of the in-register split -- the vector-ALU work that bounds the SPLIT flavour --
// disappears from the main loop.  bit 2 / bit 3 = the A / B operand (contraction-contiguous kinds) is an ordinary fp32
// matrix, staged through registers and split ONCE PER WORKGROUP when the stage is stored to LDS, as three plane images:
// the in-register split after the fragment read is repeated by every wave that shares a fragment (two of the four waves
// of a 2 x 2 layout), so this halves the vector-ALU work without the 1.5 x global traffic of pre-split operands --
// what bounds the small-batch products (measured at 8192 x 256 x 512: in-register split 21.9 us, vector-ALU-bound;
// pre-split planes 22.5 us, bound by the L2 -> LDS traffic; without loads 20.7 vs 14.3 us).
template <int WAVES_M_, int WAVES_N_, int FM_, int FN_, int KB_, int NBUF_ = 2, bool SPLIT_ = false, int PL_ = 0>
struct TileCfg {
    static constexpr int WAVES_M = WAVES_M_, WAVES_N = WAVES_N_, FM = FM_, FN = FN_, KB = KB_, NBUF = NBUF_;
    static constexpr bool SPLIT = SPLIT_;
    static constexpr int PL = PL_;
    static_assert(PL == 0 || SPLIT, "plane operands belong to the split flavour");
    // bit 4 (with bits 2 / 3, NT products): the fp32 stage arrives by LDS-DMA in a raw ring behind the plane buffers and is
    // split from THERE, once per workgroup (gemm_block: the LS loop).  Bit-identical, half the vector-ALU work, the global
    // latency hidden as in the ring loop -- and SLOWER (25.7 vs 22.8 us at 8192 x 256 x 512): the plane stores, the second
    // LDS round trip and the per-stage barrier behind the split cost more than the split they save.  Tool-only.
    static constexpr bool LS = (PL_ & 16) != 0;
    static constexpr int RAW_SZ = LS ? (WAVES_M_ * FM_ * 32 + WAVES_N_ * FN_ * 32) * KB_ : 0;   // floats of one raw stage
    // LDS floats of one stage: an fp32 tile is [T][KB] floats, a plane tile three [T][KB/2] images
    static constexpr int A_SZ = (PL & 5) ? 3 * (WAVES_M_ * FM_ * 32) * (KB_ / 2) : (WAVES_M_ * FM_ * 32) * KB_;
    static constexpr int B_SZ = (PL & 10) ? 3 * (WAVES_N_ * FN_ * 32) * (KB_ / 2) : (WAVES_N_ * FN_ * 32) * KB_;
    static_assert(NBUF >= 2 && NBUF <= 8, "NBUF");
    static constexpr int TM = WAVES_M * FM * 32;
    static constexpr int TN = WAVES_N * FN * 32;
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
    static_assert(KB == 16 || KB == 32, "KB");
};

// ------------------------------------------------------------------ LDS image addressing
template <int KB>
__device__ __forceinline__ int mmajor_off(int row, int chunk) {
    constexpr int CPR = KB / 4;    // 16-byte chunks per row
    constexpr int RPBR = 64 / KB;  // rows per 256-byte bank row
    return row * KB + ((chunk ^ ((row / RPBR) % CPR)) << 2);
}

// ------------------------------------------------------------------ stage loaders
// A stage image in LDS is written linearly: unit (t, i) of either operand kind lands at float
// offset (t + 256*i)*4.  For MMAJOR tiles the bank-conflict swizzle therefore lives on the SOURCE
// side (each thread fetches the logical 16-byte chunk that belongs in its linear slot) and on the
// fragment read.  That makes the image fillable two ways:
//   * LDS-DMA (global_load_lds_dwordx4, "glds"): when every unit of the stage is in range the
//     64 lanes of a wave write 1 KiB contiguously, no VGPR round trip, no ds_write;
//   * through registers (ragged edges, scalar loads, covariance shift): loads carry no arithmetic
//     so they all stay in flight across the MFMA phase; the shift is applied at LDS-store time.
// Issued through inline asm on purpose: hipcc treats the builtin form as an LDS write it must
// order against later ds_reads and waits vmcnt(0) right after issuing it, which serialises the
// DMA with the MFMA phase.  The asm form is outside the compiler's vmcnt bookkeeping; completion
// is awaited explicitly (asm s_waitcnt vmcnt(N), then the workgroup barrier) before the image is read.
// M0 carries the wave-uniform LDS byte address; it is compiler-reserved, so it is saved / restored
// inside the same statement (cdna_hip_programming.md section 5.7).
__device__ __forceinline__ void glds16(const float* gsrc, unsigned lds_wave_addr) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_wave_addr)
        : "memory");
}
// LDS byte address of a pointer into the dynamic shared array, made provably wave-uniform
__device__ __forceinline__ unsigned lds_addr_uniform(const float* p) {
    const unsigned a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const float*)p;
    return __builtin_amdgcn_readfirstlane(a);
}

// Batched form for tiles whose rows are an affine function of the row index (no gather): wave-uniform
// 64-bit base in SGPRs + one constant 32-bit byte offset per unit in a VGPR, so a stage costs no
// vector-ALU instruction at all (the issue slots of a SIMD are shared with the co-resident workgroup's
// MFMA stream, and address arithmetic there was measured at ~2500 cycles per stage).  lds0 is the
// wave's LDS byte address for unit 0; unit i lands 4096 bytes further.  The leading s_nop covers a
// base that the compiler produced with v_readfirstlane (5 wait states before a VMEM read of it).
template <int PER>
__device__ __forceinline__ void glds16_batch(const float* base, const unsigned (&voff)[PER], unsigned lds0) {
    static_assert(PER == 1 || PER == 2 || PER == 4, "units per thread");
    unsigned keep;
    if constexpr (PER == 1) {
        asm volatile(
            "s_mov_b32 %0, m0\n\ts_nop 3\n\t"
            "s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(voff[0]), "s"(base), "s"(lds0)
            : "memory");
    } else if constexpr (PER == 2) {
        asm volatile(
            "s_mov_b32 %0, m0\n\ts_nop 3\n\t"
            "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\t"
            "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(voff[0]), "v"(voff[1]), "s"(base), "s"(lds0), "s"(lds0 + 4096u)
            : "memory");
    } else {
        asm volatile(
            "s_mov_b32 %0, m0\n\ts_nop 3\n\t"
            "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %5\n\t"
            "s_mov_b32 m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %5\n\t"
            "s_mov_b32 m0, %8\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %5\n\t"
            "s_mov_b32 m0, %9\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %5\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "v"(voff[3]), "s"(base), "s"(lds0), "s"(lds0 + 4096u),
              "s"(lds0 + 8192u), "s"(lds0 + 12288u)
            : "memory");
    }
}
// largest leading dimension the 32-bit unit offsets of the batched form cover (128 rows * ld * 4 bytes < 2^32)
constexpr int64_t kMaxAffineLd = (int64_t)1 << 22;

// MMAJOR tile [T][KB]: unit u = t + 256*i -> row = u / CPR, slot = u % CPR.  A thread's rows are
// the same in every stage, so their base pointers are resolved once.
template <int T, int KB, bool VEC, bool GATHER>
struct MMajorStage {
    static constexpr int CPR = KB / 4;
    static constexpr int RPBR = 64 / KB;
    static constexpr int UNITS = T * CPR;
    static constexpr int PER = (UNITS + 255) / 256;
    static_assert(256 % CPR == 0 && ((256 / CPR) / RPBR) % CPR == 0, "swizzle must not depend on the unit index");
    float4 r[PER];
    const float* src[PER];  // &A[row][logical chunk * 4] or null when the row is out of range
    unsigned voff[PER];     // byte offset of unit i from the tile's first row (affine tiles)
    int kofs;               // logical chunk * 4
    bool all_rows;          // every row of the tile is in range (workgroup-uniform)
    bool affine;            // the tile's rows are row0 + m: no gather, not astride the x_t / x_lag seam

    // CLAMP (kernels with plane operands: pure LDS-DMA, no register-staged loop): rows past the end read the last
    // valid row instead of being zero-filled -- they only feed output rows the epilogue masks
    template <bool CLAMP = false>
    __device__ __forceinline__ void init(const Operand& op, int64_t m0, int64_t m_end, int t) {
        const int row0 = t / CPR;
        kofs = ((t % CPR) ^ ((row0 / RPBR) % CPR)) * 4;
        all_rows = (UNITS % 256 == 0) && (m0 + T <= m_end);
        affine = !GATHER && op.ld < kMaxAffineLd && (op.rows.half <= 0 || m0 >= op.rows.half || m0 + T <= op.rows.half);
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int u = t + 256 * i;
            int64_t m = m0 + u / CPR;
            if constexpr (CLAMP) m = m < m_end ? m : m_end - 1;
            const bool ok = (UNITS % 256 == 0 || u < UNITS) && m < m_end;
            src[i] = ok ? op.p + op.rows.template get<GATHER>(m) * op.ld + kofs : nullptr;
            voff[i] = (unsigned)(((int64_t)(u / CPR) * op.ld + kofs) * 4);
        }
    }
    // wave-uniform base of the affine form: first row of the tile, contraction offset k0
    __device__ __forceinline__ const float* tile_base(const Operand& op, int64_t m0, int64_t k0) const {
        return op.p + op.rows.template get<false>(m0) * op.ld + k0;
    }
    __device__ __forceinline__ void glds_affine(const float* base, unsigned lds0) const {
        // a tile shape with another unit count would silently issue NO copy here (a 256-row tile: 8 units per thread)
        static_assert(UNITS % 256 != 0 || PER == 1 || PER == 2 || PER == 4, "units per thread of the batched LDS-DMA form");
        if constexpr (UNITS % 256 == 0 && (PER == 1 || PER == 2 || PER == 4)) glds16_batch<PER>(base, voff, lds0);
    }
    __device__ __forceinline__ bool dense(int64_t k0, int64_t k_end) const { return VEC && all_rows && k0 + KB <= k_end; }
    __device__ __forceinline__ void glds(int64_t k0, float* lds, int t) const {
#pragma unroll
        for (int i = 0; i < PER; ++i) glds16(src[i] + k0, lds_addr_uniform(lds + ((t & ~63) + 256 * i) * 4));
    }
    __device__ __forceinline__ void load(int64_t k0, int64_t k_end) {
        if (dense(k0, k_end)) {  // uniform: unconditional 16-byte loads
#pragma unroll
            for (int i = 0; i < PER; ++i) r[i] = *reinterpret_cast<const float4*>(src[i] + k0);
            return;
        }
        const int64_t k = k0 + kofs;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (src[i] != nullptr && k < k_end) {
                const float* s = src[i] + k0;
                if constexpr (VEC) {
                    v = *reinterpret_cast<const float4*>(s);
                } else {
                    v.x = s[0];
                    if (k + 1 < k_end) v.y = s[1];
                    if (k + 2 < k_end) v.z = s[2];
                    if (k + 3 < k_end) v.w = s[3];
                }
            }
            r[i] = v;
        }
    }
    __device__ __forceinline__ void store(float* lds, int t) const {
#pragma unroll
        for (int i = 0; i < PER; ++i)
            if (UNITS % 256 == 0 || t + 256 * i < UNITS) *reinterpret_cast<float4*>(lds + (t + 256 * i) * 4) = r[i];
    }
    // The staged fp32 values split into their three bf16 pieces on the way to LDS (TileCfg::PL bits 2 / 3): plane q of the
    // tile is an MMAJOR image [T][KB / 2] floats (= [T][KB] bf16) at lds + q * T * KB / 2, its 16-byte chunks (8
    // consecutive k) swizzled exactly as the pre-split operands' images are, so read_frags8 reads either alike.  A
    // thread's 4 floats are one half of such a chunk: 8 bytes per plane.
    __device__ __forceinline__ void store_planes(float* lds, int t) const {
        constexpr int KF = KB / 2, CPRP = KF / 4, RPBRP = 64 / KF;
        const int c4 = kofs >> 2;   // this thread's logical 4-float chunk of its rows (the same in every unit)
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int u = t + 256 * i;
            if (UNITS % 256 == 0 || u < UNITS) {
                const int row = u / CPR;
                const int off = row * KF + (((c4 >> 1) ^ ((row / RPBRP) % CPRP)) << 2) + (c4 & 1) * 2;
                const unsigned a = __float_as_uint(r[i].x), b = __float_as_uint(r[i].y), c = __float_as_uint(r[i].z), d = __float_as_uint(r[i].w);
                uint2 p1, p2, p3;
                p1.x = __builtin_amdgcn_perm(b, a, 0x07060302);
                p1.y = __builtin_amdgcn_perm(d, c, 0x07060302);
                const float ra = r[i].x - __uint_as_float(a & 0xFFFF0000u), rb = r[i].y - __uint_as_float(b & 0xFFFF0000u);
                const float rc = r[i].z - __uint_as_float(c & 0xFFFF0000u), rd = r[i].w - __uint_as_float(d & 0xFFFF0000u);
                const unsigned a2 = __float_as_uint(ra), b2 = __float_as_uint(rb), c2 = __float_as_uint(rc), d2 = __float_as_uint(rd);
                p2.x = __builtin_amdgcn_perm(b2, a2, 0x07060302);
                p2.y = __builtin_amdgcn_perm(d2, c2, 0x07060302);
                const float sa = ra - __uint_as_float(a2 & 0xFFFF0000u), sb = rb - __uint_as_float(b2 & 0xFFFF0000u);
                const float sc = rc - __uint_as_float(c2 & 0xFFFF0000u), sd = rd - __uint_as_float(d2 & 0xFFFF0000u);
                p3.x = __builtin_amdgcn_perm(__float_as_uint(sb), __float_as_uint(sa), 0x07060302);
                p3.y = __builtin_amdgcn_perm(__float_as_uint(sd), __float_as_uint(sc), 0x07060302);
                *reinterpret_cast<uint2*>(lds + off) = p1;
                *reinterpret_cast<uint2*>(lds + T * KF + off) = p2;
                *reinterpret_cast<uint2*>(lds + 2 * T * KF + off) = p3;
            }
        }
    }
};

// KMAJOR tile [KB][T]: unit u -> krow = u / (T/4), c4 = u % (T/4).  A thread's columns are the
// same in every stage (and in every unit, since 256 % (T/4) == 0).  The matrix rows of a stage
// are resolved one stage ahead (resolve()), so a gathering row map costs no exposed latency.
template <int T, int KB, bool VEC, bool GATHER>
struct KMajorStage {
    static constexpr int C4 = T / 4;
    static constexpr int UNITS = KB * C4;
    static constexpr int PER = (UNITS + 255) / 256;
    static_assert(256 % C4 == 0, "column group must be unit independent");
    float4 r[PER];
    float4 sh;          // shift of this thread's 4 columns
    const float* base;  // op.p + first column of this thread
    int64_t ld;
    int64_t rowv[PER];  // matrix row of unit i in the NEXT load (or -1)
    unsigned okmask;    // units of the LAST register load that were in range
    int ncol;           // how many of the 4 columns are in range (0..4)
    bool has_shift;
    bool all_cols;      // every column of the tile in range, 16-byte loads (workgroup-uniform)
    bool affine;        // no gather and a leading dimension the 32-bit unit offsets cover
    unsigned voff[PER]; // byte offset of unit i from (first row of the stage, first column of the tile)

    __device__ __forceinline__ void init(const Operand& op, int64_t c0, int64_t c_end, int t) {
        const int64_t col = c0 + (t % C4) * 4;
        const int64_t left = c_end - col;
        ncol = left >= 4 ? 4 : (left > 0 ? (int)left : 0);
        base = op.p + col;
        ld = op.ld;
        has_shift = op.shift != nullptr;
        sh = make_float4(0.f, 0.f, 0.f, 0.f);
        if (has_shift) {
            if (ncol > 0) sh.x = op.shift[col];
            if (ncol > 1) sh.y = op.shift[col + 1];
            if (ncol > 2) sh.z = op.shift[col + 2];
            if (ncol > 3) sh.w = op.shift[col + 3];
        }
        all_cols = VEC && (UNITS % 256 == 0) && (c0 + T <= c_end);   // a shift is subtracted at store time (register path) or on the fragments (DMA path)
        affine = !GATHER && op.ld < kMaxAffineLd;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int u = t + 256 * i;
            voff[i] = (unsigned)(((int64_t)(u / C4) * op.ld + (u % C4) * 4) * 4);
        }
        okmask = 0;
    }
    // affine form: the KB rows of the stage at k0 must be consecutive matrix rows
    __device__ __forceinline__ static bool stage_affine(const Operand& op, int64_t k0) {
        return op.rows.half <= 0 || k0 >= op.rows.half || k0 + KB <= op.rows.half;
    }
    __device__ __forceinline__ static const float* stage_base(const Operand& op, int64_t k0, int64_t row_off, int64_t c0) {
        return op.p + (op.rows.template get<false>(k0) + row_off) * op.ld + c0;
    }
    __device__ __forceinline__ void glds_affine(const float* base, unsigned lds0) const {
        // a tile shape with another unit count would silently issue NO copy here (a 256-row tile: 8 units per thread)
        static_assert(UNITS % 256 != 0 || PER == 1 || PER == 2 || PER == 4, "units per thread of the batched LDS-DMA form");
        if constexpr (UNITS % 256 == 0 && (PER == 1 || PER == 2 || PER == 4)) glds16_batch<PER>(base, voff, lds0);
    }
    __device__ __forceinline__ void resolve(const Operand& op, int64_t k0, int64_t k_end, int64_t row_off, int t) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int u = t + 256 * i;
            const int64_t k = k0 + u / C4;
            const bool ok = (UNITS % 256 == 0 || u < UNITS) && k < k_end && ncol > 0;
            rowv[i] = ok ? op.rows.template get<GATHER>(k) + row_off : -1;
        }
    }
    __device__ __forceinline__ bool dense(int64_t k0, int64_t k_end) const { return all_cols && k0 + KB <= k_end; }
    // both loaders fetch the stage resolved by the previous resolve() call
    __device__ __forceinline__ void glds(float* lds, int t) const {
#pragma unroll
        for (int i = 0; i < PER; ++i) glds16(base + rowv[i] * ld, lds_addr_uniform(lds + ((t & ~63) + 256 * i) * 4));
    }
    __device__ __forceinline__ void load(int64_t k0, int64_t k_end) {
        if (dense(k0, k_end)) {  // uniform: unconditional 16-byte loads
#pragma unroll
            for (int i = 0; i < PER; ++i) r[i] = *reinterpret_cast<const float4*>(base + rowv[i] * ld);
            okmask = ~0u;
            return;
        }
        unsigned m = 0;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (rowv[i] >= 0) {
                const float* s = base + rowv[i] * ld;
                m |= 1u << i;
                if constexpr (VEC) {
                    v = *reinterpret_cast<const float4*>(s);
                } else {
                    v.x = s[0];
                    if (ncol > 1) v.y = s[1];
                    if (ncol > 2) v.z = s[2];
                    if (ncol > 3) v.w = s[3];
                }
            }
            r[i] = v;
        }
        okmask = m;
    }
    __device__ __forceinline__ void store(float* lds, int t) const {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            if (UNITS % 256 == 0 || t + 256 * i < UNITS) {
                float4 v = r[i];
                if (has_shift && ((okmask >> i) & 1u)) {  // zero-filled rows stay zero
                    v.x -= sh.x; v.y -= sh.y; v.z -= sh.z; v.w -= sh.w;
                }
                *reinterpret_cast<float4*>(lds + (t + 256 * i) * 4) = v;
            }
        }
    }
};

// ------------------------------------------------------------------ fragments
// MMAJOR fragment of the 32 rows starting at `rb`, k-group g: 4 k values per lane.
template <int KB>
__device__ __forceinline__ v4f frag_mmajor(const float* lds, int rb, int g, int lane) {
    const int row = rb + (lane & 31);
    const int chunk = 2 * g + (lane >> 5);
    return *reinterpret_cast<const v4f*>(lds + mmajor_off<KB>(row, chunk));
}
// MMAJOR: 16-byte chunk `chunk` (4 consecutive k) of row rb + (lane & 31)
template <int KB>
__device__ __forceinline__ v4f frag_mmajor_chunk(const float* lds, int rb, int chunk, int lane) {
    return *reinterpret_cast<const v4f*>(lds + mmajor_off<KB>(rb + (lane & 31), chunk));
}
// KMAJOR fragment of the 32 columns starting at `cb`, k-group g, step s.
template <int T>
__device__ __forceinline__ float frag_kmajor(const float* lds, int cb, int g, int s, int lane) {
    const int k = 8 * g + 4 * (lane >> 5) + s;
    return lds[k * T + cb + (lane & 31)];
}

__device__ __forceinline__ float f4_get(const float4& v, int s) {
    return s == 0 ? v.x : (s == 1 ? v.y : (s == 2 ? v.z : v.w));
}

// row of accumulator register `reg` inside a 32x32 MFMA tile
__device__ __forceinline__ int acc_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// ------------------------------------------------------------------ 4-column row segments
__device__ __forceinline__ void store_quad(float* p, const float4& v, int nvalid, bool vec) {
    if (vec && nvalid == 4) {
        *reinterpret_cast<float4*>(p) = v;
    } else {
        p[0] = v.x;
        if (nvalid > 1) p[1] = v.y;
        if (nvalid > 2) p[2] = v.z;
        if (nvalid > 3) p[3] = v.w;
    }
}
__device__ __forceinline__ float4 load_quad(const float* p, int nvalid, bool vec) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (vec && nvalid == 4) {
        v = *reinterpret_cast<const float4*>(p);
    } else {
        v.x = p[0];
        if (nvalid > 1) v.y = p[1];
        if (nvalid > 2) v.z = p[2];
        if (nvalid > 3) v.w = p[3];
    }
    return v;
}

// Sign mask of an activation tile, in the thread layout of the staged epilogue: for row segment q of a tile
// (the thread's q-th 4-column piece) wave w stores four 64-bit ballots, one per column of the piece, bit = lane:
//   word (((tile * NQT + q) * 4 + w) * 4 + c).
// The dgrad of a ReLU-family layer needs only sign(H) (1 bit per element instead of 32), and the dgrad epilogue
// of the same tile shape reads its bits back with the same indexing.
template <int N>
__device__ __forceinline__ void store_sign_mask(unsigned long long* mask, const float4 (&v)[N], int64_t seg0, int wave, int lane) {
#pragma unroll
    for (int q = 0; q < N; ++q) {
        const unsigned long long bx = __ballot(v[q].x > 0.f), by = __ballot(v[q].y > 0.f);
        const unsigned long long bz = __ballot(v[q].z > 0.f), bw = __ballot(v[q].w > 0.f);
        if (lane < 4) mask[((seg0 + q) * 4 + wave) * 4 + lane] = lane == 0 ? bx : (lane == 1 ? by : (lane == 2 ? bz : bw));
    }
}
// s_waitcnt vmcnt(N) through asm: the LDS-DMA instructions it counts are asm too (see glds16)
template <int N>
__device__ __forceinline__ void vm_wait() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// wait until at most min(MAXY, younger) stages of GL instructions each (plus EXTRA younger loads) are still in flight
template <int MAXY, int GL, int EXTRA = 0>
__device__ __forceinline__ void vm_wait_younger(int64_t younger) {
    if constexpr (MAXY == 0) vm_wait<EXTRA>();
    else {
        if (younger >= MAXY) vm_wait<MAXY * GL + EXTRA>();
        else vm_wait_younger<MAXY - 1, GL, EXTRA>(younger);
    }
}

// ------------------------------------------------------------------ the kernel body
// NB: number of B operands sharing A (2 for the lagged covariance: B and B shifted by `lag2`).
template <int MODE, class Cfg, int NB, bool VEC, bool GATHER, class Epi>
__device__ __forceinline__ void gemm_block(const Operand& A, const Operand& B, int64_t lag2, const GemmDims& d,
                                           int tile_m, int tile_n, int64_t k_begin, int64_t k_end, float* lds,
                                           Epi& epi, int tail_chunk = -1) {
    constexpr int TM = Cfg::TM, TN = Cfg::TN, KB = Cfg::KB, FM = Cfg::FM, FN = Cfg::FN;
    constexpr int A_SZ = Cfg::A_SZ, B_SZ = Cfg::B_SZ;
    constexpr int STAGE = A_SZ + NB * B_SZ;
    constexpr bool A_MM = (MODE == kNT || MODE == kNN);
    constexpr bool B_MM = (MODE == kNT);
    // ---- plane operands (Cfg::PL): the operand was split into its three bf16 planes ahead of time (a matrix that is
    // constant over many products: the training set, the weights between optimiser steps).  A plane tile is three
    // MMAJOR images of [T][KB/2] floats (= [T][KB] bf16) with the same chunk swizzle, filled by LDS-DMA only; a lane's
    // 16-byte fragment read IS the bf16x8 operand of v_mfma_f32_32x32x16_bf16, bit-identical to what split3 would
    // have produced from the fp32 value.  Such kernels have no register-staged loop: the launch guarantees K % KB == 0
    // and 16-byte loads, and ragged row tiles read clamped rows (MMajorStage::init<true>).
    constexpr int PL = Cfg::PL;
    constexpr bool PRE = (PL & 3) != 0;                      // an operand arrives pre-split (pure LDS-DMA kernel)
    constexpr bool PSA = (PL & 4) != 0, PSB = (PL & 8) != 0;  // fp32 operand split when its stage is stored to LDS (register-staged kernel)
    constexpr bool PA = (PL & 5) != 0, PB = (PL & 10) != 0;   // the operand's LDS image is three plane images
    constexpr bool LS = Cfg::LS;                               // ... split from a DMA-filled raw fp32 ring in LDS (below)
    static_assert(!LS || (MODE == kNT && PSA && PSB && VEC && !GATHER && NB == 1), "split-from-LDS: NT products, both operands, 16-byte rows");
    static_assert(!PRE || (MODE == kNT && NB == 1 && VEC && Cfg::SPLIT), "plane operands: NT products of the split flavour");
    static_assert(!(PSA || PSB) || (!PRE && NB == 1 && Cfg::SPLIT), "split-at-store: split flavour, one B operand, no pre-split operand beside it");
    static_assert(!PSA || MODE == kNT || MODE == kNN, "split-at-store of A needs a contraction-contiguous A");
    static_assert(!PSB || MODE == kNT, "split-at-store of B needs a contraction-contiguous B");
    constexpr int KF = KB / 2;           // floats per row of a plane image
    constexpr int KA = (PL & 1) ? KF : KB, KBB = (PL & 2) ? KF : KB;   // row length of the stage objects' GLOBAL side
    constexpr int NPA = (PL & 1) ? 3 : 1, NPB = (PL & 2) ? 3 : 1;
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int wm = (wave / Cfg::WAVES_N) * FM * 32;
    const int wn = (wave % Cfg::WAVES_N) * FN * 32;
    const int64_t m0 = (int64_t)tile_m * TM, n0 = (int64_t)tile_n * TN;
#ifdef DCV_ABL_SAMETILE   // diagnostic (wrong results): every workgroup streams the first few row tiles of A -- all cache hits
    const int64_t m0_ld = (int64_t)(tile_m % DCV_ABL_SAMETILE) * TM;
#else
    const int64_t m0_ld = m0;
#endif

    f32x16 acc[NB][FM][FN];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[b][i][j][e] = 0.f;

    MMajorStage<TM, KA, VEC, GATHER> am;
    KMajorStage<TM, KB, VEC, GATHER> ak;
    MMajorStage<TN, KBB, VEC, GATHER> bm;
    KMajorStage<TN, KB, VEC, GATHER> bk[NB];
    if constexpr (A_MM) am.template init<PRE || LS>(A, m0, d.M, t);
    else ak.init(A, m0, d.M, t);
    if constexpr (B_MM) bm.template init<PRE || LS>(B, n0, d.N, t);
    else {
#pragma unroll
        for (int b = 0; b < NB; ++b) bk[b].init(B, n0, d.N, t);
    }

    auto resolve_stage = [&](int64_t k0) {
        if constexpr (!A_MM) ak.resolve(A, k0, k_end, 0, t);
        if constexpr (!B_MM) {
            bk[0].resolve(B, k0, k_end, 0, t);
            if constexpr (NB == 2) bk[1].resolve(B, k0, k_end, lag2, t);
        }
    };
    auto stage_dense = [&](int64_t k0) -> bool {  // workgroup-uniform
#ifdef DCV_NO_GLDS
        return false;
#endif
        bool ok;
        if constexpr (A_MM) ok = am.dense(k0, k_end);
        else ok = ak.dense(k0, k_end);
        if constexpr (B_MM) ok = ok && bm.dense(k0, k_end);
        else {
            ok = ok && bk[0].dense(k0, k_end);
            if constexpr (NB == 2) ok = ok && bk[1].dense(k0, k_end);
        }
        return ok;
    };
    // pure-DMA issue of the stage at k0 into ring buffer `buf`
    const unsigned ldsw = lds_addr_uniform(lds + (t & ~63) * 4);   // this wave's unit-0 slot of buffer 0
    // plane kernels: an operand whose tile is in range and affine takes the batched form, any other (gathered rows,
    // the x_t / x_lag seam, a ragged last row tile) per-thread pointers to clamped rows; workgroup-uniform choice
    const bool pl_a_aff = am.affine && am.all_rows, pl_b_aff = bm.affine && bm.all_rows;
    auto glds_stage = [&](int64_t k0, int buf, int part = 3) {   // part: 1 = A operand, 2 = B operand(s)
        if constexpr (PRE) {
            const unsigned l0 = ldsw + (unsigned)(buf * STAGE) * 4u;
            float* b = lds + buf * STAGE;
            if (part & 1) {
                const int64_t ka = (PL & 1) ? k0 / 2 : k0;   // float offset inside a row
#pragma unroll
                for (int q = 0; q < NPA; ++q) {
                    if (pl_a_aff) am.glds_affine(am.tile_base(A, m0_ld, ka) + q * A.pstride, l0 + (unsigned)(q * TM * KF) * 4u);
                    else am.glds(ka + q * A.pstride, b + q * TM * KF, t);
                }
            }
            if (part & 2) {
                const int64_t kb = (PL & 2) ? k0 / 2 : k0;
#pragma unroll
                for (int q = 0; q < NPB; ++q) {
                    if (pl_b_aff) bm.glds_affine(bm.tile_base(B, n0, kb) + q * B.pstride, l0 + (unsigned)(A_SZ + q * TN * KF) * 4u);
                    else bm.glds(kb + q * B.pstride, b + A_SZ + q * TN * KF, t);
                }
            }
        } else if constexpr (GATHER) {   // per-thread 64-bit source pointers
            float* b = lds + buf * STAGE;
            if (part & 1) {
                resolve_stage(k0);
                if constexpr (A_MM) am.glds(k0, b, t);
                else ak.glds(b, t);
            }
            if (part & 2) {
                if constexpr (B_MM) bm.glds(k0, b + A_SZ, t);
                else {
                    bk[0].glds(b + A_SZ, t);
                    if constexpr (NB == 2) bk[1].glds(b + A_SZ + B_SZ, t);
                }
            }
        } else {                  // affine tiles: uniform base + constant unit offsets, no vector ALU
            const unsigned l0 = ldsw + (unsigned)(buf * STAGE) * 4u;
            if (part & 1) {
                if constexpr (A_MM) am.glds_affine(am.tile_base(A, m0_ld, k0), l0);
                else ak.glds_affine(ak.stage_base(A, k0, 0, m0), l0);
            }
            if (part & 2) {
                if constexpr (B_MM) bm.glds_affine(bm.tile_base(B, n0, k0), l0 + A_SZ * 4u);
                else {
                    bk[0].glds_affine(bk[0].stage_base(B, k0, 0, n0), l0 + A_SZ * 4u);
                    if constexpr (NB == 2) bk[1].glds_affine(bk[1].stage_base(B, k0, lag2, n0), l0 + (A_SZ + B_SZ) * 4u);
                }
            }
        }
    };
    // workgroup-uniform: every stage of [k_begin, k_end) can take the affine form
    auto affine_all = [&]() -> bool {
        if constexpr (GATHER) return true;
        auto seam_ok = [&](const Operand& op) {   // KMAJOR: the x_t / x_lag seam falls on a stage boundary
            const int64_t h = op.rows.half;
            return h <= 0 || k_begin >= h || k_end <= h || ((h - k_begin) % KB == 0);
        };
        bool ok;
        if constexpr (A_MM) ok = am.affine;
        else ok = ak.affine && seam_ok(A);
        if constexpr (B_MM) ok = ok && bm.affine;
        else ok = ok && bk[0].affine && seam_ok(B);
        return ok;
    };
    auto load_stage = [&](int64_t k0) {
        if constexpr (A_MM) am.load(k0, k_end);
        else ak.load(k0, k_end);
        if constexpr (B_MM) bm.load(k0, k_end);
        else {
            bk[0].load(k0, k_end);
            if constexpr (NB == 2) bk[1].load(k0, k_end);
        }
    };
    auto store_stage = [&](float* buf) {
        if constexpr (A_MM) {
            if constexpr (PSA) am.store_planes(buf, t);
            else am.store(buf, t);
        } else ak.store(buf, t);
        if constexpr (B_MM) {
            if constexpr (PSB) bm.store_planes(buf + A_SZ, t);
            else bm.store(buf + A_SZ, t);
        }
        else {
            bk[0].store(buf + A_SZ, t);
            if constexpr (NB == 2) bk[1].store(buf + A_SZ + B_SZ, t);
        }
    };
    // fragments of one k-group (4 MFMA steps): registers av[i][s], bv[b][j][s]
    struct Frags {
        v4f a[FM];
        v4f b[NB][FN];
    };
    // Column shift of KMAJOR operands (covariances: z = x - shift).  The register-staged loop subtracts it when a stage
    // is stored to LDS; a stage filled by LDS-DMA holds the raw values, and the shift -- one constant per lane and
    // fragment, since a lane always reads the same column -- comes off the fragments instead (`sub`): the same float32
    // subtraction either way, one v_sub per fragment element against 64-cycle MFMAs.
    float sha[FM], shb[FN];
    bool dma_sub = false;
#pragma unroll
    for (int i = 0; i < FM; ++i) sha[i] = 0.f;
#pragma unroll
    for (int j = 0; j < FN; ++j) shb[j] = 0.f;
    if constexpr (!A_MM) {
        if (A.shift) {
            dma_sub = true;
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int64_t c = m0 + wm + i * 32 + (lane & 31);
                sha[i] = c < d.M ? A.shift[c] : 0.f;
            }
        }
    }
    if constexpr (!B_MM) {
        if (B.shift) {
            dma_sub = true;
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int64_t c = n0 + wn + j * 32 + (lane & 31);
                shb[j] = c < d.N ? B.shift[c] : 0.f;
            }
        }
    }
    auto read_frags = [&](Frags& f, const float* la, const float* lb, int g) {
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            if constexpr (A_MM) {
                f.a[i] = frag_mmajor<KB>(la, wm + i * 32, g, lane);
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s) f.a[i][s] = frag_kmajor<TM>(la, wm + i * 32, g, s, lane);
            }
        }
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                if constexpr (B_MM) {
                    f.b[b][j] = frag_mmajor<KB>(lb, wn + j * 32, g, lane);
                } else {
#pragma unroll
                    for (int s = 0; s < 4; ++s) f.b[b][j][s] = frag_kmajor<TN>(lb + b * B_SZ, wn + j * 32, g, s, lane);
                }
            }
    };
    // The column shift of a DMA-filled stage comes off the fragments right before their MFMA group -- one group after
    // they were requested, so the subtraction never waits on LDS (applied inside read_frags it stalled every group on
    // the reads it had just issued: covariance 110 -> 130 TFLOP/s at 5M x 256 together with two waves per SIMD).
    auto sub_frags = [&](Frags& f) {
        if constexpr (!A_MM) {
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int s = 0; s < 4; ++s) f.a[i][s] -= sha[i];
        }
        if constexpr (!B_MM) {
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int j = 0; j < FN; ++j)
#pragma unroll
                    for (int s = 0; s < 4; ++s) f.b[b][j][s] -= shb[j];
        }
    };
    // Makes the 128-bit fragment registers opaque right before their MFMA group: the ds_read_b128 that
    // produced them must stay whole (hipcc otherwise scalarises a vector load whose lanes are consumed
    // one MFMA step at a time once the loop is rotated across the stage boundary).
    auto pin_frags = [&](Frags& f) {
        if constexpr (A_MM) {
#pragma unroll
            for (int i = 0; i < FM; ++i) asm volatile("" : "+v"(f.a[i]));
        }
        if constexpr (B_MM) {
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int j = 0; j < FN; ++j) asm volatile("" : "+v"(f.b[b][j]));
        }
    };
    // ---- SPLIT path: one 16-deep step.  Lane l holds, per 32-row operand tile, the 8 contraction values
    // k = 16 s + 8 (l >> 5) + e (e = 0..7) of row / column l & 31 -- the operand layout of
    // v_mfma_f32_32x32x16_bf16 -- as fp32; they are split into three bf16 planes in registers.
    struct Frags8 {
        float a[FM][8];
        float b[NB][FN][8];
        u32x4 ap[3][FM];       // plane operands: the fragment is the bf16x8 itself
        u32x4 bp[3][FN];
    };
    auto read_frags8 = [&](Frags8& f, const float* la, const float* lb, int s) {
        const int h = lane >> 5;
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            if constexpr (PA) {
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    f.ap[q][i] = __builtin_bit_cast(u32x4, frag_mmajor_chunk<KF>(la + q * TM * KF, wm + i * 32, 2 * s + h, lane));
            } else if constexpr (A_MM) {
                const v4f u = frag_mmajor_chunk<KB>(la, wm + i * 32, 4 * s + 2 * h, lane);
                const v4f v = frag_mmajor_chunk<KB>(la, wm + i * 32, 4 * s + 2 * h + 1, lane);
                f.a[i][0] = u.x; f.a[i][1] = u.y; f.a[i][2] = u.z; f.a[i][3] = u.w;
                f.a[i][4] = v.x; f.a[i][5] = v.y; f.a[i][6] = v.z; f.a[i][7] = v.w;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) f.a[i][e] = la[(16 * s + 8 * h + e) * TM + wm + i * 32 + (lane & 31)];
            }
        }
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                if constexpr (PB) {
#pragma unroll
                    for (int q = 0; q < 3; ++q)
                        f.bp[q][j] = __builtin_bit_cast(u32x4, frag_mmajor_chunk<KF>(lb + q * TN * KF, wn + j * 32, 2 * s + h, lane));
                } else if constexpr (B_MM) {
                    const v4f u = frag_mmajor_chunk<KB>(lb, wn + j * 32, 4 * s + 2 * h, lane);
                    const v4f v = frag_mmajor_chunk<KB>(lb, wn + j * 32, 4 * s + 2 * h + 1, lane);
                    f.b[b][j][0] = u.x; f.b[b][j][1] = u.y; f.b[b][j][2] = u.z; f.b[b][j][3] = u.w;
                    f.b[b][j][4] = v.x; f.b[b][j][5] = v.y; f.b[b][j][6] = v.z; f.b[b][j][7] = v.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) f.b[b][j][e] = lb[b * B_SZ + (16 * s + 8 * h + e) * TN + wn + j * 32 + (lane & 31)];
                }
            }
    };
    struct Planes {   // the three bf16 pieces of a Frags8
        u32x4 a1[FM], a2[FM], a3[FM], b1[NB][FN], b2[NB][FN], b3[NB][FN];
    };
    auto split_frags = [&](const Frags8& f, Planes& p) {
#pragma unroll
        for (int i = 0; i < FM; ++i) {
#ifdef DCV_ABL_NOSPLIT_A   // diagnostic: what a pre-split A operand would save (wrong results)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                p.a1[i][e] = __builtin_amdgcn_perm(__float_as_uint(f.a[i][2 * e + 1]), __float_as_uint(f.a[i][2 * e]), 0x07060302);
                p.a2[i][e] = p.a1[i][e];
                p.a3[i][e] = p.a1[i][e];
            }
#else
            if constexpr (PA) {
                p.a1[i] = f.ap[0][i]; p.a2[i] = f.ap[1][i]; p.a3[i] = f.ap[2][i];
            } else {
                split3(f.a[i], p.a1[i], p.a2[i], p.a3[i]);
            }
#endif
        }
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                if constexpr (PB) {
                    p.b1[b][j] = f.bp[0][j]; p.b2[b][j] = f.bp[1][j]; p.b3[b][j] = f.bp[2][j];
                } else {
                    split3(f.b[b][j], p.b1[b][j], p.b2[b][j], p.b3[b][j]);
                }
            }
    };
    // Makes the planes opaque at this point of the program: their split must have been computed by here (the compiler
    // otherwise sinks it to the first use, on the far side of a stage barrier).
    [[maybe_unused]] auto pin_planes = [&](Planes& p) {
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            asm volatile("" : "+v"(p.a1[i]), "+v"(p.a2[i]), "+v"(p.a3[i]));
        }
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int j = 0; j < FN; ++j) asm volatile("" : "+v"(p.b1[b][j]), "+v"(p.b2[b][j]), "+v"(p.b3[b][j]));
    };
    // six products per accumulator tile, small cross terms first; `between(k)` runs after the k-th product type
    // (k = 0..5) -- the stage boundary hangs its scalar DMA issue there, under MFMAs already in flight
    auto mfma_planes = [&](const Planes& p, auto&& between) {
#define DCV_SPLIT_PRODUCT(PA, PB, K)                                                                                           \
    _Pragma("unroll") for (int b = 0; b < NB; ++b) _Pragma("unroll") for (int i = 0; i < FM; ++i) _Pragma("unroll") for (int j = 0; j < FN; ++j) \
        acc[b][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(p.PA[i]), as_bf16x8(p.PB[b][j]), acc[b][i][j], 0, 0, 0); \
    between(K);
        DCV_SPLIT_PRODUCT(a3, b1, 0)
        DCV_SPLIT_PRODUCT(a1, b3, 1)
        DCV_SPLIT_PRODUCT(a2, b2, 2)
        DCV_SPLIT_PRODUCT(a2, b1, 3)
        DCV_SPLIT_PRODUCT(a1, b2, 4)
        DCV_SPLIT_PRODUCT(a1, b1, 5)
#undef DCV_SPLIT_PRODUCT
    };
    auto mfma16 = [&](const Frags8& f) {
        Planes p;
        split_frags(f, p);
        mfma_planes(p, [](int) {});
    };
    // Column sums of the A operand as the MFMAs see it (TN products, FP32-input flavour): a lane owns one column of each
    // of its A fragments, so summing the fragment values it feeds to the matrix pipe gives sum_k A[k][m] for free -- the
    // covariance's sum of z_t without a pass of its own over the matrix (4 vector adds per fragment against 16-32 MFMAs).
    constexpr bool kASum = epi_has_asum<Epi>::value && MODE == kTN && !Cfg::SPLIT;
    float asum[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) asum[i] = 0.f;
    auto add_asum = [&](const Frags& f) {
        if constexpr (kASum) {
#pragma unroll
            for (int i = 0; i < FM; ++i) asum[i] += (f.a[i][0] + f.a[i][1]) + (f.a[i][2] + f.a[i][3]);
        }
    };
    auto mfma_step = [&](const Frags& f, int s) {
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[b][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[i][s], f.b[b][j][s], acc[b][i][j], 0, 0, 0);
    };
    auto mfma_group = [&](const Frags& f) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j)
                        acc[b][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[i][s], f.b[b][j][s], acc[b][i][j], 0, 0, 0);
    };

    const int64_t nst = (k_end - k_begin + KB - 1) / KB;
    DCV_STAMP_AT(0);
#ifdef DCV_STAMP
    if (threadIdx.x == 0 && blockIdx.x < 8192 && blockIdx.z == 0)
        g_stamp[blockIdx.x * 8 + 7] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32);
#endif
    constexpr int G = KB / 8;
    auto compute_stage = [&](const float* cur) {
        const float* la = cur;
        const float* lb = cur + A_SZ;
        if constexpr (Cfg::SPLIT) {
#pragma unroll
            for (int s = 0; s < KB / 16; ++s) {
                Frags8 f;
                read_frags8(f, la, lb, s);
                mfma16(f);
            }
            return;
        }
#ifndef DCV_NO_FRAG_DBUF
        // fragments double buffered in registers: group g+1 is read while group g feeds the MFMAs
        Frags f0, f1;
        read_frags(f0, la, lb, 0);
#pragma unroll
        for (int g = 0; g < G; g += 2) {
            if (g + 1 < G) read_frags(f1, la, lb, g + 1);
            add_asum(f0);
            mfma_group(f0);
            if (g + 1 < G) {
                if (g + 2 < G) read_frags(f0, la, lb, g + 2);
                add_asum(f1);
                mfma_group(f1);
            }
        }
#else
#pragma unroll
        for (int g = 0; g < G; ++g) {
            Frags f0;
            read_frags(f0, la, lb, g);
            add_asum(f0);
            mfma_group(f0);
        }
#endif
    };
    // ---- staged-epilogue geometry (used by the side-operand prefetch below and by the epilogue)
    constexpr int LDSF = Cfg::NBUF * STAGE;   // floats of LDS the kernel owns
    constexpr int EPASS = (TM * TN + LDSF - 1) / LDSF;
    constexpr int RP = TM / (EPASS > 0 ? EPASS : 1);   // rows per pass
    constexpr bool kStaged = (NB == 1) && (EPASS <= 2) && (TM % EPASS == 0) && (RP % (FM * 32) == 0) && (RP * TN <= LDSF);
    constexpr int EC4 = TN / 4;                 // 16-byte segments per row
    constexpr int ERPP = 256 / (EC4 > 256 ? 256 : EC4);   // rows per pass of the store loop
    constexpr int ENQ = RP / ERPP;              // row segments per thread and epilogue pass
    // Side operand of the epilogue (the stored activations whose derivative scales a dgrad tile): for a
    // full tile its EPASS * ENQ row segments are requested before the first stage is awaited, so they
    // stream in under the main loop instead of in a burst at the end while the matrix pipe idles
    // (short-K products are otherwise HBM-bound for the length of their epilogue and idle before it).
    constexpr bool kSidePre = kStaged && Epi::kSide && (EPASS * ENQ <= 16) && !Cfg::SPLIT;   // SPLIT: the planes need the registers
    float4 side_pre[kSidePre ? EPASS * ENQ : 1];
    bool side_ready = false;
    // Two main loops in sequence.  The full stages of a workgroup whose tile is fully in range (the
    // common case) run the pure LDS-DMA loop: no staging registers are live in it, so the compiler has
    // no reason to wait on the vector-memory counter inside the MFMA phase.  Whatever remains -- the
    // ragged last stage of such a workgroup, or every stage of an edge tile / scalar-load / covariance
    // shift workgroup -- runs the register-staged loop behind it.
    if constexpr (LS) {
        // ---- split-from-LDS loop (TileCfg::PL bit 4).  The fp32 stage of BOTH operands arrives by LDS-DMA in a raw ring of
        // two buffers behind the plane buffers (no VGPR round trip, the latency of the global loads stays hidden as in the
        // ring loop); one pass per stage reads it back (4 ds_read_b128 per thread), splits it ONCE per workgroup and stores
        // the three plane images the MFMA side reads -- the in-register split repeats that work in every wave that shares
        // a fragment (two of the four waves of a 2 x 2 layout).  Per stage and wave: raw reads of stage st + 1, DMA issue
        // of stage st + 2, the products of stage st from its planes, the split + plane stores of stage st + 1, one barrier.
        // The launch guarantees K % KB == 0 and 16-byte rows; ragged row tiles read clamped rows (MMajorStage::init<true>).
        constexpr int PERA = MMajorStage<TM, KA, VEC, GATHER>::PER, PERB = MMajorStage<TN, KBB, VEC, GATHER>::PER;
        constexpr int RAW = Cfg::RAW_SZ;
        float* const raw0 = lds + Cfg::NBUF * STAGE;
        const int64_t nst_ls = (k_end - k_begin) / KB;
        auto dma_raw = [&](int64_t k0, int rb) {
            float* rbuf = raw0 + rb * RAW;
            const unsigned l0 = ldsw + (unsigned)(Cfg::NBUF * STAGE + rb * RAW) * 4u;
            if (pl_a_aff) am.glds_affine(am.tile_base(A, m0_ld, k0), l0);
            else am.glds(k0, rbuf, t);
            if (pl_b_aff) bm.glds_affine(bm.tile_base(B, n0, k0), l0 + (unsigned)(TM * KB) * 4u);
            else bm.glds(k0, rbuf + TM * KB, t);
        };
        auto raw_to_regs = [&](int rb) {
            const float* rbuf = raw0 + rb * RAW;
#pragma unroll
            for (int i = 0; i < PERA; ++i) am.r[i] = *reinterpret_cast<const float4*>(rbuf + (t + 256 * i) * 4);
#pragma unroll
            for (int i = 0; i < PERB; ++i) bm.r[i] = *reinterpret_cast<const float4*>(rbuf + TM * KB + (t + 256 * i) * 4);
        };
        auto split_to_planes = [&](float* pbuf) {
            am.store_planes(pbuf, t);
            bm.store_planes(pbuf + A_SZ, t);
        };
        if (nst_ls > 0) {
            dma_raw(k_begin, 0);
            if (nst_ls > 1) {
                dma_raw(k_begin + KB, 1);
                vm_wait<PERA + PERB>();   // stage 0 landed (stage 1 may still be in flight)
            } else {
                vm_wait<0>();
            }
            __syncthreads();
            raw_to_regs(0);
            split_to_planes(lds);
            vm_wait<0>();                 // stage 1 landed
            __syncthreads();              // planes of stage 0 published; raw buffer 0 read by everyone
            for (int64_t st = 0; st < nst_ls; ++st) {
                const int cur = (int)(st & 1);
                if (st + 1 < nst_ls) raw_to_regs(cur ^ 1);
                __builtin_amdgcn_sched_barrier(0);
                if (st + 2 < nst_ls) dma_raw(k_begin + (st + 2) * KB, cur);   // raw buffer `cur` was read one barrier ago
                __builtin_amdgcn_sched_barrier(0);
                if (m0 + wm < d.M) compute_stage(lds + cur * STAGE);
                if (st + 1 < nst_ls) split_to_planes(lds + (cur ^ 1) * STAGE);   // that plane buffer was multiplied one barrier ago
                vm_wait<0>();
                __syncthreads();
            }
        }
    }
    const bool dense_ok = !LS && (PRE || (!(PSA || PSB) && nst > 0 && stage_dense(k_begin) && affine_all()));   // split-at-store: every stage through registers
    const int64_t nfull = dense_ok ? (k_end - k_begin) / KB : 0;
    if (nfull > 0) {
        const int64_t nst = nfull;   // stages of the ring loop
        // Ring of NBUF stage buffers, rotated by one fragment group.  At the boundary between stages st
        // and st+1 every wave waits for its own DMA of stage st+1, the barrier publishes that stage and
        // retires buffer st % NBUF (all its fragment reads have returned), the DMA of stage st+NBUF is
        // issued into it (scalar instructions only) and the first fragments of stage st+1 are read --
        // and only then are the MFMAs of the LAST group of stage st issued, from registers: they cover
        // the barrier skew, the DMA issue and the LDS latency, so the matrix pipe does not drain at
        // stage boundaries even with one wave per SIMD.
        constexpr int NBUF = Cfg::NBUF;
        constexpr int GL = (A_MM ? NPA * MMajorStage<TM, KA, VEC, GATHER>::PER : KMajorStage<TM, KB, VEC, GATHER>::PER) +
                           NB * (B_MM ? NPB * MMajorStage<TN, KBB, VEC, GATHER>::PER : KMajorStage<TN, KB, VEC, GATHER>::PER);  // DMA instructions per thread and stage
        static_assert((NBUF - 2) * GL <= 63, "vmcnt range");
        static_assert(Cfg::SPLIT || G % 2 == 0, "fragment double buffer parity");
#pragma unroll
        for (int s = 0; s < NBUF - 1; ++s)
            if (s < nst) glds_stage(k_begin + s * KB, s);
        // boundary "-1 -> 0": stage 0 landed and published; every buffer is still free
        if constexpr (kSidePre) {
            static_assert((NBUF - 2) * GL + EPASS * ENQ <= 63, "vmcnt range");
            side_ready = epi.vec && (m0 + TM <= d.M) && (n0 + TN <= d.N);
            if constexpr (Epi::kMaskIn) side_ready = side_ready && epi.mask == nullptr;
            if (side_ready) {
                const int ec4 = t % EC4, er0 = t / EC4;
#pragma unroll
                for (int q = 0; q < EPASS * ENQ; ++q)
                    side_pre[q] = *reinterpret_cast<const float4*>(epi.side_ptr(m0 + er0 + q * ERPP, n0 + ec4 * 4));
                vm_wait_younger<NBUF - 2, GL, EPASS * ENQ>(nst - 1);   // the side loads are younger than the stage DMAs
            } else {
                vm_wait_younger<NBUF - 2, GL>(nst - 1);
            }
        } else {
            vm_wait_younger<NBUF - 2, GL>(nst - 1);
        }
        __syncthreads();
#ifndef DCV_ABL_NOLOAD
        if (NBUF - 1 < nst) glds_stage(k_begin + (NBUF - 1) * KB, NBUF - 1);
#endif
        DCV_STAMP_AT(1);
        DCV_STAMP_RT(5);
        if constexpr (Cfg::SPLIT) {
            // same ring, 16-deep steps: the last step of a stage is multiplied after the boundary work
            constexpr int S = KB / 16;
            Frags8 f0, f1;
#ifdef DCV_PIPE_SPLIT
            Planes pl0, pl1;
#endif
            read_frags8(f0, lds, lds + A_SZ, 0);
#ifdef DCV_PIPE_SPLIT
            if constexpr (S == 2) split_frags(f0, pl0);
#endif
            int cur_buf = 0;
            for (int64_t st = 0; st < nst; ++st) {
                const float* la = lds + cur_buf * STAGE;
                const float* lb = la + A_SZ;
                const int nxt_buf = cur_buf + 1 == NBUF ? 0 : cur_buf + 1;
#ifdef DCV_PIPE_SPLIT
                if constexpr (S == 2) {
                    // Software pipeline over the 16-deep steps: the six bf16 products of step s run from planes that were
                    // split while step s - 1 was being multiplied, and the operands of step s + 1 are split under them --
                    // one wave keeps the matrix pipe and the vector ALU busy together instead of in turns (with a single
                    // workgroup per CU, the small-batch case, nobody else fills the gaps).  sched_group_barrier asks for
                    // one MFMA followed by VPM vector-ALU instructions, over and over.
                    constexpr int NMF = 6 * NB * FM * FN;                      // MFMAs of a step
                    constexpr int NVA = 44 * (FM + NB * FN);                   // vector-ALU instructions of its split
                    constexpr int VPM = (NVA + NMF - 1) / NMF;
                    read_frags8(f1, la, lb, 1);
                    mfma_planes(pl0, [](int) {});
                    split_frags(f1, pl1);
                    pin_planes(pl1);
#pragma unroll
                    for (int q = 0; q < NMF; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    vm_wait_younger<NBUF - 2, GL>(nst - 2 - st);
                    __syncthreads();
                    read_frags8(f0, lds + nxt_buf * STAGE, lds + nxt_buf * STAGE + A_SZ, 0);
                    const bool more = st + NBUF < nst;
                    const int64_t k_next = k_begin + (st + NBUF) * KB;
                    __builtin_amdgcn_sched_barrier(0);
                    if (more) glds_stage(k_next, cur_buf, 3);
                    __builtin_amdgcn_sched_barrier(0);
                    mfma_planes(pl1, [](int) {});
                    split_frags(f0, pl0);
                    pin_planes(pl0);
#pragma unroll
                    for (int q = 0; q < NMF; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                } else
#endif
                if constexpr (S == 2) {
                    read_frags8(f1, la, lb, 1);
                    mfma16(f0);
                    // the last step's operands are split before the boundary; behind the barrier only its MFMAs
                    // remain, with the scalar DMA issue of stage st + NBUF and the next fragment reads between them
                    Planes pl;
                    split_frags(f1, pl);
                    vm_wait_younger<NBUF - 2, GL>(nst - 2 - st);
#ifndef DCV_ABL_NOBARRIER
                    __syncthreads();
#endif
                    read_frags8(f0, lds + nxt_buf * STAGE, lds + nxt_buf * STAGE + A_SZ, 0);
                    const bool more = st + NBUF < nst;
                    const int64_t k_next = k_begin + (st + NBUF) * KB;
                    mfma_planes(pl, [&](int k) {
#ifndef DCV_ABL_NOLOAD
                        if (k == 0 || k == 2) {
                            __builtin_amdgcn_sched_barrier(0);
                            if (more) glds_stage(k_next, cur_buf, k == 0 ? 1 : 2);
                            __builtin_amdgcn_sched_barrier(0);
                        }
#endif
                    });
                } else {
                    vm_wait_younger<NBUF - 2, GL>(nst - 2 - st);
#ifndef DCV_ABL_NOBARRIER
                    __syncthreads();
#endif
#ifndef DCV_ABL_NOLOAD
                    if (st + NBUF < nst) glds_stage(k_begin + (st + NBUF) * KB, cur_buf);
#endif
                    read_frags8(f1, lds + nxt_buf * STAGE, lds + nxt_buf * STAGE + A_SZ, 0);
                    mfma16(f0);
                    f0 = f1;
                }
                cur_buf = nxt_buf;
            }
            __syncthreads();   // last stage fully read before the buffers are reused
        } else {
        Frags f0, f1;
        read_frags(f0, lds, lds + A_SZ, 0);
        int cur_buf = 0;
        for (int64_t st = 0; st < nst; ++st) {
            const float* la = lds + cur_buf * STAGE;
            const float* lb = la + A_SZ;
            const int nxt_buf = cur_buf + 1 == NBUF ? 0 : cur_buf + 1;
#pragma unroll
            for (int g = 0; g < G; g += 2) {
                read_frags(f1, la, lb, g + 1);
                __builtin_amdgcn_sched_barrier(0);
                pin_frags(f0);
                if (dma_sub) sub_frags(f0);
                add_asum(f0);
                mfma_group(f0);
                __builtin_amdgcn_sched_barrier(0);
                if (g + 2 < G) {
                    read_frags(f0, la, lb, g + 2);
                    __builtin_amdgcn_sched_barrier(0);
                    pin_frags(f1);
                    if (dma_sub) sub_frags(f1);
                    add_asum(f1);
                    mfma_group(f1);
                    __builtin_amdgcn_sched_barrier(0);
                } else {   // stage boundary, interleaved with the last group's MFMAs
                    // One straight-line path (only the DMA issue is conditional): the last stage waits,
                    // synchronises and reads fragments of a stale buffer for nothing, which is cheaper
                    // than the accumulator copies hipcc generates when MFMAs sit on divergent paths.
                    vm_wait_younger<NBUF - 2, GL>(nst - 2 - st);
#ifndef DCV_ABL_NOBARRIER
                    __syncthreads();
#endif
                    __builtin_amdgcn_sched_barrier(0);
                    pin_frags(f1);
                    if (dma_sub) sub_frags(f1);
                    add_asum(f1);
                    mfma_step(f1, 0);
                    __builtin_amdgcn_sched_barrier(0);
#ifndef DCV_ABL_NOLOAD
                    if (st + NBUF < nst) glds_stage(k_begin + (st + NBUF) * KB, cur_buf, 1);
#endif
                    __builtin_amdgcn_sched_barrier(0);
                    mfma_step(f1, 1);
                    __builtin_amdgcn_sched_barrier(0);
#ifndef DCV_ABL_NOLOAD
                    if (st + NBUF < nst) glds_stage(k_begin + (st + NBUF) * KB, cur_buf, 2);
#endif
                    __builtin_amdgcn_sched_barrier(0);
                    mfma_step(f1, 2);
                    __builtin_amdgcn_sched_barrier(0);
                    read_frags(f0, lds + nxt_buf * STAGE, lds + nxt_buf * STAGE + A_SZ, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    mfma_step(f1, 3);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            cur_buf = nxt_buf;
        }
        __syncthreads();   // last stage fully read before the buffers are reused
        }
    } else {
        DCV_STAMP_AT(1);
        DCV_STAMP_RT(5);
    }
    const int64_t k_rem = k_begin + nfull * KB;
    const int64_t nrem = (PRE || LS) ? 0 : (k_end - k_rem + KB - 1) / KB;
    if (nrem > 0) {
        resolve_stage(k_rem);
        load_stage(k_rem);
        if (nrem > 1) resolve_stage(k_rem + KB);
        store_stage(lds);
        __syncthreads();
        for (int64_t st = 0; st < nrem; ++st) {
            const float* cur = lds + (st & 1) * STAGE;
            float* nxt = lds + ((st + 1) & 1) * STAGE;
            const bool more = st + 1 < nrem;
            if (more) {
                load_stage(k_rem + (st + 1) * KB);   // into registers, in flight during the MFMA phase
                if (st + 2 < nrem) resolve_stage(k_rem + (st + 2) * KB);
            }
            // a wave whose output rows are all past the end (the upper half of a 10-row tail tile) has nothing to add: it
            // keeps the loads and barriers and leaves its SIMD to the co-resident workgroup (26.8 -> 25.4 us at 8202 rows)
            if (MODE == kTN || m0 + wm < d.M) compute_stage(cur);
            if (more) store_stage(nxt);
            __syncthreads();
        }
    }
    if constexpr (MODE != kTN && NB == 1) {
        if (tail_chunk >= 0) {   // workgroup-uniform: one contraction chunk of the ragged last row tile (GemmDims::tail_split)
            constexpr int NACC = FM * FN * 16;
            const int S = d.tail_split;
            float* ws = d.tail_ws + (int64_t)tile_n * S * NACC * 256;
            float* mine = ws + (int64_t)tail_chunk * NACC * 256 + t;
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) handoff_store(mine + ((i * FN + j) * 16 + e) * 256, acc[0][i][j][e]);
            // Guide-form hand-off (handoff.h): write-through payload stores, every wave drains them, barrier, one
            // agent-scope ticket; the last arriver acquires at agent scope and reads the chunks with sc1 loads.
            // (Round 2 had a workgroup-scope release here: it orders nothing between workgroups, and hipcc emitted no
            // s_waitcnt vmcnt(0) between the payload stores and the ticket.)
            const bool last = handoff_arrive_last(d.tail_cnt + tile_n, (unsigned)S, reinterpret_cast<unsigned*>(lds));
            __syncthreads();   // the flag word is part of the epilogue's staging area
            if (!last) return;
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[0][i][j][e] = 0.f;
            for (int c = 0; c < S; ++c) {   // chunk order, whoever arrived last: the sum is reproducible
                const float* part = ws + (int64_t)c * NACC * 256 + t;
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j)
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[0][i][j][e] += handoff_load(part + ((i * FN + j) * 16 + e) * 256);
            }
        }
    }
#ifdef DCV_ABL_NOEPI
    {   // keep every accumulator live (no dead-code elimination of the MFMAs), then skip the epilogue
        float sink = 0.f;
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) sink += acc[b][i][j][e];
        if (sink != 12345.678f) return;
    }
#endif

    // ---------------------------------------------------------------- epilogue
    // The accumulator layout (column on the lane, rows across registers) would give 4-byte stores
    // at a row stride; instead the tile is transposed through LDS (the stage buffers are free now)
    // and leaves as 16-byte row segments: 4x fewer, fully coalesced store instructions, and the
    // epilogue's own operand loads (bias, stored activations) become 16-byte loads too.
    // Epi::transform() maps the row segments of a thread in registers (bias + activation, or the
    // activation gradient), out_ptr() says where a segment goes; when Epi::kColSum the per-column
    // sums of the stored values over the tile's rows go to epi.colsum(tile_m, col, sum) (bias
    // gradients for free).
    // When the stage buffers are smaller than the tile (KB = 16) the tile goes through in EPASS row
    // blocks, each written by the waves that own those rows.
    DCV_STAMP_AT(2);
    DCV_STAMP_RT(6);
    static_assert(!Epi::kHead || kStaged, "the fused narrow layer needs the staged epilogue");
    if constexpr (kStaged) {
        float* tile = lds;  // [RP][TN]
        constexpr int C4 = TN / 4;           // 16-byte segments per row
        constexpr int RPP = 256 / C4;        // rows per pass of the store loop
        constexpr int NQ = RP / RPP;         // row segments per thread and epilogue pass
        const int c4 = t % C4, r0 = t / C4;
        const int64_t col = n0 + c4 * 4;
        const int64_t left = d.N - col;
        const int nvalid = left >= 4 ? 4 : (left > 0 ? (int)left : 0);
        // uniform fast path: whole tile in range and 16-byte accesses legal -> no predicates at all
        const bool fast = epi.vec && (m0 + TM <= d.M) && (n0 + TN <= d.N);
        // per-column constants (bias) once; side operands (stored activations) all in flight together;
        // then every LDS segment is read, transformed in registers (the accumulators are dead now)
        // and the 16-byte stores are issued back to back with no wait in between
        const float4 cc = nvalid > 0 ? epi.colconst(col, nvalid) : make_float4(0.f, 0.f, 0.f, 0.f);
        // processed in chunks of EQ row segments to bound the register footprint (2 waves/SIMD)
        constexpr int EQ = NQ < 8 ? NQ : 8;
        static_assert(NQ % EQ == 0, "epilogue chunking");
        float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int ep = 0; ep < EPASS; ++ep) {
            if (ep > 0) __syncthreads();   // previous pass fully read
            if (wm / RP == ep) {
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j) {
                        const int c = wn + j * 32 + (lane & 31);
                        const int rb = wm - ep * RP + i * 32 + 4 * (lane >> 5);
#pragma unroll
                        for (int e = 0; e < 16; ++e) tile[(rb + (e & 3) + 8 * (e >> 2)) * TN + c] = acc[0][i][j][e];
                    }
            }
            __syncthreads();
            if (ep == 0) DCV_STAMP_AT(3);
            const int64_t mbase = m0 + ep * RP;
#pragma unroll
            for (int qc = 0; qc < NQ; qc += EQ) {
                float4 v[EQ], side[Epi::kSide ? EQ : 1];
                bool masked = false;   // workgroup-uniform: the activation derivative comes from a sign mask
                if constexpr (Epi::kMaskIn) masked = epi.mask != nullptr;
                if constexpr (Epi::kSide) {
#pragma unroll
                    for (int q = 0; q < EQ; ++q) {
                        if (masked) {
                            side[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                            continue;
                        }
                        const int64_t row = mbase + r0 + (qc + q) * RPP;
                        if constexpr (kSidePre) {
                            if (side_ready) {   // requested before the main loop (row segment ep * NQ + qc + q of this thread)
                                side[q] = side_pre[ep * NQ + qc + q];
                                continue;
                            }
                        }
                        if (fast) side[q] = *reinterpret_cast<const float4*>(epi.side_ptr(row, col));
                        else side[q] = (row < d.M && nvalid > 0) ? load_quad(epi.side_ptr(row, col), nvalid, epi.vec) : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
#pragma unroll
                for (int q = 0; q < EQ; ++q) v[q] = *reinterpret_cast<const float4*>(tile + (r0 + (qc + q) * RPP) * TN + c4 * 4);
                const int64_t seg0 = ((int64_t)tile_m * d.tiles_n + tile_n) * (EPASS * NQ) + ep * NQ + qc;   // first row segment of this chunk
                if constexpr (Epi::kMaskIn) {
                    if (masked) epi.template apply_mask<EQ>(v, seg0, wave, lane);
                    else epi.template transform<EQ>(v, side, cc);
                } else {
                    epi.template transform<EQ>(v, side, cc);
                }
                // dropout (training only; workgroup-uniform switch): the keep mask of element (row, col) is a pure
                // function of (seed, step, layer, row, col), so the dgrad epilogue recomputes the forward's mask
                if constexpr (Epi::kDrop) {
                    if (epi.drop.thr != 0u) epi.drop.template apply<EQ>(v, mbase + r0 + qc * RPP, RPP, col);
                }
                if constexpr (Epi::kMaskOut) {
                    if (epi.mask != nullptr) store_sign_mask<EQ>(epi.mask, v, seg0, wave, lane);
                }
                if constexpr (Epi::kHead) epi.template head<EQ, C4>(v, mbase + r0 + qc * RPP, RPP, col, nvalid, d.M, lane);
                if (fast) {
#pragma unroll
                    for (int q = 0; q < EQ; ++q) {
                        if (d.wt) {
                            hv4f hv = {v[q].x, v[q].y, v[q].z, v[q].w};
                            handoff_store16(epi.out_ptr(0, mbase + r0 + (qc + q) * RPP, col), hv);
                        } else {
                            *reinterpret_cast<float4*>(epi.out_ptr(0, mbase + r0 + (qc + q) * RPP, col)) = v[q];
                        }
                        if constexpr (Epi::kColSum) {
                            cs.x += v[q].x; cs.y += v[q].y; cs.z += v[q].z; cs.w += v[q].w;
                        }
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < EQ; ++q) {
                        const int64_t row = mbase + r0 + (qc + q) * RPP;
                        if (row < d.M && nvalid > 0) {
                            store_quad(epi.out_ptr(0, row, col), v[q], nvalid, epi.vec);
                            if constexpr (Epi::kColSum) {
                                cs.x += v[q].x;
                                if (nvalid > 1) cs.y += v[q].y;
                                if (nvalid > 2) cs.z += v[q].z;
                                if (nvalid > 3) cs.w += v[q].w;
                            }
                        }
                    }
                }
            }
        }
        DCV_STAMP_AT(4);
        if constexpr (Epi::kColSum) {
            __syncthreads();
            float* red = lds;  // [RPP][TN]
            *reinterpret_cast<float4*>(red + r0 * TN + c4 * 4) = cs;
            __syncthreads();
            for (int c = t; c < TN; c += 256) {
                float v = 0.f;
#pragma unroll
                for (int q = 0; q < RPP; ++q) v += red[q * TN + c];
                if (n0 + c < d.N) epi.colsum(tile_m, n0 + c, v);
            }
        }
    } else {
        static_assert(!Epi::kColSum || kStaged, "column sums need the staged epilogue");
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    const int64_t col = n0 + wn + j * 32 + (lane & 31);
                    const int64_t rbase = m0 + wm + i * 32 + 4 * (lane >> 5);
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int64_t row = rbase + (e & 3) + 8 * (e >> 2);
                        if (row < d.M && col < d.N) epi.one(b, row, col, acc[b][i][j][e]);
                    }
                }
    }
    if constexpr (kASum) {
        // one copy per (row tile of C = column block of A, contraction chunk): the first column tile's waves of column 0
        if (epi.asum != nullptr && tile_n == 0 && (wave % Cfg::WAVES_N) == 0) {
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const float v = asum[i] + __shfl_xor(asum[i], 32, 64);   // the two k-halves of the lanes
                const int64_t m = m0 + wm + i * 32 + (lane & 31);
                if (lane < 32 && m < d.M) epi.asum[epi.z * epi.M + m] = v;
            }
        }
    }
}

template <class Cfg, int NB>
constexpr size_t gemm_lds_bytes() {
    return ((size_t)Cfg::NBUF * (Cfg::A_SZ + NB * Cfg::B_SZ) + 2 * (size_t)Cfg::RAW_SZ) * sizeof(float);
}

}  // namespace dcv
