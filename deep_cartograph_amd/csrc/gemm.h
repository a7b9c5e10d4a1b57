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
// M, N, K works; 16-byte global loads are used when the operand allows it.
#pragma once
#include "common.h"

namespace dcv {

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum GemmMode { kNT = 0, kNN = 1, kTN = 2 };

// Logical row -> row of the underlying matrix.  Batch matrices of the MLP engine are never
// materialised: logical row r < half is sample r (x_t), r >= half is sample r-half shifted by
// `lag` rows (x_lag); a sample's row is idx[j] or row0 + j.
struct RowMap {
    const int64_t* idx;
    int64_t row0;
    int32_t half;
    int32_t lag;
    __device__ __forceinline__ int64_t operator()(int64_t r) const {
        int64_t j = r, off = 0;
        if (half > 0 && r >= half) {
            j = r - half;
            off = lag;
        }
        return (idx ? idx[j] : row0 + j) + off;
    }
};
inline RowMap identity_rows() { return RowMap{nullptr, 0, 0, 0}; }

struct Operand {
    const float* p;
    int64_t ld;
    RowMap rows;
    const float* shift;  // per-(non-contraction)-column value subtracted on load (KMAJOR only), or null
    int vec_ok;          // 16-byte loads legal
};

struct GemmDims {
    int64_t M, N, K;
    int64_t k_chunk;  // TN: contraction rows per blockIdx.z (K for the others)
    int tiles_m, tiles_n;
};

template <int WAVES_M_, int WAVES_N_, int FM_, int FN_, int KB_>
struct TileCfg {
    static constexpr int WAVES_M = WAVES_M_, WAVES_N = WAVES_N_, FM = FM_, FN = FN_, KB = KB_;
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
// MMAJOR tile [T][KB]: unit u = t + 256*i -> row = u / CPR, chunk = u % CPR.
template <int T, int KB>
struct MMajorStage {
    static constexpr int CPR = KB / 4;
    static constexpr int UNITS = T * CPR;
    static constexpr int PER = (UNITS + 255) / 256;
    float4 r[PER];

    __device__ __forceinline__ void load(const Operand& op, int64_t m0, int64_t m_end, int64_t k0, int64_t k_end, int t) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int u = t + 256 * i;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (UNITS % 256 == 0 || u < UNITS) {
                const int row = u / CPR, chunk = u % CPR;
                const int64_t m = m0 + row;
                const int64_t k = k0 + chunk * 4;
                if (m < m_end && k < k_end) {
                    const float* src = op.p + op.rows(m) * op.ld + k;
                    if (op.vec_ok && k + 3 < k_end) {
                        v = *reinterpret_cast<const float4*>(src);
                    } else {
                        v.x = src[0];
                        if (k + 1 < k_end) v.y = src[1];
                        if (k + 2 < k_end) v.z = src[2];
                        if (k + 3 < k_end) v.w = src[3];
                    }
                }
            }
            r[i] = v;
        }
    }
    __device__ __forceinline__ void store(float* lds, int t) const {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int u = t + 256 * i;
            if (UNITS % 256 == 0 || u < UNITS) {
                const int row = u / CPR, chunk = u % CPR;
                *reinterpret_cast<float4*>(lds + mmajor_off<KB>(row, chunk)) = r[i];
            }
        }
    }
};

// KMAJOR tile [KB][T]: unit u -> krow = u / (T/4), c4 = u % (T/4).
template <int T, int KB>
struct KMajorStage {
    static constexpr int C4 = T / 4;
    static constexpr int UNITS = KB * C4;
    static constexpr int PER = (UNITS + 255) / 256;
    float4 r[PER];

    // row_off: extra rows added after the row map (time lag of the second covariance operand)
    __device__ __forceinline__ void load(const Operand& op, int64_t c0, int64_t c_end, int64_t k0, int64_t k_end,
                                         int64_t row_off, int t) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int u = t + 256 * i;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (UNITS % 256 == 0 || u < UNITS) {
                const int krow = u / C4, c4 = u % C4;
                const int64_t k = k0 + krow;
                const int64_t c = c0 + c4 * 4;
                if (k < k_end && c < c_end) {
                    const float* src = op.p + (op.rows(k) + row_off) * op.ld + c;
                    if (op.vec_ok && c + 3 < c_end) {
                        v = *reinterpret_cast<const float4*>(src);
                        if (op.shift) {
                            const float4 s = *reinterpret_cast<const float4*>(op.shift + c);
                            v.x -= s.x; v.y -= s.y; v.z -= s.z; v.w -= s.w;
                        }
                    } else {
                        v.x = src[0] - (op.shift ? op.shift[c] : 0.f);
                        if (c + 1 < c_end) v.y = src[1] - (op.shift ? op.shift[c + 1] : 0.f);
                        if (c + 2 < c_end) v.z = src[2] - (op.shift ? op.shift[c + 2] : 0.f);
                        if (c + 3 < c_end) v.w = src[3] - (op.shift ? op.shift[c + 3] : 0.f);
                    }
                }
            }
            r[i] = v;
        }
    }
    __device__ __forceinline__ void store(float* lds, int t) const {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int u = t + 256 * i;
            if (UNITS % 256 == 0 || u < UNITS) {
                const int krow = u / C4, c4 = u % C4;
                *reinterpret_cast<float4*>(lds + krow * T + c4 * 4) = r[i];
            }
        }
    }
};

// ------------------------------------------------------------------ fragments
// MMAJOR fragment of the 32 rows starting at `rb`, k-group g: 4 k values per lane.
template <int KB>
__device__ __forceinline__ float4 frag_mmajor(const float* lds, int rb, int g, int lane) {
    const int row = rb + (lane & 31);
    const int chunk = 2 * g + (lane >> 5);
    return *reinterpret_cast<const float4*>(lds + mmajor_off<KB>(row, chunk));
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

// ------------------------------------------------------------------ the kernel body
// NB: number of B operands sharing A (2 for the lagged covariance: B and B shifted by `lag2`).
// Epi::operator()(which, row, col, value) is called for every in-range output element.
template <int MODE, class Cfg, int NB, class Epi>
__device__ __forceinline__ void gemm_block(const Operand& A, const Operand& B, int64_t lag2, const GemmDims& d,
                                           int tile_m, int tile_n, int64_t k_begin, int64_t k_end, float* lds,
                                           Epi& epi) {
    constexpr int TM = Cfg::TM, TN = Cfg::TN, KB = Cfg::KB, FM = Cfg::FM, FN = Cfg::FN;
    constexpr int A_SZ = TM * KB, B_SZ = TN * KB;
    constexpr int STAGE = A_SZ + NB * B_SZ;
    constexpr bool A_MM = (MODE == kNT || MODE == kNN);
    constexpr bool B_MM = (MODE == kNT);
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int wm = (wave / Cfg::WAVES_N) * FM * 32;
    const int wn = (wave % Cfg::WAVES_N) * FN * 32;
    const int64_t m0 = (int64_t)tile_m * TM, n0 = (int64_t)tile_n * TN;

    f32x16 acc[NB][FM][FN];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[b][i][j][e] = 0.f;

    MMajorStage<TM, KB> am;
    KMajorStage<TM, KB> ak;
    MMajorStage<TN, KB> bm;
    KMajorStage<TN, KB> bk[NB];

    auto load_stage = [&](int64_t k0) {
        if constexpr (A_MM) am.load(A, m0, d.M, k0, k_end, t);
        else ak.load(A, m0, d.M, k0, k_end, 0, t);
        if constexpr (B_MM) bm.load(B, n0, d.N, k0, k_end, t);
        else {
            bk[0].load(B, n0, d.N, k0, k_end, 0, t);
            if constexpr (NB == 2) bk[1].load(B, n0, d.N, k0, k_end, lag2, t);
        }
    };
    auto store_stage = [&](float* buf) {
        if constexpr (A_MM) am.store(buf, t);
        else ak.store(buf, t);
        if constexpr (B_MM) bm.store(buf + A_SZ, t);
        else {
            bk[0].store(buf + A_SZ, t);
            if constexpr (NB == 2) bk[1].store(buf + A_SZ + B_SZ, t);
        }
    };

    const int64_t nst = (k_end - k_begin + KB - 1) / KB;
    if (nst > 0) {
        load_stage(k_begin);
        store_stage(lds);
    }
    __syncthreads();
    for (int64_t st = 0; st < nst; ++st) {
        const float* cur = lds + (st & 1) * STAGE;
        float* nxt = lds + ((st + 1) & 1) * STAGE;
        const bool more = st + 1 < nst;
        if (more) load_stage(k_begin + (st + 1) * KB);
        const float* la = cur;
        const float* lb = cur + A_SZ;
#pragma unroll
        for (int g = 0; g < KB / 8; ++g) {
            float4 a4[FM], b4[FN];
            if constexpr (A_MM) {
#pragma unroll
                for (int i = 0; i < FM; ++i) a4[i] = frag_mmajor<KB>(la, wm + i * 32, g, lane);
            }
            if constexpr (B_MM) {
#pragma unroll
                for (int j = 0; j < FN; ++j) b4[j] = frag_mmajor<KB>(lb, wn + j * 32, g, lane);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                float av[FM], bv[NB][FN];
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    if constexpr (A_MM) av[i] = f4_get(a4[i], s);
                    else av[i] = frag_kmajor<TM>(la, wm + i * 32, g, s, lane);
                }
#pragma unroll
                for (int b = 0; b < NB; ++b)
#pragma unroll
                    for (int j = 0; j < FN; ++j) {
                        if constexpr (B_MM) bv[b][j] = f4_get(b4[j], s);
                        else bv[b][j] = frag_kmajor<TN>(lb + b * B_SZ, wn + j * 32, g, s, lane);
                    }
#pragma unroll
                for (int b = 0; b < NB; ++b)
#pragma unroll
                    for (int i = 0; i < FM; ++i)
#pragma unroll
                        for (int j = 0; j < FN; ++j)
                            acc[b][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[b][j], acc[b][i][j], 0, 0, 0);
            }
        }
        if (more) store_stage(nxt);
        __syncthreads();
    }

    // epilogue
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int64_t col = n0 + wn + j * 32 + (lane & 31);
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int64_t row = m0 + wm + i * 32 + acc_row(e, lane);
                    if (row < d.M && col < d.N) epi(b, row, col, acc[b][i][j][e]);
                }
            }
}

template <class Cfg, int NB>
constexpr size_t gemm_lds_bytes() {
    return (size_t)2 * (Cfg::TM * Cfg::KB + NB * Cfg::TN * Cfg::KB) * sizeof(float);
}

}  // namespace dcv
