// Grouped launch of a layer's weight gradient (TN, split-K slabs) and input gradient (NN): one kernel, two block ranges.
// Both read the same dZ and neither reads the other's result; launched one after the other at a small batch each leaves
// part of the chip idle (8202 rows: 256 and 516 workgroups) and pays its own ramp and drain (11.4 + 12.7 us measured).
#include "gemm_kernels.h"

namespace dcv {

template <bool S, class Cfg1, class Cfg2>
static int launch_pair_cfg(const Operand& A1, const Operand& B1, int64_t M1, int64_t N1, int64_t K1, int64_t kc1, const EpiSlab& e1,
                           const Operand& A2, const Operand& B2, int64_t M2, int64_t N2, int64_t K2, const EpiActGrad& e2, int* tiles_m_out2,
                           const TailWs* tw, hipStream_t s) {
    GemmPlan p1, p2;
    int rc = prepare_gemm<kTN, Cfg1, 1, EpiSlab>(A1, B1, M1, N1, K1, kc1, e1, nullptr, nullptr, &p1);
    if (rc) return rc;
    rc = prepare_gemm<kNN, Cfg2, 1, EpiActGrad>(A2, B2, M2, N2, K2, 0, e2, tiles_m_out2, tw, &p2);
    if (rc) return rc;
    if (!p1.vec || p1.gather || !p2.vec || p2.gather) return 1;
    const int64_t blocks1 = (int64_t)p1.d.tiles_m * p1.d.tiles_n * p1.splits;
    const int64_t blocks2 = p2.d.tail_split > 0 ? (int64_t)(p2.d.tiles_m - 1 + p2.d.tail_split) * p2.d.tiles_n : (int64_t)p2.d.tiles_m * p2.d.tiles_n;
    if (blocks1 + blocks2 >= (1ll << 31)) return 1;
    if (blocks1 % 8 != 0) p2.d.xcd_remap = 0;   // the map's congruences are taken on blockIdx.x - blocks1
    constexpr size_t l1 = gemm_lds_bytes<Cfg1, 1>(), l2 = gemm_lds_bytes<Cfg2, 1>();
    constexpr size_t lds = l1 > l2 ? l1 : l2;
    auto kern = wgrad_dgrad_kernel<Cfg1, Cfg2>;
    if (lds > 64 * 1024) {
        static bool attr_set = false;  // per instantiation
        if (!attr_set) {
            DCV_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_set = true;
        }
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(blocks1 + blocks2)), dim3(256), lds, s, A1, B1, p1.d, e1, (int)blocks1, A2, B2, p2.d, e2);
    DCV_CHECK_LAUNCH();
    return DCV_OK;
}

template <bool S>
static int launch_pair_mode(const Operand& A1, const Operand& B1, int64_t M1, int64_t N1, int64_t K1, int64_t kc1, const EpiSlab& e1,
                            const Operand& A2, const Operand& B2, int64_t M2, int64_t N2, int64_t K2, const EpiActGrad& e2, int* tiles_m_out2,
                            const TailWs* tw, hipStream_t s) {
    const CfgPick c1 = pick_cfg<kTN, false>(M1, N1, K1, kc1), c2 = pick_cfg<kNN, false>(M2, N2, K2, 0);
    if (c1 == kPickQuarter && c2 == kPickQuarter)
        return launch_pair_cfg<S, CfgQuarterT<S>, CfgQuarterT<S>>(A1, B1, M1, N1, K1, kc1, e1, A2, B2, M2, N2, K2, e2, tiles_m_out2, tw, s);
    if (c1 == kPickBig && c2 == kPickBig)
        return launch_pair_cfg<S, CfgBigT<S>, CfgBigT<S>>(A1, B1, M1, N1, K1, kc1, e1, A2, B2, M2, N2, K2, e2, tiles_m_out2, tw, s);
    // mixed families (round 4): a gathered (shuffled) batch of 8192 pairs is 16 384 rows -- the weight gradient still takes
    // 64 x 64 split-K tiles, the input gradient 64 x 128 row tiles: launched apart they were 12.8 + 13.7 us of the 152 us step
    if (c1 == kPickQuarter && c2 == kPickHalfM)
        return launch_pair_cfg<S, CfgQuarterT<S>, CfgHalfMT<S>>(A1, B1, M1, N1, K1, kc1, e1, A2, B2, M2, N2, K2, e2, tiles_m_out2, tw, s);
    if (c1 == kPickBig && c2 == kPickHalfM)
        return launch_pair_cfg<S, CfgBigT<S>, CfgHalfMT<S>>(A1, B1, M1, N1, K1, kc1, e1, A2, B2, M2, N2, K2, e2, tiles_m_out2, tw, s);
    return 1;
}

int launch_wgrad_dgrad(const Operand& A1, const Operand& B1, int64_t M1, int64_t N1, int64_t K1, int64_t k_chunk1, const EpiSlab& e1,
                       const Operand& A2, const Operand& B2, int64_t M2, int64_t N2, int64_t K2, const EpiActGrad& e2, int* tiles_m_out2,
                       const TailWs* tw, hipStream_t s) {
    static const bool off = [] { const char* e = getenv("DCV_NO_PAIR"); return e && e[0] == '1'; }();
    if (off) return 1;
    if (gemm_split()) return launch_pair_mode<true>(A1, B1, M1, N1, K1, k_chunk1, e1, A2, B2, M2, N2, K2, e2, tiles_m_out2, tw, s);
    return launch_pair_mode<false>(A1, B1, M1, N1, K1, k_chunk1, e1, A2, B2, M2, N2, K2, e2, tiles_m_out2, tw, s);
}

}  // namespace dcv
