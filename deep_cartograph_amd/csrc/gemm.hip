// Plain C = op(A) . op(B) entry point over the FP32 MFMA engine (building block, used by the
// parity tests to check every operand form and tile shape against a float64 product).
#include <type_traits>
#include "gemm_kernels.h"

using namespace dcv;

namespace dcv {
int gemm_nn_store(const Operand& A, const Operand& B, int64_t M, int64_t N, int64_t K, const EpiStore& epi, hipStream_t s) {
    return launch_gemm<kNN, EpiStore>(A, B, M, N, K, 0, epi, s);
}
}  // namespace dcv

extern "C" int dcv_gemm_f32(int32_t mode, const float* A_d, int64_t lda, const float* B_d, int64_t ldb, float* C_d,
                            int64_t ldc, int64_t M, int64_t N, int64_t K, void* stream) {
    DCV_REQUIRE(A_d && B_d && C_d && M > 0 && N > 0 && K > 0, "dcv_gemm_f32: bad arguments");
    hipStream_t s = as_stream(stream);
    const Operand A = make_operand(A_d, lda, 0);
    const Operand B = make_operand(B_d, ldb, 0);
    EpiStore epi{C_d, ldc, quad_ok(C_d, ldc)};
    switch (mode) {
        case kNT: return launch_gemm<kNT, EpiStore>(A, B, M, N, K, 0, epi, s);
        case kNN: return gemm_nn_store(A, B, M, N, K, epi, s);
        case kTN: return launch_gemm<kTN, EpiStore>(A, B, M, N, K, 0, epi, s);  // single split
        default: set_error("dcv_gemm_f32: mode %d", mode); return DCV_EINVAL;
    }
}

extern "C" int dcv_gemm_tn_split(const float* A_d, int64_t lda, const float* B_d, int64_t ldb, float* slab_d, int64_t slab_cap,
                                 int64_t M, int64_t N, int64_t K, int64_t k_chunk, void* stream) {
    DCV_REQUIRE(A_d && B_d && slab_d && M > 0 && N > 0 && K > 0 && k_chunk > 0 && slab_cap > 0, "dcv_gemm_tn_split: bad arguments");
    const Operand A = make_operand(A_d, lda, 0);
    const Operand B = make_operand(B_d, ldb, 0);
    EpiSlab epi{slab_d, M, N, 1, 0, quad_ok(slab_d, N), slab_cap};
    return gemm_tn_slab(A, B, M, N, K, k_chunk, epi, as_stream(stream));
}
