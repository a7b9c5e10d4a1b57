// Weight gradient dW = dZ^T X as contraction-split slabs (TN).
#include "gemm_kernels.h"

namespace dcv {

int gemm_tn_slab(const Operand& A, const Operand& B, int64_t M, int64_t N, int64_t K, int64_t k_chunk, const EpiSlab& epi, hipStream_t s) {
    return launch_gemm<kTN, EpiSlab>(A, B, M, N, K, k_chunk, epi, s);
}

}  // namespace dcv
