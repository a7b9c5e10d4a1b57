// Layer forward with the fused linear head of 5..8 outputs.
#include "gemm_kernels.h"

namespace dcv {

int gemm_nt_head8(const Operand& A, const Operand& B, int64_t M, int64_t N, int64_t K, const EpiBiasActHead<8>& epi, hipStream_t s, const TailWs* tw) {
    return launch_gemm<kNT, EpiBiasActHead<8>>(A, B, M, N, K, 0, epi, s, nullptr, tw);
}

}  // namespace dcv
