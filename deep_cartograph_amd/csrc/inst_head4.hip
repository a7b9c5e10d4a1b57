// Layer forward with the fused linear head of up to 4 outputs.
#include "gemm_kernels.h"

namespace dcv {

int gemm_nt_head4(const Operand& A, const Operand& B, int64_t M, int64_t N, int64_t K, const EpiBiasActHead<4>& epi, hipStream_t s, const TailWs* tw) {
    return launch_gemm<kNT, EpiBiasActHead<4>>(A, B, M, N, K, 0, epi, s, nullptr, tw);
}

}  // namespace dcv
