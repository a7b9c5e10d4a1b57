// Layer forward H = act(X W^T + b): the NT product with the bias / activation / dropout epilogue.
#include "gemm_kernels.h"

namespace dcv {

int gemm_nt_bias_act(const Operand& A, const Operand& B, int64_t M, int64_t N, int64_t K, const EpiBiasAct& epi, hipStream_t s, const TailWs* tw) {
    return launch_gemm<kNT, EpiBiasAct>(A, B, M, N, K, 0, epi, s, nullptr, tw);
}

}  // namespace dcv
