// Input gradient dZ_prev = (dZ W) * act'(H) (NN) with the bias-gradient partials.
#include "gemm_kernels.h"

namespace dcv {

int gemm_nn_act_grad(const Operand& A, const Operand& B, int64_t M, int64_t N, int64_t K, const EpiActGrad& epi, hipStream_t s,
                     int* tiles_m_out, const TailWs* tw) {
    return launch_gemm<kNN, EpiActGrad>(A, B, M, N, K, 0, epi, s, tiles_m_out, tw);
}

}  // namespace dcv
