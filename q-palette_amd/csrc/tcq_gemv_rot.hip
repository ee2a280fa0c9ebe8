// TCQ fused decode+GEMV kernels that also apply the incoherence rotation to x while staging it (x_rot jobs).
#include "tcq_kernels_api.h"

namespace qpal {

int launch_tcq_gemv_rot(const TcMultiParams &p, int S, int KV1, int KV2, int grid, hipStream_t stream) {
#define QPAL_TCQ(S_, A_, B_)                                                                                        \
    if (S == S_ && KV1 == A_ && KV2 == B_) {                                                                        \
        using C1 = TcqCodec<S_, A_>;                                                                                \
        using C2 = std::conditional_t<B_ == 0, void, TcqCodec<S_, B_ == 0 ? A_ : B_>>;                              \
        hipLaunchKernelGGL((tc_gemv_kernel<C1, C2, 1, true>), dim3(grid), dim3(1024), 0, stream, p);                \
        return (int)hipGetLastError();                                                                              \
    }
#include "tcq_table.inc"
#undef QPAL_TCQ
    return QPAL_E_PARAM;
}

}  // namespace qpal
