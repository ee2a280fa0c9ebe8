// TCQ fused decode+GEMV kernels whose x staging applies the 14336-wide rotation hadK(28) (x) H_512 (rot_k28.h): down_proj of
// Llama-3.1-8B behind the incoherence wrapper.  Their own instantiation (ROT = 2): the rotation's registers would make the other
// rotating launches spill.
#include "tcq_launch.h"

namespace qpal {

int launch_tcq_gemv_rot28(const TcMultiParams &p, int S, int KV1, int KV2, int grid, hipStream_t stream) {
#define QPAL_TCQ(S_, A_, B_)                                                                                        \
    if (S == S_ && KV1 == A_ && KV2 == B_) {                                                                        \
        return launch_one<S_, A_, B_, 1, 2>(p, grid, stream);                                                       \
    }
#include "tcq_table.inc"
#undef QPAL_TCQ
    return QPAL_E_PARAM;
}

}  // namespace qpal
