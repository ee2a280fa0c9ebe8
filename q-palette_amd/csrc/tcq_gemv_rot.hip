// TCQ fused decode+GEMV kernels that also apply the incoherence rotation to x while staging it (x_rot jobs).
#include "tcq_launch.h"

namespace qpal {

int launch_tcq_gemv_rot(const TcMultiParams &p, int S, int KV1, int KV2, int grid, hipStream_t stream) {
    if (p.job[0].x_rot == kK28) return launch_tcq_gemv_rot28(p, S, KV1, KV2, grid, stream);  // the 14336-wide rotation: its own kernels
#define QPAL_TCQ(S_, A_, B_)                                                                                        \
    if (S == S_ && KV1 == A_ && KV2 == B_) {                                                                        \
        return launch_one<S_, A_, B_, 1, 1>(p, grid, stream);                                                    \
    }
#include "tcq_table.inc"
#undef QPAL_TCQ
    return QPAL_E_PARAM;
}

}  // namespace qpal
