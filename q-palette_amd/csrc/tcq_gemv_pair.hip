// TCQ fused decode+GEMV kernels (batch 1..8) that understand pair-mode jobs (TcParams: sk == -1): launches the planner paired.
#include "tcq_launch.h"

namespace qpal {

int launch_tcq_gemv_pair(const TcMultiParams &p, int S, int KV1, int KV2, int grid, hipStream_t stream) {
#define QPAL_TCQ(S_, A_, B_) \
    if (S == S_ && KV1 == A_ && KV2 == B_) return launch_one<S_, A_, B_, 1, 3>(p, grid, stream);
#include "tcq_table.inc"
#undef QPAL_TCQ
    return QPAL_E_PARAM;
}

}  // namespace qpal
