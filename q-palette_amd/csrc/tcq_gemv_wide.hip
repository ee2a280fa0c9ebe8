// TCQ fused decode + skinny GEMM for batches 17..64 (4 / 8 MFMA column groups per decoded step, 512-thread workgroups).
#include "tcq_launch.h"

namespace qpal {

int launch_tcq_gemv_wide(const TcMultiParams &p, int S, int KV1, int KV2, int nbg, int grid, hipStream_t stream) {
#define QPAL_TCQ(S_, A_, B_)                                                                 \
    if (S == S_ && KV1 == A_ && KV2 == B_)                                                   \
        return nbg == 4 ? launch_one<S_, A_, B_, 4>(p, grid, stream) : launch_one<S_, A_, B_, 8>(p, grid, stream);
#include "tcq_table.inc"
#undef QPAL_TCQ
    return QPAL_E_PARAM;
}

}  // namespace qpal
