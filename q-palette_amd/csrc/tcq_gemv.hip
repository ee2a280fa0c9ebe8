// TCQ fused decode+GEMV kernels (batch 1..8) + launcher.
#include "tcq_launch.h"

namespace qpal {

int launch_tcq_gemv(const TcMultiParams &p, int S, int KV1, int KV2, int nbg, int grid, hipStream_t stream) {
    for (int j = 0; j < p.njobs; j++)
        if (p.job[j].x_rot) return nbg == 1 ? launch_tcq_gemv_rot(p, S, KV1, KV2, grid, stream) : QPAL_E_SHAPE;
#define QPAL_TCQ(S_, A_, B_)                                                                 \
    if (S == S_ && KV1 == A_ && KV2 == B_)                                                   \
        return nbg == 1 ? launch_one<S_, A_, B_, 1>(p, grid, stream) : nbg == 2 ? launch_one<S_, A_, B_, 2>(p, grid, stream) : \
               launch_tcq_gemv_wide(p, S, KV1, KV2, nbg, grid, stream);
#include "tcq_table.inc"
#undef QPAL_TCQ
    return QPAL_E_PARAM;
}

}  // namespace qpal
