// TCQ fused decode+GEMV for launches whose jobs have DIFFERENT KV (same codebook size S): one kernel per S that switches
// to the job's decode loop per work item (tc_kernels.h: TcqAny).  Single-stream layers, batch <= 8.
#include "seq.h"
#include "tcq_kernels_api.h"

namespace qpal {

int launch_tcq_gemv_any(const TcMultiParams &p, int S, int grid, hipStream_t stream) {
    const TcEarly e = early_args(p);
    if (S == 9) return launch_gemv_kernel<TcqAny<9>, void, 1, 0>(e, p, grid, stream);
    if (S == 10) return launch_gemv_kernel<TcqAny<10>, void, 1, 0>(e, p, grid, stream);
    if (S == 11) return launch_gemv_kernel<TcqAny<11>, void, 1, 0>(e, p, grid, stream);
    return QPAL_E_PARAM;
}

}  // namespace qpal
