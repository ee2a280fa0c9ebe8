// TCQ fused decode+GEMV for launches whose jobs have DIFFERENT KV (same codebook size S): one kernel per S that switches
// to the job's decode loop per work item (tc_kernels.h: TcqAny).  Single-stream layers, batch <= 8.
#include "tcq_kernels_api.h"

namespace qpal {

int launch_tcq_gemv_any(const TcMultiParams &p, int S, int grid, hipStream_t stream) {
    const TcEarly e = early_args(p, grid);
    if (S == 9) hipLaunchKernelGGL((tc_gemv_kernel<TcqAny<9>, void, 1, false>), dim3(grid), dim3(64 * gemv_waves<1>()), 0, stream, e.x, e.tab, e.n, e.k, e.on, e.ie, e.su, e.rw, p);
    else if (S == 10) hipLaunchKernelGGL((tc_gemv_kernel<TcqAny<10>, void, 1, false>), dim3(grid), dim3(64 * gemv_waves<1>()), 0, stream, e.x, e.tab, e.n, e.k, e.on, e.ie, e.su, e.rw, p);
    else if (S == 11) hipLaunchKernelGGL((tc_gemv_kernel<TcqAny<11>, void, 1, false>), dim3(grid), dim3(64 * gemv_waves<1>()), 0, stream, e.x, e.tab, e.n, e.k, e.on, e.ie, e.su, e.rw, p);
    else return QPAL_E_PARAM;
    return (int)hipGetLastError();
}

}  // namespace qpal
