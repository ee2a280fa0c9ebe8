// Fused decode+GEMV launches whose jobs are of DIFFERENT FAMILIES: TCQ layers (any KV of one codebook size S, single-stream or
// column-split) and VQ/SQ layers in tensor-core packing, e.g. q | k | v of a mixed-scheme model (tc_kernels.h: TcqMix).
// Batch <= 8, no fused rotation.
#include "tcq_kernels_api.h"

namespace qpal {

int launch_tcq_gemv_mix(const TcMultiParams &p, int S, int grid, hipStream_t stream) {
    TcEarly e = early_args(p);
    e.on = 0;  // the early staging builds the TCQ image from preloaded arguments: jobs of another family have another image
    if (S == 9) hipLaunchKernelGGL((tc_gemv_kernel<TcqMix<9>, void, 1, false>), dim3(grid), dim3(64 * gemv_waves<1>()), 0, stream, e.x, e.tab, e.n, e.k, e.on, e.ie, e.su, e.rw, p);
    else if (S == 10) hipLaunchKernelGGL((tc_gemv_kernel<TcqMix<10>, void, 1, false>), dim3(grid), dim3(64 * gemv_waves<1>()), 0, stream, e.x, e.tab, e.n, e.k, e.on, e.ie, e.su, e.rw, p);
    else if (S == 11) hipLaunchKernelGGL((tc_gemv_kernel<TcqMix<11>, void, 1, false>), dim3(grid), dim3(64 * gemv_waves<1>()), 0, stream, e.x, e.tab, e.n, e.k, e.on, e.ie, e.su, e.rw, p);
    else return QPAL_E_PARAM;
    return (int)hipGetLastError();
}

}  // namespace qpal
