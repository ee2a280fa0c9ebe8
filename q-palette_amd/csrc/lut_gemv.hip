// VQ/SQ tensor-core-format fused decode+GEMV kernels (batch 1..8) + launcher.
#include "lut_kernels_api.h"

namespace qpal {

int launch_lut_tc_gemv(const TcMultiParams &p, int bits, int vec, int nbg, int grid, hipStream_t stream) {
    for (int j = 0; j < p.njobs; j++)
        if (p.job[j].x_rot) return nbg == 1 ? launch_lut_tc_gemv_rot(p, bits, vec, grid, stream) : QPAL_E_SHAPE;
    const TcEarly e = early_args(p, grid);
#define QPAL_LUT(B_, V_)                                                                                       \
    if (bits == B_ && vec == V_) {                                                                             \
        if (nbg == 1)                                                                                          \
            hipLaunchKernelGGL((tc_gemv_kernel<LutCodec<B_, V_>, void, 1>), dim3(grid), dim3(64 * gemv_waves<1>()), 0, stream, e.x, e.tab, e.n, e.k, e.on, e.ie, e.su, e.rw, p); \
        else if (nbg == 2)                                                                                     \
            hipLaunchKernelGGL((tc_gemv_kernel<LutCodec<B_, V_>, void, 2>), dim3(grid), dim3(64 * gemv_waves<2>()), 0, stream, e.x, e.tab, e.n, e.k, e.on, e.ie, e.su, e.rw, p); \
        else                                                                                                   \
            return launch_lut_tc_gemv_wide(p, bits, vec, nbg, grid, stream);                                   \
        return (int)hipGetLastError();                                                                         \
    }
#include "lut_table.inc"
#undef QPAL_LUT
    return QPAL_E_PARAM;
}

}  // namespace qpal
