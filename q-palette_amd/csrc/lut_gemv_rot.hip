// VQ/SQ tensor-core-format GEMV kernels that also apply the incoherence rotation to x while staging it.
#include "lut_kernels_api.h"

namespace qpal {

int launch_lut_tc_gemv_rot(const TcMultiParams &p, int bits, int vec, int grid, hipStream_t stream) {
    if (p.job[0].x_rot == kK28) return launch_lut_tc_gemv_rot28(p, bits, vec, grid, stream);
    const TcEarly e = early_args(p, grid);
#define QPAL_LUT(B_, V_)                                                                                           \
    if (bits == B_ && vec == V_) {                                                                                 \
        hipLaunchKernelGGL((tc_gemv_kernel<LutCodec<B_, V_>, void, 1, 1>), dim3(grid), dim3(64 * gemv_waves<1>()), 0, stream, e.x, e.tab, e.n, e.k, e.on, e.ie, e.su, e.rw, p); \
        return (int)hipGetLastError();                                                                             \
    }
#include "lut_table.inc"
#undef QPAL_LUT
    return QPAL_E_PARAM;
}

}  // namespace qpal
