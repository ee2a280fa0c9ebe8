// VQ/SQ tensor-core-format fused decode+GEMV kernels (batch 1..8) that understand pair-mode jobs (TcParams: sk == -1).
#include "lut_kernels_api.h"

namespace qpal {

int launch_lut_tc_gemv_pair(const TcMultiParams &p, int bits, int vec, int grid, hipStream_t stream) {
    const TcEarly e = early_args(p);
#define QPAL_LUT(B_, V_)                                                                                       \
    if (bits == B_ && vec == V_) {                                                                             \
        hipLaunchKernelGGL((tc_gemv_kernel<LutCodec<B_, V_>, void, 1, 3>), dim3(grid), dim3(64 * gemv_waves<1>()), 0, stream, e.x, e.tab, e.n, e.k, e.on, e.ie, e.su, e.rw, p); \
        return (int)hipGetLastError();                                                                         \
    }
#include "lut_table.inc"
#undef QPAL_LUT
    return QPAL_E_PARAM;
}

}  // namespace qpal
