// VQ/SQ tensor-core-format GEMV kernels whose x staging applies the 14336-wide rotation of rot_k28.h (codecs whose codebook image
// can lend it 40 KiB of LDS; the host refuses the others: qpal_capi.hip rot_k28_ok).
#include "lut_kernels_api.h"

namespace qpal {

int launch_lut_tc_gemv_rot28(const TcMultiParams &p, int bits, int vec, int grid, hipStream_t stream) {
    const TcEarly e = early_args(p, grid);
#define QPAL_LUT(B_, V_)                                                                                           \
    if (bits == B_ && vec == V_) {                                                                                 \
        if constexpr (LutCodec<B_, V_>::LDS_DWORDS * 4 >= kP28 * kTbRow) {                                         \
            hipLaunchKernelGGL((tc_gemv_kernel<LutCodec<B_, V_>, void, 1, 2>), dim3(grid), dim3(64 * gemv_waves<1>()), 0, stream, e.x, e.tab, e.n, e.k, e.on, e.ie, e.su, e.rw, p); \
            return (int)hipGetLastError();                                                                         \
        } else {                                                                                                   \
            return QPAL_E_SHAPE;                                                                                   \
        }                                                                                                          \
    }
#include "lut_table.inc"
#undef QPAL_LUT
    return QPAL_E_PARAM;
}

}  // namespace qpal
