// VQ/SQ (tensor-core packing) fused decode + skinny GEMM for batches 17..64 (512-thread workgroups).
#include "lut_kernels_api.h"

namespace qpal {

int launch_lut_tc_gemv_wide(const TcMultiParams &p, int bits, int vec, int nbg, int grid, hipStream_t stream) {
    const TcEarly e = early_args(p, grid);
#define QPAL_LUT(B_, V_)                                                                                       \
    if (bits == B_ && vec == V_) {                                                                             \
        if (nbg == 4) {                                                                                        \
            hipLaunchKernelGGL((tc_gemv_kernel<LutCodec<B_, V_>, void, 4>), dim3(grid), dim3(512), 0, stream, e.x, e.tab, e.n, e.k, e.on, e.ie, e.su, e.rw, p); \
        } else if constexpr (LutCodec<B_, V_>::LDS_DWORDS * 4 <= 64 * 1024) {  /* 8 groups: 64 KiB reduction buffer */ \
            hipLaunchKernelGGL((tc_gemv_kernel<LutCodec<B_, V_>, void, 8>), dim3(grid), dim3(512), 0, stream, e.x, e.tab, e.n, e.k, e.on, e.ie, e.su, e.rw, p); \
        } else {                                                                                               \
            return QPAL_E_SHAPE;                                                                               \
        }                                                                                                      \
        return (int)hipGetLastError();                                                                         \
    }
#include "lut_table.inc"
#undef QPAL_LUT
    return QPAL_E_PARAM;
}

}  // namespace qpal
