// VQ/SQ (tensor-core packing) fused decode + skinny GEMM with the step's activations shared through LDS (tc_gemm.h).
#include "tc_gemm16.h"
#include "lut_kernels_api.h"

namespace qpal {

int launch_lut_tc_gemm(const TcMultiParams &p, int bits, int vec, int nbg, int grid, hipStream_t stream) {
// nbg = 1, 2: batches <= 8 / 16 (tc_gemm.h); 4, 8, 16: groups of 16 batch rows with the lane-pair exchange (tc_gemm16.h) where the
// codebook image leaves room for two x tiles
#define QPAL_G16(NG_)                                                                                                      \
    if constexpr (C::LDS_DWORDS * 4 + 2 * NG_ * kG16Group <= 160 * 1024)                                                   \
        hipLaunchKernelGGL((tc_gemm16_kernel<C, void, NG_>), dim3(grid), dim3(64 * kG16Waves), 0, stream, gemm_item_table(p), p); \
    else return QPAL_E_SHAPE;
#define QPAL_LUT(B_, V_)                                                                                                   \
    if (bits == B_ && vec == V_) {                                                                                         \
        using C = LutCodec<B_, V_>;                                                                                        \
        if (nbg == 1) hipLaunchKernelGGL((tc_gemm_kernel<C, void, 1>), dim3(grid), dim3(64 * kGemmWaves), 0, stream, gemm_item_table(p), p);   \
        else if (nbg == 2) hipLaunchKernelGGL((tc_gemm_kernel<C, void, 2>), dim3(grid), dim3(64 * kGemmWaves), 0, stream, gemm_item_table(p), p);   \
        else if (nbg == 4) { QPAL_G16(2) }                                                                                 \
        else if (nbg == 8) { QPAL_G16(4) }                                                                                 \
        else if (nbg == 10) { QPAL_G16(5) }                                                                                \
        else { QPAL_G16(8) }                                                                                               \
        return (int)hipGetLastError();                                                                                     \
    }
#include "lut_table.inc"
#undef QPAL_LUT
#undef QPAL_G16
    return QPAL_E_PARAM;
}

}  // namespace qpal
