// VQ/SQ (tensor-core packing) fused decode + skinny GEMM with the step's activations shared through LDS (tc_gemm.h).
#include "tc_gemm.h"
#include "lut_kernels_api.h"

namespace qpal {

int launch_lut_tc_gemm(const TcMultiParams &p, int bits, int vec, int nbg, int grid, hipStream_t stream) {
#define QPAL_LUT(B_, V_)                                                                                                   \
    if (bits == B_ && vec == V_) {                                                                                         \
        using C = LutCodec<B_, V_>;                                                                                        \
        if (nbg == 1) hipLaunchKernelGGL((tc_gemm_kernel<C, void, 1>), dim3(grid), dim3(64 * kGemmWaves), 0, stream, gemm_item_table(p), p);   \
        else if (nbg == 2) hipLaunchKernelGGL((tc_gemm_kernel<C, void, 2>), dim3(grid), dim3(64 * kGemmWaves), 0, stream, gemm_item_table(p), p);   \
        else if (nbg == 4) hipLaunchKernelGGL((tc_gemm_kernel<C, void, 4>), dim3(grid), dim3(64 * kGemmWaves), 0, stream, gemm_item_table(p), p); \
        else if constexpr (C::LDS_DWORDS * 4 + 2 * 8 * kGemmXGroup <= 160 * 1024)                                          \
            hipLaunchKernelGGL((tc_gemm_kernel<C, void, 8>), dim3(grid), dim3(64 * kGemmWaves), 0, stream, gemm_item_table(p), p);             \
        else return QPAL_E_SHAPE;                                                                                          \
        return (int)hipGetLastError();                                                                                     \
    }
#include "lut_table.inc"
#undef QPAL_LUT
    return QPAL_E_PARAM;
}

}  // namespace qpal
