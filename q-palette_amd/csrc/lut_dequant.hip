// VQ/SQ tensor-core-format decode-to-fp16 kernels, and the TC -> SIMT re-pack.
#include "lut_kernels_api.h"

namespace qpal {

int launch_lut_tc_dequant(const TcParams &p, int bits, int vec, int grid, hipStream_t stream) {
#define QPAL_LUT(B_, V_)                                                                                   \
    if (bits == B_ && vec == V_) {                                                                         \
        hipLaunchKernelGGL((tc_dequant_kernel<LutCodec<B_, V_>, void>), dim3(grid), dim3(1024), 0, stream, p); \
        return (int)hipGetLastError();                                                                     \
    }
#include "lut_table.inc"
#undef QPAL_LUT
    return QPAL_E_PARAM;
}

int launch_tc_to_simt(uint32_t *dst, const uint32_t *src, int m, int k, int bits, int vec, hipStream_t stream) {
    const long ncodes = (long)m * k / vec;
    long blocks = (ncodes + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(tc_to_simt_kernel<0>, dim3((int)blocks), dim3(256), 0, stream, dst, src, m, k, bits, vec);
    return (int)hipGetLastError();
}

}  // namespace qpal
