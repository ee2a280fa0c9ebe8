// SIMT-format decode-to-fp16 kernels and the batch-width dispatcher of the SIMT GEMV.
#include "lut_kernels_api.h"

namespace qpal {

int launch_simt_gemv_nb1(const SimtParams &p, int bits, int vec, const SimtGeometry &g, hipStream_t stream);
int launch_simt_gemv_nb2(const SimtParams &p, int bits, int vec, const SimtGeometry &g, hipStream_t stream);
int launch_simt_gemv_nb4(const SimtParams &p, int bits, int vec, const SimtGeometry &g, hipStream_t stream);
int launch_simt_gemv_nb8(const SimtParams &p, int bits, int vec, const SimtGeometry &g, hipStream_t stream);

int launch_simt_gemv(const SimtParams &p, int bits, int vec, int nb, const SimtGeometry &g, hipStream_t stream) {
    switch (nb) {
        case 1: return launch_simt_gemv_nb1(p, bits, vec, g, stream);
        case 2: return launch_simt_gemv_nb2(p, bits, vec, g, stream);
        case 4: return launch_simt_gemv_nb4(p, bits, vec, g, stream);
        default: return launch_simt_gemv_nb8(p, bits, vec, g, stream);
    }
}

int launch_simt_dequant(const SimtParams &p, int bits, int vec, int grid, hipStream_t stream) {
#define QPAL_SIMT(B_, V_)                                                                         \
    if (bits == B_ && vec == V_) {                                                                \
        hipLaunchKernelGGL((simt_kernel<B_, V_, 1, 1>), dim3(grid), dim3(1024), 0, stream, p);    \
        return (int)hipGetLastError();                                                            \
    }
#include "simt_table.inc"
#undef QPAL_SIMT
    return QPAL_E_PARAM;
}

}  // namespace qpal
