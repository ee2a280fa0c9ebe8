// VQ/SQ tensor-core-format fused decode+GEMV kernels (batch 1..8) + launcher.
#include "lut_kernels_api.h"
#include "seq.h"

namespace qpal {

int launch_lut_tc_gemv(const TcMultiParams &p, int bits, int vec, int nbg, int grid, hipStream_t stream) {
    for (int j = 0; j < p.njobs; j++)
        if (p.job[j].x_rot) return nbg == 1 ? launch_lut_tc_gemv_rot(p, bits, vec, grid, stream) : QPAL_E_SHAPE;
    const TcEarly e = early_args(p);
#define QPAL_LUT(B_, V_)                                                                                       \
    if (bits == B_ && vec == V_) {                                                                             \
        if (nbg == 1) return launch_gemv_kernel<LutCodec<B_, V_>, void, 1, 0>(e, p, grid, stream);              \
        if (nbg == 2) return launch_gemv_kernel<LutCodec<B_, V_>, void, 2, 0>(e, p, grid, stream);              \
        return launch_lut_tc_gemv_wide(p, bits, vec, nbg, grid, stream);                                       \
    }
#include "lut_table.inc"
#undef QPAL_LUT
    return QPAL_E_PARAM;
}

}  // namespace qpal
