// VQ/SQ (tensor-core packing) persistent chain kernels (tc_chain.h) + launcher.
#include "chain_api.h"

namespace qpal {

int launch_lut_chain(const TcMultiParams *phases, int nphases, ChainWs *ws, int bits, int vec, int grid,
                     unsigned long long *dbg, hipStream_t stream) {
#define QPAL_LUT(B_, V_)                                                                                              \
    if (bits == B_ && vec == V_) {                                                                                    \
        hipLaunchKernelGGL((tc_chain_kernel<LutCodec<B_, V_>, void>), dim3(grid), dim3(kChainThreads), 0, stream, phases, \
                           nphases, ws, dbg);                                                                         \
        return (int)hipGetLastError();                                                                                \
    }
#include "lut_table.inc"
#undef QPAL_LUT
    return QPAL_E_PARAM;
}

}  // namespace qpal
