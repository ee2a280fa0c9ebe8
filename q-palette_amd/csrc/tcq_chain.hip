// TCQ persistent chain kernels (tc_chain.h) + launcher.
#include "chain_api.h"

namespace qpal {

template <int S, int KV1, int KV2>
static int launch_chain_one(const TcMultiParams *phases, int nphases, ChainWs *ws, int grid, unsigned long long *dbg,
                            hipStream_t stream) {
    using C1 = TcqCodec<S, KV1>;
    using C2 = std::conditional_t<KV2 == 0, void, TcqCodec<S, KV2 == 0 ? KV1 : KV2>>;
    hipLaunchKernelGGL((tc_chain_kernel<C1, C2>), dim3(grid), dim3(kChainThreads), 0, stream, phases, nphases, ws, dbg);
    return (int)hipGetLastError();
}

int launch_tcq_chain(const TcMultiParams *phases, int nphases, ChainWs *ws, int S, int KV1, int KV2, int grid,
                     unsigned long long *dbg, hipStream_t stream) {
#define QPAL_TCQ(S_, A_, B_) \
    if (S == S_ && KV1 == A_ && KV2 == B_) return launch_chain_one<S_, A_, B_>(phases, nphases, ws, grid, dbg, stream);
#include "tcq_table.inc"
#undef QPAL_TCQ
    return QPAL_E_PARAM;
}

}  // namespace qpal
