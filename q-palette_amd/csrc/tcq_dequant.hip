// TCQ decode-to-fp16 kernels + launcher.
#include "tcq_kernels_api.h"

namespace qpal {

template <int S, int KV1, int KV2>
static int launch_one(const TcParams &p, int grid, hipStream_t stream) {
    using C1 = TcqCodec<S, KV1>;
    using C2 = std::conditional_t<KV2 == 0, void, TcqCodec<S, KV2 == 0 ? KV1 : KV2>>;
    hipLaunchKernelGGL((tc_dequant_kernel<C1, C2>), dim3(grid), dim3(1024), 0, stream, p);
    return (int)hipGetLastError();
}

int launch_tcq_dequant(const TcParams &p, int S, int KV1, int KV2, int grid, hipStream_t stream) {
#define QPAL_TCQ(S_, A_, B_) \
    if (S == S_ && KV1 == A_ && KV2 == B_) return launch_one<S_, A_, B_>(p, grid, stream);
#include "tcq_table.inc"
#undef QPAL_TCQ
    return QPAL_E_PARAM;
}

}  // namespace qpal
