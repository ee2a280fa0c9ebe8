// Launchers of the persistent chain kernels (tc_chain.h).
#pragma once
#include "tc_chain.h"

namespace qpal {
int launch_tcq_chain(const TcMultiParams *phases, int nphases, ChainWs *ws, int S, int KV1, int KV2, int grid,
                     unsigned long long *dbg, hipStream_t stream);
int launch_lut_chain(const TcMultiParams *phases, int nphases, ChainWs *ws, int bits, int vec, int grid,
                     unsigned long long *dbg, hipStream_t stream);
}  // namespace qpal
