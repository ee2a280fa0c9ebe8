// Launchers of the TCQ kernel instantiations (one translation unit per batch width).
#pragma once
#include "tc_kernels.h"

namespace qpal {
int launch_tcq_gemv_nb1(const TcParams &p, int S, int KV1, int KV2, int grid, hipStream_t stream);
int launch_tcq_gemv_nb2(const TcParams &p, int S, int KV1, int KV2, int grid, hipStream_t stream);
int launch_tcq_gemv_nb4(const TcParams &p, int S, int KV1, int KV2, int grid, hipStream_t stream);
int launch_tcq_gemv_nb8(const TcParams &p, int S, int KV1, int KV2, int grid, hipStream_t stream);
int launch_tcq_dequant(const TcParams &p, int S, int KV1, int KV2, int grid, hipStream_t stream);
}  // namespace qpal
