// Launchers of the TCQ kernel instantiations .
#pragma once
#include "tc_kernels.h"

namespace qpal {
int launch_tcq_gemv(const TcMultiParams &p, int S, int KV1, int KV2, int nbg, int grid, hipStream_t stream);
int launch_tcq_gemv_wide(const TcMultiParams &p, int S, int KV1, int KV2, int nbg, int grid, hipStream_t stream);
int launch_tcq_gemm_nbg1(const TcMultiParams &p, int S, int KV1, int KV2, int grid, hipStream_t stream);
int launch_tcq_gemm_nbg2(const TcMultiParams &p, int S, int KV1, int KV2, int grid, hipStream_t stream);
int launch_tcq_gemm_nbg4(const TcMultiParams &p, int S, int KV1, int KV2, int grid, hipStream_t stream);
int launch_tcq_gemm_nbg8(const TcMultiParams &p, int S, int KV1, int KV2, int grid, hipStream_t stream);
int launch_tcq_gemm_nbg10(const TcMultiParams &p, int S, int KV1, int KV2, int grid, hipStream_t stream);
int launch_tcq_gemm_nbg16(const TcMultiParams &p, int S, int KV1, int KV2, int grid, hipStream_t stream);
int launch_tcq_gemv_any(const TcMultiParams &p, int S, int grid, hipStream_t stream);
int launch_tcq_gemv_rot28(const TcMultiParams &p, int S, int KV1, int KV2, int grid, hipStream_t stream);
int launch_tcq_gemv_rot(const TcMultiParams &p, int S, int KV1, int KV2, int grid, hipStream_t stream);
int launch_tcq_dequant(const TcParams &p, int S, int KV1, int KV2, int grid, hipStream_t stream);
}  // namespace qpal
