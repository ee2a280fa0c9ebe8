// Launchers of the VQ/SQ (LUT) kernel instantiations.
#pragma once
#include "simt_kernels.h"
#include "tc_kernels.h"

namespace qpal {
int launch_lut_tc_gemv(const TcMultiParams &p, int bits, int vec, int nbg, int grid, hipStream_t stream);
int launch_lut_tc_gemv_wide(const TcMultiParams &p, int bits, int vec, int nbg, int grid, hipStream_t stream);
int launch_lut_tc_gemm(const TcMultiParams &p, int bits, int vec, int nbg, int grid, hipStream_t stream);
int launch_lut_tc_gemv_rot28(const TcMultiParams &p, int bits, int vec, int grid, hipStream_t stream);
int launch_lut_tc_gemv_rot(const TcMultiParams &p, int bits, int vec, int grid, hipStream_t stream);
int launch_lut_tc_dequant(const TcParams &p, int bits, int vec, int grid, hipStream_t stream);
int launch_simt_gemv(const SimtParams &p, int bits, int vec, int nb, const SimtGeometry &g, hipStream_t stream);
int launch_simt_dequant(const SimtParams &p, int bits, int vec, int grid, hipStream_t stream);
int launch_tc_to_simt(uint32_t *dst, const uint32_t *src, int m, int k, int bits, int vec, hipStream_t stream);
}  // namespace qpal
