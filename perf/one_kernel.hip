// Scratch translation unit for ISA inspection: instantiates ONLY the headline kernel (tcomb_6_7 fused GEMV, batch <= 8).
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -Iinclude -Iq-palette_amd/csrc -mllvm -amdgpu-kernarg-preload-count=8 \
//         -S --cuda-device-only perf/one_kernel.hip -o /tmp/k/one.s -Rpass-analysis=kernel-resource-usage
#include "tc_kernels.h"
namespace qpal {
template __global__ void tc_gemv_kernel<TcqCodec<9, 6>, TcqCodec<9, 7>, 1, false>(const uint16_t *, const void *, int, int, int, const TcMultiParams);
}
