// VQ/SQ tensor-core-format fused decode+GEMV kernels, batch width 8.
#define QPAL_NB 8
#include "lut_gemv_inst.inc"
