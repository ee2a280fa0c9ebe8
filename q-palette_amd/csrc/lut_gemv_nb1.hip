// VQ/SQ tensor-core-format fused decode+GEMV kernels, batch width 1.
#define QPAL_NB 1
#include "lut_gemv_inst.inc"
