// VQ/SQ tensor-core-format fused decode+GEMV kernels, batch width 2.
#define QPAL_NB 2
#include "lut_gemv_inst.inc"
