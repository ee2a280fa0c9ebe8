// VQ/SQ tensor-core-format fused decode+GEMV kernels, batch width 4.
#define QPAL_NB 4
#include "lut_gemv_inst.inc"
