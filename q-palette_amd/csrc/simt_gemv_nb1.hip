// SIMT-format fused decode+GEMV kernels, batch width 1.
#define QPAL_NB 1
#include "simt_gemv_inst.inc"
