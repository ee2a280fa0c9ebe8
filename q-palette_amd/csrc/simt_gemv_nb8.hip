// SIMT-format fused decode+GEMV kernels, batch width 8.
#define QPAL_NB 8
#include "simt_gemv_inst.inc"
