// SIMT-format fused decode+GEMV kernels, batch width 2.
#define QPAL_NB 2
#include "simt_gemv_inst.inc"
