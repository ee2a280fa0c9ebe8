// SIMT-format fused decode+GEMV kernels, batch width 4.
#define QPAL_NB 4
#include "simt_gemv_inst.inc"
