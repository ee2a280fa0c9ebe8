// TCQ fused decode+GEMV kernels, batch width 8.
#define QPAL_NB 8
#include "tcq_gemv_inst.inc"
