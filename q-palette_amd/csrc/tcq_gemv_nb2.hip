// TCQ fused decode+GEMV kernels, batch width 2.
#define QPAL_NB 2
#include "tcq_gemv_inst.inc"
