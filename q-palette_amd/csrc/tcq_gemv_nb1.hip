// TCQ fused decode+GEMV kernels, batch width 1.
#define QPAL_NB 1
#include "tcq_gemv_inst.inc"
