// TCQ fused decode+GEMV kernels, batch width 4.
#define QPAL_NB 4
#include "tcq_gemv_inst.inc"
