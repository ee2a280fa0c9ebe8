// TCQ fused decode + skinny GEMM with the step's activations shared through LDS (tc_gemm.h), 4 batch groups (batch <= 32).
#define QPAL_GEMM_NBG 4
#include "tcq_gemm_inst.inc"
