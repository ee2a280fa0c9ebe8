// TCQ fused decode + skinny GEMM with the step's activations shared through LDS (tc_gemm.h), 8 batch groups (batch <= 64).
#define QPAL_GEMM_NBG 8
#include "tcq_gemm_inst.inc"
