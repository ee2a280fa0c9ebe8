// TCQ fused decode + skinny GEMM with the step's activations shared through LDS (tc_gemm.h), 1 batch group (batch 2..8).
#define QPAL_GEMM_NBG 1
#include "tcq_gemm_inst.inc"
