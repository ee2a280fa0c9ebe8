// TCQ fused decode + skinny GEMM with the step's activations shared through LDS (tc_gemm.h), 2 batch groups of 8 (batch 9..16).
#define QPAL_GEMM_NBG 2
#include "tcq_gemm_inst.inc"
