// TCQ fused decode + skinny GEMM, lane-pair-exchange mapping (tc_gemm16.h), 2 groups of 16 batch rows (batch <= 32).
#define QPAL_GEMM_NBG 4
#include "tcq_gemm_inst.inc"
