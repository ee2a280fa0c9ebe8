// TCQ fused decode + skinny GEMM, lane-pair-exchange mapping (tc_gemm16.h), 4 groups of 16 batch rows (batch <= 64).
#define QPAL_GEMM_NBG 8
#include "tcq_gemm_inst.inc"
