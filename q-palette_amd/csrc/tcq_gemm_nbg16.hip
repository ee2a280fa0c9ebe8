// TCQ fused decode + skinny GEMM, lane-pair-exchange mapping (tc_gemm16.h), 8 groups of 16 batch rows (batch <= 128).
#define QPAL_GEMM_NBG 16
#include "tcq_gemm_inst.inc"
