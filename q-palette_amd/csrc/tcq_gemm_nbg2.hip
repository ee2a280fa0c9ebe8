// TCQ fused decode + skinny GEMM, lane-pair-exchange mapping (tc_gemm16.h), 1 group of 16 batch rows (batch <= 16).
#define QPAL_GEMM_NBG 2
#include "tcq_gemm_inst.inc"
