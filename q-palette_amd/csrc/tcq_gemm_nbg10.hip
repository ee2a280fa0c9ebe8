// TCQ fused decode + skinny GEMM, lane-pair-exchange mapping (tc_gemm16.h), 5 groups of 16 batch rows (batch 65..80: four tile slots
// still fit beside the image, so 65 rows cost 5 / 4 of 64, not the 8-group kernel's barrier per step).
#define QPAL_GEMM_NBG 10
#include "tcq_gemm_inst.inc"
