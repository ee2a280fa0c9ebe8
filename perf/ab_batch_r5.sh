#!/bin/bash
# Round 5: the lockstep GEMM with the lane-pair exchange (tc_gemm16.h) against the round's starting library, one box.
#   bash perf/ab_batch_r5.sh            tokens/s, ms per step at batches 8..128 (the old library: up to 64)
run() { QPAL_LIB=$1 timeout -k 10 300 python bench.py --workload $3 --batch $2 --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs --no-incoherent-extra --no-kind-breakdown --no-whole-model --no-calibration 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%-22s %-36s batch %4d  %9.1f tok/s  %8.4f ms/step  mfma frac %s' % ('$3', '$1'.split('/')[-1], $2, d['value'], d['ms_per_step'], ('%.3f' % d['roofline_mfma']['frac']) if 'roofline_mfma' in d else '-'))"; }
for pass in 1 2; do
  for n in 8 16 32 64; do run q-palette_amd/libqpal_hip_r5base.so $n llama3.1-8b_tcomb_6_7; run q-palette_amd/libqpal_hip.so $n llama3.1-8b_tcomb_6_7; done
  for n in 65 96 128 129 256; do run q-palette_amd/libqpal_hip.so $n llama3.1-8b_tcomb_6_7; done
done
run q-palette_amd/libqpal_hip_r5base.so 16 llama3.1-70b_tcq_6; run q-palette_amd/libqpal_hip.so 16 llama3.1-70b_tcq_6
run q-palette_amd/libqpal_hip.so 64 llama3.1-70b_tcq_6; run q-palette_amd/libqpal_hip.so 128 llama3.1-70b_tcq_6
