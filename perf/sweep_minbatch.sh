#!/bin/bash
# where does the lockstep kernel (tc_gemm.h) overtake the per-wave-K-chunk kernel at small batches?
for n in 2 3 4 6 8; do for mb in 9 2; do
QPAL_GEMM_MIN_BATCH=$mb timeout -k 10 300 python bench.py --batch $n --steps 30 --warmup 5 --no-cpu-baseline --no-incoherent-extra --no-kind-breakdown --no-whole-model --no-calibration 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('batch $n  %s: %8.1f tok/s %7.3f ms/step' % ('gemm kernel' if $mb <= $n else 'gemv kernel', d['value'], d['ms_per_step']))"; done; done
