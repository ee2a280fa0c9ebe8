#!/bin/bash
# planner sweep of the lockstep skinny-GEMM kernel: bash perf/sweep_gemm.sh <batch>
n=${1:-64}
for items in 96 192 256 512; do for ms in 2 4; do
QPAL_GEMM_ITEMS=$items QPAL_GEMM_MINSTEPS=$ms QPAL_GEMM=1 timeout -k 10 300 python bench.py --batch $n --steps 20 --warmup 3 --no-cpu-baseline --no-incoherent-extra --no-kind-breakdown --no-whole-model 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('batch $n items $items minsteps $ms: %8.1f tok/s %7.3f ms/step' % (d['value'], d['ms_per_step']))"; done; done
