#!/bin/bash
# A/B of the two batched kernels (QPAL_GEMM=0: per-wave K chunks, x from L2; 1: lockstep rows, x tile through LDS) over batch sizes:
#   bash perf/ab_batch.sh [workload] [batches...]
WL=${1:-llama3.1-8b_tcomb_6_7}; shift; B=${@:-9 16 17 32 33 64}
run() { QPAL_GEMM=$1 timeout -k 10 300 python bench.py --workload $WL --batch $2 --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs --no-incoherent-extra --no-kind-breakdown --no-whole-model 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$WL batch %3d QPAL_GEMM=$1: %8.1f tok/s %7.3f ms/step  mfma frac %.4f' % ($2, d['value'], d['ms_per_step'], d.get('roofline_mfma',{}).get('frac',0)))"; }
for n in $B; do run 0 $n; run 1 $n; done
