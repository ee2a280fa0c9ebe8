#!/bin/bash
# Same-box A/B of two library builds on the batched path:  bash perf/ab_lib_batch.sh <libA> <libB> [workload] [batches...]
A=$1; B=$2; WL=${3:-llama3.1-8b_tcomb_6_7}; shift 3; NS=${@:-4 8 16 32}
run() { QPAL_LIB=$1 timeout -k 10 300 python bench.py --workload $WL --batch $2 --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs --no-incoherent-extra --no-kind-breakdown --no-whole-model --no-calibration 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$WL batch %3d %-40s: %8.1f tok/s %7.3f ms/step' % ($2, '$1'.split('/')[-1], d['value'], d['ms_per_step']))"; }
for r in 1 2; do for n in $NS; do run $A $n; run $B $n; done; done
