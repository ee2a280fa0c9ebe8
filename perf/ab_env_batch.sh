#!/bin/bash
# A/B of one environment knob on the batched path: bash perf/ab_env_batch.sh VAR "v1 v2 ..." [workload] [batches...]
VAR=$1; VALS=$2; WL=${3:-llama3.1-8b_tcomb_6_7}; shift 3; NS=${@:-16 32 64}
for r in 1 2; do for n in $NS; do for v in $VALS; do
  env $VAR=$v timeout -k 10 300 python bench.py --workload $WL --batch $n --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs --no-incoherent-extra --no-kind-breakdown --no-whole-model 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$WL batch %3d $VAR=%-4s: %8.1f tok/s %7.3f ms/step' % ($n, '$v', d['value'], d['ms_per_step']))"
done; done; done
