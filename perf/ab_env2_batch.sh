#!/bin/bash
# A/B of environment settings on the batched path: bash perf/ab_env2_batch.sh "A=1,B=2 A=0,B=3 ..." [workload] [batches...]
SETS=$1; WL=${2:-llama3.1-8b_tcomb_6_7}; shift 2; NS=${@:-16 32 64}
for r in 1 2; do for n in $NS; do for st in $SETS; do
  env $(echo $st | tr ',' ' ') timeout -k 10 300 python bench.py --workload $WL --batch $n --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs --no-incoherent-extra --no-kind-breakdown --no-whole-model 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$WL batch %3d %-50s: %8.1f tok/s %7.3f ms/step' % ($n, '$st', d['value'], d['ms_per_step']))"
done; done; done
