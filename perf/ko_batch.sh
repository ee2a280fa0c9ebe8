#!/bin/bash
# Knock-out timing of the lockstep skinny-GEMM step (results invalid by construction): [WL=<workload>] bash perf/ko_batch.sh <batch> <variants...>
N=$1; shift
for v in "$@"; do
  lib=$PWD/q-palette_amd/libqpal_hip${v:+_$v}.so; [ "$v" = base ] && lib=$PWD/q-palette_amd/libqpal_hip.so
  QPAL_LIB=$lib timeout -k 10 300 python bench.py --workload ${WL:-llama3.1-8b_tcomb_6_7} --batch $N --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs --no-calibration --no-incoherent-extra --no-kind-breakdown --no-whole-model 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('batch %3d %-8s: %7.3f ms/step' % ($N, '$v', d['ms_per_step']))"
done
