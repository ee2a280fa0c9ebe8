#!/bin/bash
# Same-box A/B of library builds on the batched path: bash perf/ab_batch_libs.sh "<batches>" lib1.so lib2.so ...   (two interleaved passes)
NS=$1; shift
run() { QPAL_LIB=q-palette_amd/$1 timeout -k 10 300 python bench.py --workload ${WL:-llama3.1-8b_tcomb_6_7} --batch $2 --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs --no-incoherent-extra --no-kind-breakdown --no-whole-model --no-calibration 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%-36s batch %4d  %9.1f tok/s  %8.4f ms/step' % ('$1', $2, d['value'], d['ms_per_step']))"; }
for pass in 1 2; do for n in $NS; do for lib in "$@"; do run $lib $n; done; done; done
