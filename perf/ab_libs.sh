#!/bin/bash
# Same-box A/B of library builds on the headline token (batch 1, multi-job launches):  bash perf/ab_libs.sh <out> <lib suffix ...>
# ("base" = q-palette_amd/libqpal_hip.so, "x" = libqpal_hip_x.so; three interleaved rounds)
out=$1; shift; mkdir -p $(dirname $out)
B="python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-other-configs --no-incoherent-extra --no-whole-model --no-calibration"
for r in 1 2 3; do for v in "$@"; do
  lib=$PWD/q-palette_amd/libqpal_hip_$v.so; [ "$v" = base ] && lib=$PWD/q-palette_amd/libqpal_hip.so
  QPAL_LIB=$lib timeout -k 10 300 $B 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['roofline'].get('by_launch_kind',{}); print('%-8s %7.1f tok/s %.4f ms' % ('$v', d['value'], d['ms_per_step']), {a:round(b['us_per_launch'],2) for a,b in k.items()})" >> $out
done; done; cat $out
