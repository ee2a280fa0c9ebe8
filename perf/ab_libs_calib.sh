#!/bin/bash
# as perf/ab_libs.sh, with the calibration block (decode rate on register-resident words, stream token):  bash perf/ab_libs_calib.sh <out> <lib suffix ...>
out=$1; shift; mkdir -p $(dirname $out)
B="python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-other-configs --no-incoherent-extra --no-whole-model"
for r in 1 2 3; do for v in "$@"; do
  lib=$PWD/q-palette_amd/libqpal_hip_$v.so; [ "$v" = base ] && lib=$PWD/q-palette_amd/libqpal_hip.so
  QPAL_LIB=$lib timeout -k 10 300 $B 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; k=r.get('by_launch_kind',{}); print('%-8s %7.1f tok/s %.4f ms' % ('$v', d['value'], d['ms_per_step']), {a:round(b['us_per_launch'],2) for a,b in k.items()}, 'decode floor %.4f ms' % r['decode_floor_ms'], {a:round(b['ns_per_wave_step_per_simd_slot'],1) for a,b in r['decode_rate'].items()})" >> $out
done; done; cat $out
