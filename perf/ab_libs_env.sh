#!/bin/bash
# Same-box A/B of (library, environment) settings:  bash perf/ab_libs_env.sh [-w workload] [-a "bench args"] "QPAL_LIB=q-palette_amd/libqpal_hip_x.so" "A=0" ...
# (a setting is a space-separated list of VAR=value; QPAL_LIB picks another build of the library).  Three interleaved passes.
WL=llama3.1-8b_tcomb_6_7; ARGS=""
while getopts "w:a:" o; do case $o in w) WL=$OPTARG;; a) ARGS=$OPTARG;; esac; done; shift $((OPTIND-1))
run() { env $1 timeout -k 10 300 python bench.py --workload $WL --steps 100 --warmup 10 --no-cpu-baseline --no-other-configs --no-incoherent-extra --no-whole-model --no-calibration $ARGS 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; bk=r.get('by_launch_kind') or {}
        print('%-50s %7.1f tok/s %7.4f ms frac %.4f | ' % ('[$1]', d['value'], d['ms_per_step'], r['frac']) + '  '.join('%s %.2f' % (k, v['us_per_launch']) for k, v in bk.items() if isinstance(v, dict)))"; }
for pass in 1 2 3; do for v in "$@"; do run "$v"; done; done
