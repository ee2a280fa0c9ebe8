#!/bin/bash
# Same-box A/B of library variants (make -C q-palette_amd/csrc VARIANT=<name> EXTRA="-D..." -> libqpal_hip_<name>.so):
#   bash perf/ab.sh [-w workload] [-a "extra bench args"] "" _base _ring3 ...      ("" = the default library)
# One line per variant: tokens/s, ms/token, roofline fraction and us per launch kind.  Runs every variant TWICE, interleaved,
# so that drift of the box (clock, neighbours) shows up as disagreement between the two passes.
WL=llama3.1-8b_tcomb_6_7; ARGS=""
while getopts "w:a:" o; do case $o in w) WL=$OPTARG;; a) ARGS=$OPTARG;; esac; done; shift $((OPTIND-1))
run() { QPAL_LIB=q-palette_amd/libqpal_hip$1.so timeout -k 10 300 python bench.py --workload $WL --steps 100 --warmup 10 --no-cpu-baseline --no-other-configs --no-incoherent-extra $ARGS 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; bk=r.get('by_launch_kind') or {}
        print('%-12s %7.1f tok/s %7.4f ms frac %.4f | ' % ('lib$1', d['value'], d['ms_per_step'], r['frac']) + '  '.join('%s %.2f' % (k, v['us_per_launch']) for k, v in bk.items() if isinstance(v, dict)))"; }
for pass in 1 2; do for v in "$@"; do run "$v"; done; done
