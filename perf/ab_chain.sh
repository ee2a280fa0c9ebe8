#!/bin/bash
# A/B of library variants (make -C q-palette_amd/csrc VARIANT=<name> EXTRA="-D..."): bash perf/ab_chain.sh "" _<name> ...
run() { QPAL_LIB=$1 timeout -k 10 200 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-incoherent-extra $2 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1 $2', round(d['value'],1), 'tok/s', round(d['ms_per_step'],4), 'ms frac', round(d['roofline']['frac'],4))"; }
for v in "$@"; do run q-palette_amd/libqpal_hip$v.so "--launch chain"; done
run q-palette_amd/libqpal_hip.so "--launch multi"
