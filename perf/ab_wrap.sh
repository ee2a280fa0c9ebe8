#!/bin/bash
# Same-box A/B of library variants on the three figures of the default bench line: plain token, token behind the incoherence
# wrapper, whole-model decode step.   bash perf/ab_wrap.sh "" _variant ...
run() { QPAL_LIB=q-palette_amd/libqpal_hip$1.so timeout -k 10 400 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-kind-breakdown --no-calibration 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); w=d.get('with_incoherence_wrapper',{}); m=d.get('whole_model_decode',{})
        print('%-12s plain %7.1f tok/s %.4f ms | wrapper %7.1f tok/s %.4f ms | whole model %7.1f tok/s %.4f ms' % ('lib$1', d['value'], d['ms_per_step'], w.get('value',0), w.get('ms_per_step',0), m.get('value',0), m.get('ms_per_step',0)))"; }
for pass in 1 2; do for v in "$@"; do run "$v"; done; done
