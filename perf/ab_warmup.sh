#!/bin/bash
# Does the length of the untimed warm-up change a 20-step measurement?  (The driver runs `bench.py --steps 20 --warmup 5`; a default run times 200
# steps after 20.)  Same box, two interleaved passes:  bash perf/ab_warmup.sh  -> gpurun_out/ab_warmup.txt
out=gpurun_out/ab_warmup.txt; mkdir -p gpurun_out; : > $out
B="python bench.py --steps 20 --no-cpu-baseline --no-other-configs --no-incoherent-extra --no-whole-model --no-calibration --no-kind-breakdown"
for pass in 1 2; do for w in 5 20 50 100 200 400; do
  timeout -k 10 200 $B --warmup $w 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('steps 20 warmup %4d: %7.1f tok/s %.4f ms frac %.4f' % ($w, d['value'], d['ms_per_step'], d['roofline']['frac']))" >> $out
done; done
timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-other-configs --no-incoherent-extra --no-whole-model --no-calibration --no-kind-breakdown 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('steps 200 warmup 20: %7.1f tok/s %.4f ms frac %.4f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))" >> $out
cat $out
