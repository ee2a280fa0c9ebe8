#!/bin/bash
# Shader clock, power and temperature of the card while the headline token replays back to back (rocm-smi sampled every 0.2 s beside a
# 3000-step bench run), then while the decode-floor / stream calibration kernels run:  bash perf/clock_trace.sh -> gpurun_out/clock_trace.txt
out=${1:-gpurun_out/clock_trace.txt}; mkdir -p gpurun_out; : > $out
sample() { while true; do echo "$(date +%s.%N | cut -c1-14) $(rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E 'sclk|mclk|Power|Temperature \(Sensor (edge|junction|hotspot)' | sed -E 's/^GPU\[[0-9]+\]\s*:\s*//' | tr '\n' '|')" >> $out; sleep 0.2; done; }
rocm-smi --showclocks --showpower --showtemp --showmaxpower 2>&1 | head -40 >> $out
echo "== sampling (idle 2 s, then bench --steps 3000)" >> $out
sample & SP=$!
sleep 2
python bench.py --steps 3000 --warmup 20 --no-cpu-baseline --no-other-configs --no-incoherent-extra --no-whole-model --no-kind-breakdown 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('BENCH 3000 steps: %.1f tok/s %.4f ms' % (d['value'], d['ms_per_step']))" >> $out
sleep 1
kill $SP
grep -E "BENCH|Mhz" $out | sed -E "s/.*sclk clock level: [0-9S]+: \(([0-9]+)Mhz\).*Power \(W\): *([0-9.]*).*/\1 MHz \2 W/" | tail -30
