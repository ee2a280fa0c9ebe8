#!/bin/bash
# Everything DESIGN.md §5 quotes: bash perf/final_round.sh r03 [sections]   (outputs under gpurun_out/final_<tag>/)
# sections (default all; one gpurun call has a 20-minute limit: the latency table alone takes ~15): workloads latency decode bench
tag=${1:-rXX}; shift; sections=${@:-workloads latency decode bench}; out=gpurun_out/final_$tag; mkdir -p $out
has() { [[ " $sections " == *" $1 "* ]]; }
if has workloads; then
rm -f $out/workloads.txt
B="python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-other-configs --no-incoherent-extra --no-kind-breakdown --no-whole-model --no-calibration"
line() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1', round(d['value'],1), d['unit'], round(d['ms_per_step'],4), 'ms frac', round(d['roofline']['frac'],4), 'launches', d['config'].get('launches_per_token'))"; }
for w in llama3.1-8b_mem3p25 llama3.1-8b_figure1c llama3.1-8b_figure1d llama3.1-8b_tcq_6 llama3.1-8b_ldlq_1_4 llama3.1-70b_tcq_6; do
  timeout -k 10 300 $B --workload $w 2>/dev/null | line "$w multi" >> $out/workloads.txt
done
timeout -k 10 300 $B --workload llama3.1-8b_figure1c --packing mi355x 2>/dev/null | line "llama3.1-8b_figure1c multi packing=mi355x" >> $out/workloads.txt
timeout -k 10 300 $B --launch single 2>/dev/null | line "llama3.1-8b_tcomb_6_7 single" >> $out/workloads.txt
for n in 8 16 17 32 64 65 128 129 256; do timeout -k 10 300 $B --batch $n 2>/dev/null | line "llama3.1-8b_tcomb_6_7 batch $n" >> $out/workloads.txt; done
echo "workloads done"; cat $out/workloads.txt
fi
if has latency; then
rm -f $out/lat_p0.jsonl $out/lat_p1.jsonl
timeout -k 10 500 python perf/latency_table.py --out $out/lat_p0.jsonl --part 0 --parts 2 > $out/lat_p0.log 2>&1
timeout -k 10 500 python perf/latency_table.py --out $out/lat_p1.jsonl --part 1 --parts 2 > $out/lat_p1.log 2>&1
echo "latency table done"
fi
if has decode; then
rm -f $out/decode_tcomb67_contexts.json $out/attention_probe.txt
timeout -k 10 300 python perf/decode_llama.py --tokens 128 2>/dev/null | tail -1 > $out/decode_tcomb67.json
for c in "8192 8" "8192 2000" "8192 7900" "32768 32000"; do set -- $c; timeout -k 10 300 python perf/decode_llama.py --tokens 64 --context $1 --start-pos $2 --no-modular 2>/dev/null | tail -1 >> $out/decode_tcomb67_contexts.json; done
timeout -k 10 400 python perf/decode_llama.py --model 3_70b --no-modular --tokens 32 --quantizer tcq_6_none_0.9 2>/dev/null | tail -1 > $out/decode_70b_tcq6.json
timeout -k 10 400 python perf/decode_llama.py --model 3_70b --no-modular --tokens 32 --quantizer tcq_6_none_0.9 --context 8192 --start-pos 8000 2>/dev/null | tail -1 >> $out/decode_70b_tcq6.json
for a in "2048 2000" "4096 4000" "8192 600" "8192 8000" "32768 32000" "8192 8000 64 8 128"; do timeout -k 10 100 python perf/probe_attn.py $a 2>/dev/null >> $out/attention_probe.txt; done
timeout -k 10 300 python perf/decode_llama.py --tokens 64 --qdict figure1d 2>/dev/null | tail -1 > $out/decode_figure1d.json
echo "decode done"
fi
if has bench; then
timeout -k 10 600 python bench.py > $out/bench_default.json 2>$out/bench_default.err
tail -c 600 $out/bench_default.json
fi
