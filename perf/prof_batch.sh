#!/bin/bash
# per-kernel times of one batched step: bash perf/prof_batch.sh <batch> <tag>
n=${1:-64}; tag=${2:-rXX}; out=$GRAFT_REPO_ROOT/gpurun_out/prof_batch_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/kt -o kt --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --batch $n --steps 10 --warmup 2 --no-cpu-baseline --no-incoherent-extra --no-kind-breakdown --no-whole-model > $out/bench.json 2>$out/err.txt
cp $(find $out/kt -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv 2>/dev/null
rm -rf $out/kt
head -12 $out/kernel_stats.csv | cut -c1-260
