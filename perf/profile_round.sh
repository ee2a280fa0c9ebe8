#!/bin/bash
# Round profile set (run on the GPU box from the repo root): bash perf/profile_round.sh r02
# kernel-trace stats, PMC traffic and SQ counters in separate rocprofv3 passes (no tracing beside counters), in-kernel stamps.
tag=${1:-rXX}; out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs --no-incoherent-extra --no-kind-breakdown --no-whole-model --no-calibration"
$B > $out/bench.json 2>/dev/null
rocprofv3 --kernel-trace --stats -d $out/kt -o kt --output-format csv -- $B > $out/kt_bench.json 2>$out/kt.err
cp $(find $out/kt -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv 2>/dev/null
rocprofv3 --pmc FETCH_SIZE -d $out/p1 -o pmc --output-format csv -- $B > /dev/null 2>$out/p1.err
python3 $GRAFT_REPO_ROOT/perf/pmc_summary.py $out/p1 tc_gemv > $out/pmc_traffic.txt
rocprofv3 --pmc WRITE_SIZE -d $out/p2 -o pmc --output-format csv -- $B > /dev/null 2>$out/p2.err
python3 $GRAFT_REPO_ROOT/perf/pmc_summary.py $out/p2 tc_gemv >> $out/pmc_traffic.txt
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD"; do
  rocprofv3 --pmc $set -d $out/p3 -o pmc --output-format csv -- $B --layers 8 > /dev/null 2>$out/p3.err
  python3 $GRAFT_REPO_ROOT/perf/pmc_summary.py $out/p3 tc_gemv >> $out/pmc_sq.txt
  rm -rf $out/p3
done
rm -rf $out/p1 $out/p2 $out/kt
cd $GRAFT_REPO_ROOT
QPAL_LIB=q-palette_amd/libqpal_hip_stamps.so python3 bench.py --steps 2 --warmup 1 --layers 2 --no-graph --no-cpu-baseline --no-other-configs --no-incoherent-extra --no-kind-breakdown --no-whole-model 2>/dev/null | grep "^\[stamps\]" > $out/inkernel_stamps.txt
ls -la $out
