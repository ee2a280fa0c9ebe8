#!/bin/bash
# SQ counters of the batched kernels (separate rocprofv3 passes, counters only): bash perf/prof_batch_pmc.sh <batch> <tag> [workload]
n=${1:-64}; tag=${2:-rXX}; wl=${3:-llama3.1-8b_tcomb_6_7}; out=$GRAFT_REPO_ROOT/gpurun_out/prof_batch_pmc_${tag}_b$n; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --workload $wl --batch $n --steps 4 --warmup 1 --layers 4 --no-cpu-baseline --no-other-configs --no-incoherent-extra --no-kind-breakdown --no-whole-model --no-calibration"
rm -f $out/pmc_sq.txt
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"; do
  rocprofv3 --pmc $set -d $out/p3 -o pmc --output-format csv -- $B > /dev/null 2>$out/p3.err
  python3 $GRAFT_REPO_ROOT/perf/pmc_summary.py $out/p3 tc_gem >> $out/pmc_sq.txt
  rm -rf $out/p3
done
cat $out/pmc_sq.txt
