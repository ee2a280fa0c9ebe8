#!/bin/bash
# SQ counters of one library variant's chain kernel: bash perf/pmc_chain.sh <lib suffix> <out dir>
# (counters in their own rocprofv3 passes, no tracing beside them)
lib=q-palette_amd/libqpal_hip$1.so; out=$2; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export QPAL_LIB=$GRAFT_REPO_ROOT/$lib
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_INSTS_SALU SQ_ACTIVE_INST_SCA"; do
  rocprofv3 --pmc $set -d $GRAFT_REPO_ROOT/$out/p -o pmc --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --layers 8 --no-cpu-baseline --no-incoherent-extra --launch ${3:-chain} > $GRAFT_REPO_ROOT/$out/log.txt 2>&1
  python3 $GRAFT_REPO_ROOT/perf/pmc_summary.py $GRAFT_REPO_ROOT/$out/p ${4:-tc_chain} >> $GRAFT_REPO_ROOT/$out/summary.txt
  rm -rf $GRAFT_REPO_ROOT/$out/p
done
cat $GRAFT_REPO_ROOT/$out/summary.txt
