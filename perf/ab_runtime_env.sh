#!/bin/bash
# Same-box A/B of HIP-RUNTIME switches (the clr flags the runtime library reads at start-up) on (i) the cost of an empty dependent launch in a
# replayed graph (perf/launch_floor.bin) and (ii) the headline token:  bash perf/ab_runtime_env.sh  → gpurun_out/ab_runtime_env.txt
out=gpurun_out/ab_runtime_env.txt; mkdir -p gpurun_out; : > $out
SETS=("BASE=1" "AMD_OPT_FLUSH=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "DEBUG_HIP_GRAPH_BATCH_SIZE=512" "ROC_SYSTEM_SCOPE_SIGNAL=0" "GPU_FLUSH_ON_EXECUTION=0" "DEBUG_HIP_KERNARG_COPY_OPT=0" "ROC_USE_FGS_KERNARG=0")
echo "== empty dependent launch under graph replay (perf/launch_floor.bin), by setting" >> $out
for v in "${SETS[@]}"; do echo "[$v]" >> $out; env $v timeout -k 10 60 perf/launch_floor.bin >> $out 2>&1 || echo "  failed" >> $out; done
echo "== headline token (3 interleaved passes)" >> $out
bash perf/ab_env.sh "${SETS[@]}" >> $out 2>&1
cat $out
