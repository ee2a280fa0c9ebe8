for wl in llama3.1-70b_tcq_6 llama3.1-8b_figure1c llama3.1-8b_mem3p25 llama3.1-8b_figure1d llama3.1-8b_tcq_6 llama3.1-8b_ldlq_1_4; do
  for lib in q-palette_amd/libqpal_hip_r5base.so q-palette_amd/libqpal_hip.so; do
    for span in 1 0; do
      if [ $lib = q-palette_amd/libqpal_hip_r5base.so ] && [ $span = 0 ]; then continue; fi
      QPAL_SPAN=$span QPAL_LIB=$lib timeout -k 10 300 python bench.py --workload $wl --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs --no-incoherent-extra --no-whole-model --no-calibration --no-kind-breakdown 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%-24s %-40s span %s %8.1f tok/s %8.4f ms frac %.4f launches %d' % ('$wl', '$lib', '$span', d['value'], d['ms_per_step'], d['roofline']['frac'], d['config']['launches_per_token']))"
    done
  done
done
