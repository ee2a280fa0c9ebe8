#!/bin/bash
# re-measure the k / v entries of the wide-codebook quantizers in tensor-core packing (the 12 entries slower than the 4090 table)
rm -f gpurun_out/lat_slow12.jsonl
python perf/latency_table.py --out gpurun_out/lat_slow12.jsonl --only '^(k|v)_ldlq_(1_5|1_6|1_7|1_8|2_10|2_11|2_12)_none_1.0_False$' > /dev/null 2>&1
python - <<'PY'
import json
for l in open('gpurun_out/lat_slow12.jsonl'):
    d=json.loads(l); print('%-30s %.2f us' % (d['key'], d['seconds']*1e6))
PY
