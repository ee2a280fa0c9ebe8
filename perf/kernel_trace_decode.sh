#!/bin/bash
# Per-kernel time of the whole-model decode step (rocprofv3 --kernel-trace --stats; run on the GPU box from the repo root):
#   bash perf/kernel_trace_decode.sh <outfile> [decode_llama.py args]
out=$GRAFT_REPO_ROOT/$1; shift
d=$(mktemp -d /tmp/kt.XXXX); cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $d -o kt --output-format csv -- python3 $GRAFT_REPO_ROOT/perf/decode_llama.py --no-modular --tokens 64 "$@" > $d/decode.json 2> $d/err.txt
f=$(find $d -name "*kernel_stats.csv" | head -1)
python3 - "$f" "$d/decode.json" > $out <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("#", open(sys.argv[2]).read().strip().split("\n")[-1][:600])
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("# kernel | calls | avg us | total ms | share")
for r in rows[:14]:
    print("%-110s %7d %8.2f %9.2f %5.1f%%" % (r["Name"][:110].replace("\n", " "), int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
PY
cd $GRAFT_REPO_ROOT; rm -rf $d
