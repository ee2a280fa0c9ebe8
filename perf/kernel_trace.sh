#!/bin/bash
# Per-kernel time of one bench workload (rocprofv3 --kernel-trace --stats; run on the GPU box from the repo root):
#   bash perf/kernel_trace.sh <workload> <outfile> [extra bench args]
wl=$1; out=$GRAFT_REPO_ROOT/$2; shift 2
d=$(mktemp -d /tmp/kt.XXXX); cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $d -o kt --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs --no-incoherent-extra --no-kind-breakdown --no-whole-model --no-calibration "$@" > $d/bench.json 2> $d/err.txt
f=$(find $d -name "*kernel_stats.csv" | head -1)
python3 - "$f" "$d/bench.json" > $out <<'PY'
import csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
try:
    d = json.loads(open(sys.argv[2]).read().strip().split("\n")[-1])
    print("# %s: %.1f tok/s, %.4f ms/token under tracing, %d launches/token" % (d["config"]["workload"][:60], d["value"], d["ms_per_step"], d["config"]["launches_per_token"]))
    steps = d["steps"] + d["warmup"] + 1
except Exception as e:
    print("# bench line unreadable:", e); steps = 24
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("# kernel | calls | calls/token | avg us | us/token | share")
for r in rows[:16]:
    calls = int(r["Calls"])
    print("%-120s %6d %7.1f %8.2f %9.1f %5.1f%%" % (r["Name"][:120].replace("\n", " "), calls, calls / steps, float(r["AverageNs"]) / 1e3,
                                                   float(r["TotalDurationNs"]) / 1e3 / steps, 100 * float(r["TotalDurationNs"]) / tot))
PY
cd $GRAFT_REPO_ROOT; rm -rf $d
