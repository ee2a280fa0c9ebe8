#!/bin/bash
# per-kernel times of one batched step, by grid size (= launch kind): bash perf/prof_batch.sh <batch> <tag>
n=${1:-64}; tag=${2:-rXX}; out=$GRAFT_REPO_ROOT/gpurun_out/prof_batch_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/kt -o kt --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --batch $n --steps 10 --warmup 2 --no-cpu-baseline --no-incoherent-extra --no-kind-breakdown --no-whole-model > $out/bench.json 2>$out/err.txt
cp $(find $out/kt -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv 2>/dev/null
python3 - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(list)
for path in glob.glob(out + "/kt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "tc_gem" not in r["Kernel_Name"]:
            continue
        g = int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0)
        acc[(r["Kernel_Name"].split("(")[0][-60:], g)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open(out + "/by_grid.txt", "w") as f:
    for (k, g), v in sorted(acc.items(), key=lambda kv: kv[0][1]):
        line = f"{k} grid {g} ({g // 512 if g % 512 == 0 else g} workgroups): n {len(v)} mean {sum(v) / len(v) / 1e3:.2f} us min {min(v) / 1e3:.2f} max {max(v) / 1e3:.2f}"
        print(line); f.write(line + "\n")
PY
rm -rf $out/kt
