#!/usr/bin/env python3
"""profiles/traffic.json from a PMC summary (perf/pmc_summary.py output of the FETCH_SIZE pass of perf/profile_round.sh):
    python perf/make_traffic_json.py gpurun_out/prof_r05/pmc_traffic.txt profiles/r05_pmc_traffic.txt
FETCH_SIZE is in KiB per launch and tallies 128-byte requests as 64 bytes on gfx950 (MI355X_MICROARCH.md, HBM): x 2.  The figure
bench.py quotes is the mean over the token's GEMV launches; it carries the library version it was measured on."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
src, committed_as = sys.argv[1], sys.argv[2]
text = open(src).read()
tot, n, kinds = 0.0, 0, []
for m in re.finditer(r"\('([^']*)', '(\d+)'\)\s*\n\s*FETCH_SIZE\s+n=\s*(\d+)\s+mean=\s*([\d.]+)", text):
    name, grid, cnt, mean = m.group(1), int(m.group(2)), int(m.group(3)), float(m.group(4))
    tot += cnt * mean * 1024 * 2
    n += cnt
    kinds.append(f"{name[-40:]} grid {grid // 1024}: {mean * 2048 / 1e6:.2f} MB x {cnt}")
import qpalette_amd as qp
ver = qp._native.lib().qpal_version()
out = {"llama3.1-8b_tcomb_6_7": round(tot / n), "_qpal_version": ver,
       "_source": f"{committed_as}: rocprofv3 --pmc FETCH_SIZE pass of `bench.py --steps 20` (perf/profile_round.sh), mean FETCH_SIZE (KiB) per launch x 2 "
                  f"(gfx950 tallies 128-B requests as 64 B), averaged over the token's GEMV launches ({'; '.join(kinds)}). A committed measurement of library "
                  f"version {ver} (QPAL_VERSION), NOT measured by the bench run that quotes it: bench.py reports null when the loaded library's version differs."}
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print(out["llama3.1-8b_tcomb_6_7"], "bytes per launch;", n, "launches")
