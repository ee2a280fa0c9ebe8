#!/usr/bin/env python3
"""Offline view of the raw in-kernel stamps a STAMPS build dumps (QPAL_STAMPS_DUMP=<prefix>): when does workgroup b start and end,
as a function of its block index?  python perf/stamps_by_block.py <file.bin> ..."""
import sys

import numpy as np

for path in sys.argv[1:]:
    a = np.fromfile(path, dtype=np.uint64).reshape(-1, 16, 8).astype(np.float64) / 100.0  # [block][wave][stamp] us
    live = a[:, :, 0] > 0
    t0 = a[:, :, 0][live].min()
    start = np.where(live, a[:, :, 0], np.nan)
    first_w = np.where(live, a[:, :, 1], np.nan)
    steps_end = np.where(live, a[:, :, 4], np.nan)
    end = np.where(live, a[:, :, 7], np.nan)
    nb = a.shape[0]
    print(f"== {path}: {nb} workgroups; start / steps-begin / steps-end / end (us after the first wave start), mean over the waves of a workgroup, by block index")
    for lo in range(0, nb, max(1, nb // 16)):
        hi = min(nb, lo + max(1, nb // 16))
        sl = slice(lo, hi)
        print(f"  blocks {lo:3d}-{hi - 1:3d}: start {np.nanmean(start[sl]) - t0:5.2f}  first-weights {np.nanmean(first_w[sl]) - t0:5.2f}  "
              f"steps-end {np.nanmean(steps_end[sl]) - t0:5.2f} (max {np.nanmax(steps_end[sl]) - t0:5.2f})  end {np.nanmax(end[sl]) - t0:5.2f}")
    if (a[:, :, 2][live] > 0).all():  # early-staging launches of a round-4 STAMPS build: slot 2 = wave entry, slot 3 = early loads landed
        entry, landed = np.where(live, a[:, :, 2], np.nan), np.where(live, a[:, :, 3], np.nan)
        e0 = np.nanmin(entry)
        print(f"  per wave, us: entry -> kernel arguments in hand (stamp 0) {np.nanmean(start - entry):.2f} (max {np.nanmax(start - entry):.2f}); entry -> early x / table "
              f"loads landed {np.nanmean(landed - entry):.2f} (max {np.nanmax(landed - entry):.2f}); landed -> staged + barrier (stamp 1) {np.nanmean(first_w - landed):.2f}; "
              f"wave entries spread over {np.nanmax(entry) - e0:.2f} (inside a workgroup {np.nanmean(np.nanmax(entry, axis=1) - np.nanmin(entry, axis=1)):.2f}); "
              f"first entry -> last end {np.nanmax(end) - e0:.2f}")
    per_block_start = np.nanmin(start, axis=1) - t0
    order = np.argsort(per_block_start)
    r = np.corrcoef(np.arange(nb), per_block_start)[0, 1]
    print(f"  start time vs block index: correlation {r:.3f}; by block%8 (XCD): " +
          " ".join(f"{np.mean(per_block_start[x::8]):.2f}" for x in range(8)))
    print(f"  earliest blocks {order[:8].tolist()}, latest {order[-8:].tolist()}")
