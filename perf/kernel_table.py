#!/usr/bin/env python3
"""Per-kernel table (name, grid, calls, avg / min us) of a rocprofv3 --kernel-trace results .db: this library's kernels.

    python perf/kernel_table.py <results.db> <steps> "<header line>"
"""
import sqlite3
import sys


def main():
    db, steps, head = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    cur = sqlite3.connect(db).cursor()
    print(f"# {head}")
    print("# kernel | grid threads | workgroup | calls | avg us | min us")
    q = ("select name, grid_x, workgroup_x, count(*), avg(end-start)/1000.0, min(end-start)/1000.0, sum(end-start)/1000.0 "
         "from kernels where name like '%qpal::%' group by name, grid_x order by 7 desc")
    tot = 0.0
    for r in cur.execute(q):
        print(f"{r[0][:110]} | {r[1]} | {r[2]} | {r[3]} | {r[4]:.2f} | {r[5]:.2f}")
        tot += r[6]
    print(f"# sum of these kernels per step: {tot / steps / 1000:.3f} ms")


if __name__ == "__main__":
    main()
