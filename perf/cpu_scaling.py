#!/usr/bin/env python3
"""How the CPU baseline (oracle port, OpenMP) scales on this host: seconds per 4096 x 14336 tcomb_6_7 linear at 1 .. N threads,
beside what the box says about the CPUs a job may use.  python perf/cpu_scaling.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
import qpalette_amd as qp
from oracle import oracle


def main():
    print("usable_cpus():", bench.usable_cpus(), "| os.cpu_count:", os.cpu_count(), "| omp max threads:", oracle.num_threads())
    for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
        try:
            print(p, "=", open(p).read().strip())
        except OSError as e:
            print(p, "->", e.strerror)
    k, m = 4096, 14336
    info = qp.mem_op.dummy_linear_info(k, m, "tcomb_6_7_0.5_none_0.9", seed=1, device="cpu")
    x = np.random.default_rng(0).standard_normal((1, k)).astype(np.float16)
    scratch = np.empty((m, k), dtype=np.uint16)
    args = (info["trellis1"].numpy(), info["trellis2"].numpy(), info["tlut"].numpy(), x, m, 1, k, info["tlut_bits"], info["KV"][0], info["KV"][1], 2)
    nmax = oracle.num_threads()
    base = None
    for nt in [1, 2, 4, 8, 16, 32, 64, 128, 256]:
        if nt > nmax:
            break
        oracle.set_num_threads(nt)
        oracle.cpu_tcq_linear(*args, scratch)
        ts = []
        for _ in range(3 if nt < 4 else 7):
            t0 = time.perf_counter()
            oracle.cpu_tcq_linear(*args, scratch)
            ts.append(time.perf_counter() - t0)
        t = float(np.median(ts))
        base = base or t
        print(f"threads {nt:4d}: {t * 1e3:9.2f} ms per linear (materialise)  speedup {base / t:6.2f}  efficiency {base / t / nt:5.2f}")
    oracle.set_num_threads(nmax)


if __name__ == "__main__":
    main()
