#!/usr/bin/env python3
"""Latency of the incoherence rotation kernel (qpal_hadamard) under HIP-graph replay: us per launch in a chain of
dependent launches, for the Llama sizes.  python perf/had_bench.py [rows]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import qpalette_amd as qp

had = qp.hadamard
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda")
CHAIN = 50


def bench(name, fn):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(CHAIN):
                fn()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(20):
            g.replay()
        e1.record(s)
        torch.cuda.synchronize()
    print(f"{name:40s} {e0.elapsed_time(e1) * 1e3 / 20 / CHAIN:8.2f} us/launch", flush=True)


for n in (4096, 8192, 14336, 28672):
    hadK, K = had.get_hadK(n)
    h = None if hadK is None else hadK.T.contiguous().half().to(dev)
    x = torch.randn(rows, n, device=dev).half()
    su = (torch.randint(0, 2, (n,), device=dev) * 2 - 1).half()
    out = torch.empty_like(x)
    bench(f"rotate f16 n={n} K={K} rows={rows}", lambda: had.rotate(x, hadK=h, K=K, su=su, post_scale=1 / 64, out=out))
    ug = torch.randn(rows, 2 * n, device=dev)
    bench(f"rotate swiglu n={n} K={K} rows={rows}",
          lambda: had.rotate(ug, hadK=h, K=K, su=su, post_scale=1 / 64, in_mode=had.IN_SWIGLU_F32, out=out))
