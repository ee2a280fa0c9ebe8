#!/usr/bin/env python3
"""Per-kernel microbench: one quantized linear shape, many distinct weight buffers (> Infinity Cache),
eager back-to-back launches timed with HIP events on the launch stream.

    python perf/microbench.py --qstr tcomb_6_7_0.5_none_0.9 --m 14336 --k 4096 --copies 24 --iters 20
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--qstr", default="tcomb_6_7_0.5_none_0.9")
    ap.add_argument("--m", type=int, default=14336)
    ap.add_argument("--k", type=int, default=4096)
    ap.add_argument("--n", type=int, default=1)
    ap.add_argument("--copies", type=int, default=0, help="distinct weight buffers (0: enough for 600 MB)")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--simt", action="store_true")
    ap.add_argument("--dequant", action="store_true")
    ap.add_argument("--graph", action="store_true")
    args = ap.parse_args()
    import torch
    import qpalette_amd as qp

    dev = torch.device("cuda", 0)
    info0 = qp.mem_op.dummy_linear_info(args.k, args.m, args.qstr, seed=0, device=dev)
    nbytes = qp.mem_op.packed_bytes(info0)
    copies = args.copies or max(2, int(600e6 // nbytes) + 1)
    mods = []
    for c in range(copies):
        info = qp.mem_op.dummy_linear_info(args.k, args.m, args.qstr, seed=c, device=dev)
        if args.simt:
            mods.append(qp.VQLinearPackSIMT.gen_layer_from_info(info, device=dev))
        else:
            mods.append(qp.make_linear_from_info(args.qstr, info).to(dev))
    x = torch.randn(args.n, args.k, device=dev).half()
    s = torch.cuda.Stream(dev)

    def sweep():
        for mod in mods:
            if args.dequant:
                mod.get_weight()
            else:
                mod._gemv(x, args.n)

    with torch.cuda.stream(s):
        sweep()
        torch.cuda.synchronize()
        run = sweep
        if args.graph:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                sweep()
            run = g.replay
        run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(args.iters):
            run()
        e1.record(s)
        torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (args.iters * copies)
    alg = nbytes + args.n * args.k * 2 + args.n * args.m * 4
    if args.dequant:
        alg = nbytes + args.m * args.k * 2
    print(json.dumps({"qstr": args.qstr, "m": args.m, "k": args.k, "n": args.n, "copies": copies,
                      "us_per_launch": round(us, 3), "alg_MB": round(alg / 1e6, 3),
                      "GBps": round(alg / us / 1e3, 1), "frac_of_8TBps": round(alg / us / 1e3 / 8000, 4),
                      "mode": "dequant" if args.dequant else ("simt" if args.simt else "gemv"),
                      "graph": args.graph}))


if __name__ == "__main__":
    main()
