#!/usr/bin/env python3
"""MI355X counterpart of the reference's assets/3_8b_latency_coeffs_4090_cc.pt: per-kernel batch-1 latency of
every {q,k,v,o,u,g,d,qk,kv,qv,qkv,ug} projection x quantizer x {tensor-core-order, SIMT} packing
(consumed by solve_lat_const.py:113-123 as `{layer}_{quantizer_str}_{True|False}` -> seconds).

Method: the layer's fused decode+GEMV op, HIP-graph replay over enough distinct weight buffers to exceed the
256 MB Infinity Cache, HIP events on the launch stream; seconds per launch.

    python perf/latency_table.py --out gpurun_out/lat_part0.jsonl --part 0 --parts 3
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHAPES = {"q": (4096, 4096), "k": (1024, 4096), "v": (1024, 4096), "o": (4096, 4096), "u": (14336, 4096),
          "g": (14336, 4096), "d": (4096, 14336), "qk": (5120, 4096), "kv": (2048, 4096), "qv": (5120, 4096),
          "qkv": (6144, 4096), "ug": (28672, 4096)}
QUANTIZERS = ([f"tcq_{kv}_none_0.9" for kv in range(3, 11)] +
              [f"tcomb_{kv}_{kv + 1}_0.5_none_0.9" for kv in range(3, 10)] +
              [f"ldlq_2_{b}_none_1.0" for b in range(3, 13)] + [f"ldlq_1_{b}_none_1.0" for b in range(2, 9)])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--part", type=int, default=0)
    ap.add_argument("--parts", type=int, default=1)
    ap.add_argument("--iters", type=int, default=4)
    ap.add_argument("--only", default=None, help="regular expression on the key (re-measure a few entries)")
    args = ap.parse_args()
    import torch
    import qpalette_amd as qp

    dev = torch.device("cuda", 0)
    keys = [(lk, q, simt) for q in QUANTIZERS for lk in SHAPES for simt in ((False, True) if q.startswith("ldlq") else (False,))]
    if args.only:
        import re
        keys = [k for k in keys if re.search(args.only, f"{k[0]}_{k[1]}_{k[2]}")]
    keys = keys[args.part::args.parts]
    done = set()
    if os.path.exists(args.out):
        done = {json.loads(l)["key"] for l in open(args.out)}
    s = torch.cuda.Stream(dev)
    t_start = time.time()
    with open(args.out, "a") as f:
        for i, (lk, qstr, simt) in enumerate(keys):
            key = f"{lk}_{qstr}_{simt}"
            if key in done:
                continue
            m, k = SHAPES[lk]
            info0 = qp.mem_op.dummy_linear_info(k, m, qstr, seed=0, device=dev, codebook_seed=7)
            nbytes = qp.mem_op.packed_bytes(info0)
            copies = max(2, int(400e6 // nbytes) + 1)
            mods = []
            for c in range(copies):
                info = info0 if c == 0 else qp.mem_op.dummy_linear_info(k, m, qstr, seed=c, device=dev, codebook_seed=7)
                mods.append(qp.VQLinearPackSIMT.gen_layer_from_info(info, device=dev) if simt
                            else qp.make_linear_from_info(qstr, info).to(dev))
            qp.share_codebooks(mods)
            x = torch.randn(1, k, device=dev).half()
            with torch.cuda.stream(s):
                for mod in mods:
                    mod._gemv(x, 1)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=s):
                    for mod in mods:
                        mod._gemv(x, 1)
                g.replay()
                torch.cuda.synchronize()
                best = float("inf")  # three batches, the fastest counts (one entry of a full pass came back 13x slow once)
                for _ in range(3):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(s)
                    for _ in range(args.iters):
                        g.replay()
                    e1.record(s)
                    torch.cuda.synchronize()
                    best = min(best, e0.elapsed_time(e1))
            sec = best * 1e-3 / (args.iters * copies)
            f.write(json.dumps({"key": key, "seconds": sec, "MB": nbytes / 1e6, "GBps": nbytes / sec / 1e9}) + "\n")
            f.flush()
            del mods, g
            if i % 10 == 0:
                print(f"[{i}/{len(keys)}] {key}: {sec * 1e6:.2f} us  ({time.time() - t_start:.0f} s)", flush=True)


if __name__ == "__main__":
    main()
