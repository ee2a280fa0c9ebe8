#!/usr/bin/env python3
"""Finish the MI355X latency table: merge the parts perf/latency_table.py wrote, add the `constant` term and
emit the file the reference's solver loads.

The reference's fusion-aware MSQ solver models a decoded token as
    latency = constant + sum over layers and launches of lat_coeff[f"{layer}_{quantizer_str}_{simt}"]
(solve_lat_const.py:113-123; `lat_coeff_dict['constant'].item()`, loaded by lat_coeff_routine, l.219-221, from
assets/{model_key}_latency_coeffs_{nodename}.pt).  `constant` is everything of a token that is not a quantized linear.  Here it is
MEASURED with perf/decode_llama.py (whole decode step of a Llama-3.1-8B-shaped model under one HIP graph, context 1024):
    constant = ms_fused_glue - 32 * (qkv + o + ug + d entries of the quantizer that run used)
so that the solver's formula reproduces the measured step for that (merged) model: rotations, norms, attention, lm_head.

    python perf/latency_finish.py --parts gpurun_out/lat_part*.jsonl --decode gpurun_out/decode_tcomb67.json
      -> perf/latency/3_8b_latency_coeffs_mi355x.json  and  perf/latency/3_8b_latency_coeffs_mi355x.pt
    (nodename "mi355x": copy the .pt to assets/ of a Q-Palette checkout and run solve_lat_const.py --nodename mi355x)
"""
import argparse
import glob
import json
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
UNFUSED = ["q", "k", "v", "o", "g", "u", "d"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--parts", nargs="*", default=[])
    ap.add_argument("--decode", help="JSON line printed by perf/decode_llama.py (uniform quantizer run)")
    ap.add_argument("--table", default=os.path.join(HERE, "latency", "3_8b_latency_coeffs_mi355x.json"))
    args = ap.parse_args()
    with open(args.table) as f:
        table = json.load(f)
    n_new = 0
    for pattern in args.parts:
        for path in glob.glob(pattern):
            for line in open(path):
                rec = json.loads(line)
                table[rec["key"]] = rec["seconds"]
                n_new += 1
    if args.decode:
        with open(args.decode) as f:
            dec = json.loads([l for l in f if l.startswith("{")][-1])
        q = dec["quantizer"]
        # The fused-glue step (perf/decode_llama.py: RMSNorm / rotation inside the GEMV launches, one attention launch) is what a
        # MI355X decode runs; it launches q|k|v and up|gate as ONE launch each, so the matching table entries are the merged ones.
        if dec.get("ms_fused_glue"):
            linears = dec["layers"] * sum(table[f"{lk}_{q}_False"] for lk in ["qkv", "o", "ug", "d"])
            step, what = dec["ms_fused_glue"], "fused-glue step"
        else:
            linears = dec["layers"] * sum(table[f"{lk}_{q}_False"] for lk in UNFUSED)
            step, what = dec["ms_whole_step"], "modular step (torch glue)"
        const = max(0.0, step * 1e-3 - linears)
        table["constant"] = const
        table["_constant_doc"] = (f"seconds; measured: perf/decode_llama.py {what} {step:.3f} ms (model {dec['model']}, "
                                  f"{dec['layers']} layers, {q}, context {dec['context']}) minus {dec['layers']} x the table's "
                                  f"{'qkv,o,ug,d' if dec.get('ms_fused_glue') else 'q,k,v,o,g,u,d'} entries ({linears * 1e3:.3f} ms); the same "
                                  f"run's modular step (Incoherent* modules + ~1700 torch glue launches): {dec['ms_whole_step']:.3f} ms")
    table["_n"] = sum(1 for k in table if not k.startswith("_"))
    with open(args.table, "w") as f:
        json.dump(table, f, indent=0, sort_keys=True)
    # the reference's format: {key: float seconds ..., 'constant': 0-d tensor}
    pt = {k: float(v) for k, v in table.items() if not k.startswith("_") and k != "constant"}
    if "constant" in table:
        pt["constant"] = torch.tensor(float(table["constant"]))
    out = os.path.splitext(args.table)[0] + ".pt"
    torch.save(pt, out)
    print(f"{args.table}: {table['_n']} entries ({n_new} updated), constant = {table.get('constant')}; wrote {out}")


if __name__ == "__main__":
    main()
