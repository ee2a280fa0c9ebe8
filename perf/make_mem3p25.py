#!/usr/bin/env python3
"""BASELINE.json configs[2]: "Llama-3.1-8B mem-constrained MSQ @3.25 avg bits (mixed TCQ/VQ/SQ)" as a committed fixture.

The reference produces such a qdict with an OR-Tools ILP (solve_mem_const.py:24-125; quantizer set l.4-22: tcq_3..10 and the
tcomb half-steps) from per-layer error coefficients; OR-Tools is not in this image and the result for 3.25 b is not
published, so the qdict is HAND-CONSTRUCTED here with the shape such solutions have — sensitive
projections (v, o, down, first / last blocks) above the average, the wide gate / up below — from the reference's
memory-constrained quantizer set plus the VQ / SQ (ldlq) entries that the published figure1c result uses, and then tuned so
that the parameter-weighted average is 3.25 bits/weight.  No layer fusion (the memory-constrained solver has none).
Every value is (quantizer_str, simt) like the published qdicts; a few VQ layers use the SIMT packing.

    python perf/make_mem3p25.py   ->  perf/qdicts/mem3p25.json  (prints the achieved average)
"""
import json
import os

HIDDEN, INTER, KV, NL = 4096, 14336, 1024, 32
SHAPE = {"self_attn.q_proj": HIDDEN * HIDDEN, "self_attn.k_proj": KV * HIDDEN, "self_attn.v_proj": KV * HIDDEN,
         "self_attn.o_proj": HIDDEN * HIDDEN, "mlp.gate_proj": INTER * HIDDEN, "mlp.up_proj": INTER * HIDDEN,
         "mlp.down_proj": HIDDEN * INTER}


def bits(q):
    p = q.split("_")
    if p[0] == "tcq":
        return int(p[1]) / 2
    if p[0] == "tcomb":
        return (int(p[1]) + int(p[2])) / 4
    if p[0] == "ldlq":
        return int(p[2]) / int(p[1])
    raise ValueError(q)


def build(n_wide_325):
    """n_wide_325: how many of the 64 gate/up projections (from the last block backwards) get tcomb_6_7 instead of tcq_6."""
    qd = {}
    wide = [(i, key) for i in range(NL) for key in ("mlp.gate_proj", "mlp.up_proj")]
    upgraded = set(wide[len(wide) - n_wide_325:])
    for i in range(NL):
        edge = i < 2 or i >= NL - 2
        qd[f"{i}_self_attn.q_proj"] = ("tcq_7_none_0.9" if edge else "tcq_6_none_0.9", "0")
        qd[f"{i}_self_attn.k_proj"] = ("ldlq_2_8_none_1.0", "1") if i % 8 == 3 else ("tcomb_7_8_0.5_none_0.9", "0")
        qd[f"{i}_self_attn.v_proj"] = ("ldlq_2_10_none_1.0", "0") if edge or i % 4 == 0 else ("tcq_8_none_0.9", "0")
        qd[f"{i}_self_attn.o_proj"] = ("ldlq_1_4_none_1.0", "0") if i % 8 == 5 else ("tcomb_7_8_0.5_none_0.9", "0")
        for key in ("mlp.gate_proj", "mlp.up_proj"):
            if i % 8 == 6 and key == "mlp.up_proj":
                qd[f"{i}_{key}"] = ("ldlq_2_6_none_1.0", "0")           # 3.0 b VQ
            else:
                qd[f"{i}_{key}"] = ("tcomb_6_7_0.5_none_0.9" if (i, key) in upgraded else "tcq_6_none_0.9", "0")
        if edge:
            qd[f"{i}_mlp.down_proj"] = ("tcq_8_none_0.9", "0")
        elif i % 8 == 1:
            qd[f"{i}_mlp.down_proj"] = ("ldlq_1_4_none_1.0", "0")       # 4.0 b SQ
        elif i % 2 == 0:
            qd[f"{i}_mlp.down_proj"] = ("tcq_7_none_0.9", "0")
        else:
            qd[f"{i}_mlp.down_proj"] = ("tcomb_6_7_0.5_none_0.9", "0")
    return qd


def average(qd):
    tot = sum(SHAPE[k.split("_", 1)[1]] * bits(v[0]) for k, v in qd.items())
    return tot / sum(SHAPE[k.split("_", 1)[1]] for k in qd)


def main():
    best = min(range(65), key=lambda n: abs(average(build(n)) - 3.25))
    qd = build(best)
    avg = average(qd)
    out = {"source": "hand-constructed (perf/make_mem3p25.py): reference quantizer set solve_mem_const.py:4-22 + the ldlq entries of "
                     "msq_results/figure1c; no OR-Tools in this image",
           "avg_bits": round(avg, 5), "qdict": {k: list(v) for k, v in qd.items()}, "merge_info": [[] for _ in range(NL)]}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "qdicts", "mem3p25.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    kinds = {}
    for v in qd.values():
        kinds[tuple(v)] = kinds.get(tuple(v), 0) + 1
    print(f"{path}: {len(qd)} linears, avg {avg:.4f} bits/weight ({best} wide projections at 3.25 b)")
    for k, c in sorted(kinds.items(), key=lambda kv: -kv[1]):
        print(f"  {c:3d} x {k}")


if __name__ == "__main__":
    main()
