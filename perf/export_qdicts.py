#!/usr/bin/env python3
"""Export the reference's published MSQ results (quantizer per linear + layer-fusion choices) as JSON data.
Run in the builder container only (reads /root/reference/msq_results; data files, no code).

  figure1c: latency-aware MSQ without fusion (avg 2.86 b/w)   msq_results/figure1c/0.0_8.0bit_1.11{,_merge_info}.pt
  figure1d: fusion-aware MSQ (avg 2.96 b/w)                    msq_results/figure1d/0.0_8.0bit_1.17{,_merge_info}.pt
Values are (quantizer_str, simt) with simt "1" = the SIMT ("CUDA-core") kernel variant
(eval/measure_latency_merge_simt.py)."""
import json
import os

import torch

REF = "/root/reference/msq_results"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "qdicts")
for name, stem in (("figure1c", "figure1c/0.0_8.0bit_1.11"), ("figure1d", "figure1d/0.0_8.0bit_1.17")):
    qdict = torch.load(f"{REF}/{stem}.pt", weights_only=True)
    merge = torch.load(f"{REF}/{stem}_merge_info.pt", weights_only=True)
    data = {"source": f"msq_results/{stem}.pt", "qdict": {k: list(v) if isinstance(v, (tuple, list)) else [v, "0"]
                                                          for k, v in qdict.items()},
            "merge_info": [list(m) for m in merge]}
    with open(os.path.join(OUT, f"{name}.json"), "w") as f:
        json.dump(data, f, indent=0, sort_keys=True)
    print(name, len(qdict), "linears;", sum(len(m) for m in merge), "merges")
