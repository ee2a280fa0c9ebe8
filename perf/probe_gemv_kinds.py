#!/usr/bin/env python3
"""Profiling helper: a few eager calls of one layer's fused GEMV per quantizer (run under rocprofv3 --kernel-trace --stats).

    python perf/probe_gemv_kinds.py k ldlq_1_8_none_1.0 ldlq_2_8_none_1.0 [--simt]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import qpalette_amd as qp
from latency_table import SHAPES


def main():
    simt = "--simt" in sys.argv
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    lk, qstrs = args[0], args[1:]
    dev = torch.device("cuda", 0)
    m, k = SHAPES[lk]
    for qstr in qstrs:
        mods = []
        for c in range(8):
            info = qp.mem_op.dummy_linear_info(k, m, qstr, seed=c, device=dev, codebook_seed=7)
            mods.append(qp.VQLinearPackSIMT.gen_layer_from_info(info, device=dev) if simt else qp.make_linear_from_info(qstr, info).to(dev))
        qp.share_codebooks(mods)
        x = torch.randn(1, k, device=dev).half()
        for rep in range(5):
            for mod in mods:
                mod._gemv(x, 1)
        torch.cuda.synchronize()
        print(qstr, "done", flush=True)


if __name__ == "__main__":
    main()
