#!/usr/bin/env python3
"""In-kernel timeline of a chain launch (diagnostic build: make -C q-palette_amd/csrc STAMPS=1 -> libqpal_hip_stamps.so).

Every wave stamps s_memrealtime (100 MHz) at 8 points of every phase (csrc/tc_chain.h QPAL_CSTAMP):
  0 phase start   1 first weights requested (+ codebook image)   2 decode-ahead done   3 dependency seen (after the barrier)
  4 x staged      5 MACs + remaining steps done                  6 reduce barrier       7 stores drained, arrival signalled
Prints, per phase kind of a Llama-3.1-8B block (steady state: the last block of the chain), the median / max over waves of
every segment and where the phase's wall time goes.

    QPAL_LIB=q-palette_amd/libqpal_hip_stamps.so python perf/chain_stamps.py [--blocks 4] [--qstr tcomb_6_7_0.5_none_0.9]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("QPAL_LIB", os.path.join(ROOT, "q-palette_amd", "libqpal_hip_stamps.so"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import qpalette_amd as qp  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=4)
    ap.add_argument("--qstr", default="tcomb_6_7_0.5_none_0.9")
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    n, hidden, inter, kvdim = 1, 4096, 14336, 1024
    Phase = qp.chain.Phase
    xs = {k: torch.randn(n, k, device=dev).half() for k in (hidden, inter)}
    plan, kinds = [], []

    def layer(k, m, seed):
        info = qp.mem_op.dummy_linear_info(k, m, args.qstr, seed=seed, device=dev, codebook_seed=777)
        return qp.make_linear_from_info(args.qstr, info).to(dev)

    mods = []
    for b in range(args.blocks):
        s = 100 * b
        q, k_, v, o = layer(hidden, hidden, s + 1), layer(hidden, kvdim, s + 2), layer(hidden, kvdim, s + 3), layer(hidden, hidden, s + 4)
        g, u, d = layer(hidden, inter, s + 5), layer(hidden, inter, s + 6), layer(inter, hidden, s + 7)
        mods += [q, k_, v, o, g, u, d]
        o_out = torch.empty((n, hidden), dtype=torch.float32, device=dev)
        d_out = torch.empty((n, hidden), dtype=torch.float32, device=dev)
        plan += [Phase(layers=[q, k_, v], x=xs[hidden], prezero=o_out), Phase(layers=[o], x=xs[hidden], outs=[o_out], outs_zeroed=True),
                 Phase(layers=[g, u], x=xs[hidden], prezero=d_out), Phase(layers=[d], x=xs[inter], outs=[d_out], outs_zeroed=True)]
        kinds += ["q|k|v", "o", "gate|up", "down"]
    qp.share_codebooks(mods)
    (chain,) = qp.chain.build_chains(plan, n, dev)
    grid = torch.cuda.get_device_properties(dev).multi_processor_count
    nph = len(plan)
    W = int(os.environ.get("QPAL_CHAIN_WAVES", "16"))
    dbg_all = torch.zeros(nph * grid * W * 8 + grid * 2, dtype=torch.int64, device=dev)
    dbg = dbg_all[:nph * grid * W * 8].view(nph, grid, W, 8)
    for _ in range(args.reps):
        chain.launch(dbg=dbg_all)
    torch.cuda.synchronize()
    assert qp.chain.chain_error(dev) == 0
    t = dbg.cpu().numpy().astype(np.float64) / 100.0  # us
    names = ["request", "decode-ahead", "wait dep", "stage x", "MACs+steps", "reduce barrier", "store+arrive"]
    print(f"# chain of {nph} phases ({args.blocks} blocks, {args.qstr}), grid {grid}; us; last block")
    t_end_prev = None
    for ph in range(nph - 4, nph):
        a = t[ph]  # [grid][8][8]
        start = a[:, :, 0].min()
        end = a[:, :, 7].max()
        prev_end = t[ph - 1][:, :, 7].max() if ph > 0 else start
        seg = a[:, :, 1:] - a[:, :, :-1]
        line = "  ".join(f"{nm} {np.median(seg[:, :, i]):.2f}/{seg[:, :, i].max():.2f}" for i, nm in enumerate(names))
        print(f"{kinds[ph]:8s} wall (prev phase's last arrival -> this phase's last arrival) {end - prev_end:6.2f} us | first start -> last "
              f"arrival {end - start:6.2f} | median/max per wave: {line}")
        flag = a[:, :, 3].max() - prev_end
        macs_first = a[:, :, 4].min() - prev_end
        print(f"{'':8s} last arrival of previous phase -> dependency seen by the last WG {flag:5.2f} us; -> first x staged {macs_first:5.2f} us; "
              f"MACs phase (first staged -> last done) {a[:, :, 5].max() - a[:, :, 4].min():5.2f} us; tail (last MACs done -> last arrival) "
              f"{end - a[:, :, 5].max():5.2f} us")
    clk = dbg_all[nph * grid * W * 8:].view(grid, 2).cpu().numpy().astype(np.float64)
    ghz = clk[:, 0] / clk[:, 1] * 0.1
    print(f"shader clock over the launch (s_memtime / s_memrealtime): median {np.median(ghz):.3f} GHz, min {ghz.min():.3f}, max {ghz.max():.3f}")
    tot = t[nph - 1][:, :, 7].max() - t[nph - 5][:, :, 7].max()
    if tot > 0:
        print(f"block total {tot:.2f} us -> {1e6 / (tot * 32):.0f} tokens/s at 32 blocks")


if __name__ == "__main__":
    main()
