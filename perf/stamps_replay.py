#!/usr/bin/env python3
"""Steady-state in-kernel timeline of the headline token: the token's HIP graph is replayed and EVERY GEMV launch stamps into one
buffer (STAMPS build, QPAL_STAMPS_BUF: csrc/tcq_launch.h), so the stamps read afterwards are those of the last launch of each
shape inside a replayed graph — not of an eager launch on a chip that was idle a moment ago (those show 2-4 us of wake-up).
    make -C q-palette_amd/csrc STAMPS=1 && python perf/stamps_replay.py [--layers 4]"""
import argparse
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("QPAL_LIB", os.path.join(ROOT, "q-palette_amd", "libqpal_hip_stamps.so"))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=4)
    ap.add_argument("--workload", default="llama3.1-8b_tcomb_6_7")
    ap.add_argument("--replays", type=int, default=20)
    ap.add_argument("--decode", action="store_true", help="the whole-model fused decode step of perf/decode_llama.py (rotating kernels) instead of the plain token")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    buf = torch.zeros(8 * 256 * 16 * 8, dtype=torch.int64, device=dev)
    os.environ["QPAL_STAMPS_BUF"] = hex(buf.data_ptr())
    if args.decode:
        import importlib.util
        spec = importlib.util.spec_from_file_location("_dl", os.path.join(ROOT, "perf", "decode_llama.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.main(["--no-modular", "--tokens", "16", "--layers", str(args.layers), "--context", "1024"], quiet=True)
        torch.cuda.synchronize()
        return report(buf)
    import bench
    import qpalette_amd as qp

    model_key, qstr = bench.WORKLOADS[args.workload]
    layers = bench.build_model(qp, torch, model_key, qstr, args.layers, dev)
    xs = {}
    for groups in layers:
        for mod, k, _ in (u for grp in groups for u in grp):
            xs.setdefault(k, torch.randn(1, k, device=dev).half())
    token, _ = bench.make_token(qp, torch, layers, xs, 1, dev)
    s = torch.cuda.Stream(dev)
    # capture stderr lines of the slot table: they are printed by the library at the first launch of each shape
    with torch.cuda.stream(s):
        token()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            token()
        for _ in range(args.replays):
            g.replay()
        torch.cuda.synchronize()
    report(buf)


def report(buf):
    raw = buf.cpu().numpy().astype(np.uint64).reshape(8, 256, 16, 8)
    since_entry = (raw[:, :, :, 0] >> np.uint64(52)).astype(np.float64) / 100.0  # stamp 0 - wave entry, us
    raw[:, :, :, 0] &= np.uint64((1 << 52) - 1)
    a = raw.astype(np.float64) / 100.0
    names = ["kernel arguments in hand", "x + image staged (barrier)", "(2)", "(3)", "steps done", "partials in LDS", "barrier 2", "stored"]
    for slot in range(8):
        t = a[slot]
        live = (t[:, :, 0] > 0) & (t[:, :, 7] > 0)
        if not live.any():
            continue
        nb = int(live.any(axis=1).sum())
        e = np.where(live, t[:, :, 0] - since_entry[slot], np.nan)  # wave entry
        e0 = np.nanmin(e)
        line = (f"slot {slot}: {nb} workgroups stamped; first wave entry -> last stored {np.nanmax(np.where(live, t[:, :, 7], np.nan)) - e0:.2f} us; "
                f"entries spread over {np.nanmax(e) - e0:.2f}")
        for i in (0, 2, 3, 1, 4, 5, 6, 7):
            col = np.where(live & (t[:, :, i] > 0), t[:, :, i], np.nan)
            if np.isnan(col).all():
                continue
            r = col - e
            line += f" | {names[i]} {np.nanmean(r):.2f}/{np.nanmax(r):.2f}"
        print(line + "   (us after the wave's own entry, mean/max over waves)")


if __name__ == "__main__":
    main()
