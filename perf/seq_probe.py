#!/usr/bin/env python3
"""Launch-sequence probe: the headline token (or another bench workload) replayed (a) from a HIP graph, (b) from a LaunchSequence with
overlapped launch boundaries, (c) from a LaunchSequence in plain stream order (QPAL_SEQ_OVERLAP=0).  Outputs are compared bit for bit.

    python perf/seq_probe.py [--workload llama3.1-8b_tcomb_6_7] [--layers N] [--steps 50] [--batch 1]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="llama3.1-8b_tcomb_6_7")
    ap.add_argument("--layers", type=int, default=0)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--batch", type=int, default=1)
    args = ap.parse_args()
    import torch

    import bench
    import qpalette_amd as qp

    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    model_key, qstr = bench.WORKLOADS[args.workload]
    nlayers = args.layers or qp.mem_op.get_layer_info(model_key)["nlayers"]
    torch.manual_seed(1234)
    layers = bench.build_model(qp, torch, model_key, qstr, nlayers, device)
    n = args.batch
    xs = {}
    for groups in layers:
        for mod, k, _ in (u for grp in groups for u in grp):
            if k not in xs:
                xs[k] = torch.randn(n, k, device=device).half()
    token, owned = bench.make_token(qp, torch, layers, xs, n, device, launch="multi")
    stream = torch.cuda.Stream(device)

    def timed(run, steps):
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(stream)
        for _ in range(steps):
            run()
        e1.record(stream)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3, e0.elapsed_time(e1) / steps

    with torch.cuda.stream(stream):
        outs = token()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            outs = token()
        g.replay()
        torch.cuda.synchronize()
        ref = [o.clone() for o in outs]
        wall_g, dev_g = timed(g.replay, args.steps)
        print(f"graph replay         : {wall_g:.4f} ms/token wall, {dev_g:.4f} ms by events -> {n * 1e3 / wall_g:.1f} tok/s", flush=True)

        results = {}
        for name, env, need0 in (("sequence, overlapped", "1", "0"), ("sequence, ordered   ", "0", "0"), ("overlapped, NO WAIT (invalid)", "1", "1")):
            os.environ["QPAL_SEQ_OVERLAP"] = env
            os.environ["QPAL_SEQ_NEED0"] = need0
            seq = qp.LaunchSequence()
            souts = seq.capture(token)
            for o in souts:
                o.fill_(float("nan"))
            if owned is not None:  # the first launch's block is zeroed by nobody's prezero inside the token: by the library's memset or not at all
                pass
            seq.launch(stream)
            torch.cuda.synchronize()
            info = seq.info(read_error=True)
            bad = sum(0 if torch.equal(a.view(torch.int32), b.view(torch.int32)) else 1 for a, b in zip(souts, ref))
            for rep in range(3):   # replays must stay bit-identical (the counters only grow)
                seq.launch(stream)
            torch.cuda.synchronize()
            bad2 = sum(0 if torch.equal(a.view(torch.int32), b.view(torch.int32)) else 1 for a, b in zip(souts, ref))
            wall_s, dev_s = timed(lambda: seq.launch(stream), args.steps)
            info2 = seq.info(read_error=True)
            print(f"{name} : {wall_s:.4f} ms/token wall, {dev_s:.4f} ms by events -> {n * 1e3 / wall_s:.1f} tok/s | {info} | outputs differing from "
                  f"the graph's: {bad} after 1 replay, {bad2} after 4 of {len(ref)} | error word after timing {info2['error']}", flush=True)
            results[name] = (wall_s, bad, bad2, info2["error"])
        ok = all(r[1] == 0 and r[2] == 0 and r[3] == 0 for k_, r in results.items() if "invalid" not in k_)
        print("PARITY", "ok" if ok else "FAILED")
        return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
