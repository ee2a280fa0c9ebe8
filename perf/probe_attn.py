#!/usr/bin/env python3
"""Profiling helper: qpal_attn_rope_decode at one (max_len, pos), with and without the split-context workspace; eager calls
timed with HIP events in a stream-ordered loop over 32 distinct caches (run under rocprofv3 for kernel durations).

    python perf/probe_attn.py 2048 40   [nq nkv hd]
"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import qpalette_amd as qp


def main():
    ctx, pos = int(sys.argv[1]), int(sys.argv[2])
    nq, nkv, hd = (int(a) for a in sys.argv[3:6]) if len(sys.argv) > 5 else (32, 8, 128)
    dev = torch.device("cuda", 0)
    nat = qp._native
    L = 32
    kc = [(torch.randn(nkv, ctx, hd, device=dev) * 0.5).half() for _ in range(L)]
    vc = [(torch.randn(nkv, ctx, hd, device=dev) * 0.5).half() for _ in range(L)]
    q, k, v = torch.randn(nq * hd, device=dev), torch.randn(nkv * hd, device=dev), torch.randn(nkv * hd, device=dev)
    inv_freq = 1.0 / (500000.0 ** (torch.arange(0, hd, 2, device=dev).float() / hd))
    pos_t = torch.tensor([pos], dtype=torch.long, device=dev)
    out = torch.empty(nq * hd, dtype=torch.float16, device=dev)
    wsb = nat.lib().qpal_attn_ws_bytes(nq, nkv, hd, ctx)
    ws = torch.zeros(max(wsb, 4) // 4, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    for name, w, wb in (("one workgroup per head", None, 0), ("split-context", ws.data_ptr(), wsb)):
        if name.startswith("split") and not wsb:
            continue
        def token():
            for i in range(L):
                nat.check(nat.lib().qpal_attn_rope_decode(q.data_ptr(), k.data_ptr(), v.data_ptr(), kc[i].data_ptr(), vc[i].data_ptr(), out.data_ptr(),
                                                          pos_t.data_ptr(), inv_freq.data_ptr(), nq, nkv, hd, ctx, 1.0 / math.sqrt(hd), w, wb, stream), "attn")
        token()
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream(dev)
        with torch.cuda.stream(s):
            stream = s.cuda_stream
            token()
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                token()
            g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(20):
                g.replay()
            e1.record(s)
            torch.cuda.synchronize()
        stream = torch.cuda.current_stream(dev).cuda_stream
        print(f"max_len {ctx} pos {pos} {name}: {e0.elapsed_time(e1) * 1e3 / (20 * L):.2f} us per launch (graph replay, {L} caches)", flush=True)


if __name__ == "__main__":
    main()
