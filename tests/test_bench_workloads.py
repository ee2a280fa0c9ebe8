"""GPU tests of the mixed-scheme workloads of bench.py — BASELINE.json configs[2] (memory-constrained MSQ @3.25 b, mixed
TCQ / VQ / SQ: perf/qdicts/mem3p25.json) and configs[3] (the reference's published fusion-aware MSQ result with merge_info and
(qstr, simt) tuples: figure1d; plus the unfused figure1c): bench.build_model at 2 layers, ONE token through bench.make_token
(multi-job launches and one launch per linear), EVERY output checked against the oracle's float64 GEMV over the
oracle-dequantised weights of that (possibly row-merged, possibly SIMT-repacked) layer.
Reference counterpart of the loader: eval/measure_latency_merge_simt.py:24-100."""
import os
import sys

import numpy as np
import pytest
import torch

gpu = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

GEMV_RTOL_ABS = 1e-5


@pytest.fixture(scope="module")
def env():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import bench
    import qpalette_amd
    from oracle import oracle
    qpalette_amd._native.lib()
    return bench, qpalette_amd, oracle


def _oracle_weight(oracle, info):
    m, k = info["out_features"], info["in_features"]
    if "trellis1" in info:
        return oracle.tcq_dequant(info["trellis1"].cpu().numpy(), info["tlut"].cpu().numpy(), m, k, info["tlut_bits"], info["KV"][0],
                                  c2=info["trellis2"].cpu().numpy(), KV2=info["KV"][1], split=2)
    if "trellis" in info:
        return oracle.tcq_dequant(info["trellis"].cpu().numpy(), info["tlut"].cpu().numpy(), m, k, info["tlut_bits"], info["KV"])
    return oracle.lut_tc_dequant(info["qweight"].cpu().numpy(), info["lut"].cpu().numpy(), m, k, info["lut_bits"], info["vec_sz"])


def _check(y, W, x, oracle):
    fp16_out = y.dtype == torch.float16
    ref, scale = oracle.gemv(W, x)
    tol = GEMV_RTOL_ABS * scale + 1e-30
    if fp16_out:
        tol = tol + 2.0 ** -10 * np.abs(ref) + 2.0 ** -24
    err = np.abs(y.float().cpu().numpy().astype(np.float64) - ref)
    assert np.all(err <= tol), f"max err {err.max():.3e}, max tol-ratio {(err / tol).max():.2f}"


@gpu
@pytest.mark.parametrize("workload,launch,packing", [
    ("llama3.1-8b_mem3p25", "multi", "qdict"), ("llama3.1-8b_mem3p25", "single", "qdict"),
    ("llama3.1-8b_figure1d", "multi", "qdict"), ("llama3.1-8b_figure1d", "single", "qdict"),
    ("llama3.1-8b_figure1c", "multi", "qdict"), ("llama3.1-8b_figure1c", "multi", "mi355x"),
    ("llama3.1-8b_figure1c", "single", "qdict"),
])
def test_mixed_scheme_token_matches_oracle(env, workload, launch, packing):
    bench, qp, oracle = env
    device = torch.device("cuda", 0)
    model_key, qstr = bench.WORKLOADS[workload]
    torch.manual_seed(1234)
    # layers 2 and 3 of figure1c / mem3p25 carry VQ / SQ / SIMT entries; 2 layers keep the oracle decode short
    layers = bench.build_model(qp, torch, model_key, qstr, 4, device, keep_infos=True, packing=packing)[2:4] \
        if workload != "llama3.1-8b_figure1d" else bench.build_model(qp, torch, model_key, qstr, 3, device, keep_infos=True,
                                                                     packing=packing)[1:3]
    n = 1
    gen = torch.Generator().manual_seed(5)
    xs = {}
    for groups in layers:
        for mod, k, _ in (u for grp in groups for u in grp):
            if k not in xs:
                xs[k] = torch.randn(n, k, generator=gen).half().to(device)
    token, parts = bench.make_token(qp, torch, layers, xs, n, device, launch=launch)
    outs = token()
    torch.cuda.synchronize()
    mods = [u for groups in layers for grp in groups for u in grp]
    assert len(outs) == len(mods)
    kinds = set()
    for y, (mod, k, info) in zip(outs, mods):
        kinds.add(type(mod).__name__)
        assert tuple(y.shape) == (n, mod.out_features)
        _check(y, _oracle_weight(oracle, info), xs[k].cpu().numpy(), oracle)
    if workload == "llama3.1-8b_mem3p25":  # the point of configs[2]: trellis, vector and scalar quantizers in one model
        assert {"QTIPLinearTCQ", "CombtLinearTCQ"} <= kinds and kinds & {"VQLinearPackTensorCore", "VQLinearPackSIMT"}, kinds


@gpu
def test_full_size_token_properties(env):
    """BASELINE.json configs[1] at FULL size (32 layers x 7 linears, 2.8 GB of packed weights: too big for the oracle's
    dense decode) through size-independent properties of the path: linearity in x (small-integer inputs: x1 + x2 is exact in
    fp16, so token(x1 + x2) = token(x1) + token(x2) up to fp32 summation), run-to-run determinism under HIP-graph replay, and
    agreement of the two launch structures (one launch per linear, multi-job launches) — plus
    the oracle on one linear of the LAST layer (so the model was built as specified all the way down)."""
    bench, qp, oracle = env
    device = torch.device("cuda", 0)
    model_key, qstr = bench.WORKLOADS["llama3.1-8b_tcomb_6_7"]
    torch.manual_seed(1234)
    layers = bench.build_model(qp, torch, model_key, qstr, 32, device, keep_infos=True)
    mods = [u for groups in layers for grp in groups for u in grp]
    assert len(mods) == 224
    gen = torch.Generator().manual_seed(11)

    def inputs():
        return {k: torch.randint(-3, 4, (1, k), generator=gen).half().to(device) for k in (4096, 14336)}

    x1, x2 = inputs(), inputs()
    x12 = {k: x1[k] + x2[k] for k in x1}
    xs = {k: v.clone() for k, v in x1.items()}
    token, _ = bench.make_token(qp, torch, layers, xs, 1, device, launch="multi")

    def run(x):
        for k in xs:
            xs[k].copy_(x[k])
        return [y.float().clone() for y in token()]

    y1, y2, y12 = run(x1), run(x2), run(x12)
    torch.cuda.synchronize()
    for a, b, c in zip(y1, y2, y12):
        scale = float(a.abs().max() + b.abs().max()) + 1e-30
        assert torch.allclose(a + b, c, rtol=0, atol=2e-5 * scale)
    # graph replay: the same bits every time
    stream = torch.cuda.Stream(device)
    with torch.cuda.stream(stream):
        token()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            outs = token()
        g.replay()
        torch.cuda.synchronize()
        first = [y.clone() for y in outs]
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(first, outs))
    assert all(torch.allclose(a.float(), b, rtol=1e-4, atol=1e-4 * float(b.abs().max() + 1e-30)) for a, b in zip(first, y12))
    # the other launch structures compute the same sums
    for launch in ("single",):
        tok2, _ = bench.make_token(qp, torch, layers, xs, 1, device, launch=launch)
        alt = tok2()
        torch.cuda.synchronize()
        for a, b in zip(alt, y12):
            assert torch.allclose(a.float(), b, rtol=1e-4, atol=1e-4 * float(b.abs().max() + 1e-30)), launch
    mod, k, info = mods[-1]  # down_proj of layer 31
    _check(y12[-1], _oracle_weight(oracle, info), x12[k].cpu().numpy(), oracle)


def test_mem3p25_average_bits():
    """The committed fixture is at 3.25 bits/weight (parameter-weighted), uses TCQ, VQ and SQ entries, no fusion."""
    import json
    with open(os.path.join(ROOT, "perf", "qdicts", "mem3p25.json")) as f:
        data = json.load(f)
    sys.path.insert(0, os.path.join(ROOT, "perf"))
    import make_mem3p25 as mk
    qd = {k: tuple(v) for k, v in data["qdict"].items()}
    assert len(qd) == 224 and abs(mk.average(qd) - 3.25) < 0.005 and abs(data["avg_bits"] - mk.average(qd)) < 1e-4
    names = {v[0].split("_")[0] + ("_" + v[0].split("_")[1] if v[0].startswith("ldlq") else "") for v in qd.values()}
    assert {"tcq", "tcomb", "ldlq_1", "ldlq_2"} <= names
    assert all(not m for m in data["merge_info"])
