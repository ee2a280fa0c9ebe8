"""GPU parity tests of the persistent chain launches (csrc/tc_chain.h, qpalette_amd.chain): a sequence of dependent multi-job
GEMV phases in ONE kernel, every phase checked against the CPU oracle; a chain whose activations really flow from phase to
phase (x_f32) replayed from a HIP graph with changing inputs (stale-read detection); equality with the per-launch path."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GEMV_RTOL_ABS = 1e-5  # times sum |w x| (fp32 accumulation vs float64), as tests/test_gpu_parity.py


@pytest.fixture(scope="module")
def qp():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import qpalette_amd
    qpalette_amd._native.lib()
    return qpalette_amd


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as o
    return o


def _oracle_weight(oracle, qstr, info, m, k):
    if "tcomb" in qstr:
        return oracle.tcq_dequant(info["trellis1"].numpy(), info["tlut"].numpy(), m, k, info["tlut_bits"], info["KV"][0],
                                  c2=info["trellis2"].numpy(), KV2=info["KV"][1], split=2)
    if "tcq" in qstr:
        return oracle.tcq_dequant(info["trellis"].numpy(), info["tlut"].numpy(), m, k, info["tlut_bits"], info["KV"])
    return oracle.lut_tc_dequant(info["qweight"].numpy(), info["lut"].numpy(), m, k, info["lut_bits"], info["vec_sz"])


def _check(y, W, x, oracle):
    ref, scale = oracle.gemv(W, x)
    tol = GEMV_RTOL_ABS * scale + 1e-30
    err = np.abs(y.astype(np.float64) - ref)
    assert np.all(err <= tol), f"max err {err.max():.3e}, max tol-ratio {(err / tol).max():.2f}"


def _layer(qp, qstr, k, m, seed):
    info = qp.mem_op.dummy_linear_info(k, m, qstr, seed=seed, device="cpu", codebook_seed=777)
    return qp.make_linear_from_info(qstr, info).cuda(), info


def _block_plan(qp, qstr, hidden, inter, kvdim, n, seed, nblocks):
    """[q|k|v, o, gate|up, down] x nblocks with static inputs, the prezero chain of a decoder block."""
    Phase = qp.chain.Phase
    gen = torch.Generator().manual_seed(seed)
    xs = {k: torch.randn(n, k, generator=gen).half().cuda() for k in (hidden, inter)}
    plan, meta = [], []
    for b in range(nblocks):
        s = seed + 100 * b
        q, k_, v = _layer(qp, qstr, hidden, hidden, s + 1), _layer(qp, qstr, hidden, kvdim, s + 2), _layer(qp, qstr, hidden, kvdim, s + 3)
        o = _layer(qp, qstr, hidden, hidden, s + 4)
        g, u = _layer(qp, qstr, hidden, inter, s + 5), _layer(qp, qstr, hidden, inter, s + 6)
        d = _layer(qp, qstr, inter, hidden, s + 7)
        o_out = torch.full((n, hidden), float("nan"), device="cuda")
        d_out = torch.full((n, hidden), float("nan"), device="cuda")
        plan += [Phase(layers=[q[0], k_[0], v[0]], x=xs[hidden], prezero=o_out),
                 Phase(layers=[o[0]], x=xs[hidden], outs=[o_out], outs_zeroed=True),
                 Phase(layers=[g[0], u[0]], x=xs[hidden], prezero=d_out),
                 Phase(layers=[d[0]], x=xs[inter], outs=[d_out], outs_zeroed=True)]
        meta += [[(q, hidden), (k_, hidden), (v, hidden)], [(o, hidden)], [(g, hidden), (u, hidden)], [(d, inter)]]
    return plan, meta, xs


@pytest.mark.parametrize("qstr,n", [("tcomb_6_7_0.5_none_0.9", 1), ("tcq_6_none_0.9", 1), ("tcomb_6_7_0.5_none_0.9", 8),
                                    ("ldlq_2_8_none_1.0", 1), ("ldlq_1_4_none_1.0", 3), ("tcq_9_none_0.9", 2),
                                    ("ldlq_1_8_none_1.0", 1)])
def test_chain_block_matches_oracle(qp, oracle, qstr, n):
    # Llama proportions, every phase kind; batch 8 of the widest input must fit the LDS scratch beside the codebook image
    hidden, inter, kvdim = 1024, (3584 if n < 8 else 2560), 256
    plan, meta, xs = _block_plan(qp, qstr, hidden, inter, kvdim, n, seed=11, nblocks=2)
    chains = qp.chain.build_chains(plan, n, "cuda")
    assert len(chains) == 1 and isinstance(chains[0], qp.chain.GemvChain) and chains[0].nphases == 8
    chains[0].launch()
    torch.cuda.synchronize()
    assert qp.chain.chain_error("cuda") == 0
    for ph, m_ in zip(plan, meta):
        for y, ((layer, info), k) in zip(ph.results, m_):
            W = _oracle_weight(oracle, qstr, info, layer.out_features, k)
            _check(y.cpu().numpy(), W, xs[k].cpu().numpy(), oracle)
    # the per-launch path computes the same sums (fp32 summation order may differ: the planners cut K differently)
    for ph in plan:
        for y, ref in zip(ph.results, qp.multi_gemv(ph.layers, ph.x)):
            assert torch.allclose(y, ref, rtol=1e-4, atol=1e-4 * float(ref.abs().max()))


def test_chain_llama8b_block(qp, oracle):
    """Real Llama-3.1-8B shapes (BASELINE configs[1]): q|k|v, o, gate|up, down of one block, tcomb_6_7."""
    qstr, n = "tcomb_6_7_0.5_none_0.9", 1
    plan, meta, xs = _block_plan(qp, qstr, 4096, 14336, 1024, n, seed=5, nblocks=1)
    (chain,) = qp.chain.build_chains(plan, n, "cuda")
    chain.launch()
    torch.cuda.synchronize()
    assert qp.chain.chain_error("cuda") == 0
    for ph, m_ in zip(plan, meta):
        for y, ((layer, info), k) in zip(ph.results, m_):
            W = _oracle_weight(oracle, qstr, info, layer.out_features, k)
            _check(y.cpu().numpy(), W, xs[k].cpu().numpy(), oracle)


@pytest.mark.parametrize("qstr", ["tcomb_6_7_0.5_none_0.9", "ldlq_2_8_none_1.0"])
def test_chain_data_flows_between_phases_under_graph_replay(qp, oracle, qstr):
    """y0 = W0 x0; x1 = fp16(y0 * s); y1 = W1 x1; x2 = fp16(y1 * s); y2 = W2 x2 — inside ONE launch, replayed from a HIP
    graph with a different x0 each time: every phase must see the values the previous phase wrote in THIS replay."""
    Phase = qp.chain.Phase
    dim, n, s = 2048, 1, 0.02
    layers = [_layer(qp, qstr, dim, dim, 40 + i) for i in range(3)]
    Ws = [_oracle_weight(oracle, qstr, info, dim, dim) for _, info in layers]
    x0 = torch.zeros(n, dim, dtype=torch.float16, device="cuda")
    ys = [torch.zeros(n, dim, dtype=torch.float32, device="cuda") for _ in range(3)]
    plan = [Phase(layers=[layers[0][0]], x=x0, outs=[ys[0]], publish=True),
            Phase(layers=[layers[1][0]], x_f32=(ys[0], s), outs=[ys[1]], publish=True),
            Phase(layers=[layers[2][0]], x_f32=(ys[1], s), outs=[ys[2]])]
    (chain,) = qp.chain.build_chains(plan, n, "cuda")
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        chain.launch()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream):
            chain.launch()
    gen = torch.Generator().manual_seed(9)
    for rep in range(6):
        xh = torch.randn(n, dim, generator=gen).half()
        x0.copy_(xh.cuda())
        graph.replay()
        torch.cuda.synchronize()
        assert qp.chain.chain_error("cuda") == 0
        x = xh.numpy()
        for i in range(3):
            y = ys[i].cpu().numpy()
            _check(y, Ws[i], x, oracle)
            x = (y.astype(np.float32) * np.float32(s)).astype(np.float16)  # what the next phase must have staged


def test_chain_partition_and_fallback(qp):
    """build_chains cuts the plan at codec changes and hands back what no chain can take."""
    Phase = qp.chain.Phase
    a, _ = _layer(qp, "tcq_6_none_0.9", 512, 512, 1)
    b, _ = _layer(qp, "tcq_7_none_0.9", 512, 512, 2)
    info = qp.mem_op.dummy_linear_info(512, 512, "ldlq_2_8_none_1.0", seed=3)
    c = qp.VQLinearPackSIMT.gen_layer_from_info(info, device="cuda")
    x = torch.randn(1, 512).half().cuda()
    plan = [Phase(layers=[a], x=x), Phase(layers=[a], x=x), Phase(layers=[b], x=x), Phase(layers=[c], x=x), Phase(layers=[a, b], x=x)]
    parts = qp.chain.build_chains(plan, 1, "cuda")
    kinds = [type(p).__name__ for p in parts]
    assert kinds == ["GemvChain", "GemvChain", "Phase", "Phase"], kinds
    assert parts[0].nphases == 2 and parts[1].nphases == 1
    for p in parts[:2]:
        p.launch()
    torch.cuda.synchronize()
    assert torch.equal(plan[0].results[0], plan[1].results[0])
    ref = qp.multi_gemv([a], x)[0]
    assert torch.allclose(plan[0].results[0], ref, rtol=1e-4, atol=1e-5)
