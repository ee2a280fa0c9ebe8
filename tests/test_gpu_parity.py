"""GPU parity tests (run on the MI355X box: ``pytest -m gpu``).  Every call goes through the
``torch.ops.ours_lib`` operators, i.e. through the C-ABI of include/qpal.h, and is compared with the CPU
oracle on the same inputs.

Bars: decode-to-fp16 and all index arithmetic — BIT-EXACT.  Fused GEMV — fp32 accumulation vs the
oracle's float64 sum: |err| <= 1e-5 * sum_k |w_k x_k| (k <= 28672; fp32 unit roundoff 6e-8 times a
sqrt(k)-to-k growth factor), with fp16-output kernels (SIMT family) adding one fp16 rounding.
"""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GEMV_RTOL_ABS = 1e-5  # times sum |w x|


@pytest.fixture(scope="module")
def qp():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import qpalette_amd
    qpalette_amd._native.lib()  # fail loudly if the HIP library is missing
    return qpalette_amd


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as o
    return o


def _g(name):
    return np.load(os.path.join(os.path.dirname(__file__), "golden", name))


def _cuda(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _bits(t):
    return t.cpu().numpy().view(np.uint16)


def _check_gemv(y, W, x, oracle, fp16_out=False):
    ref, scale = oracle.gemv(W, x)
    tol = GEMV_RTOL_ABS * scale + 1e-30
    if fp16_out:
        tol = tol + 2.0 ** -10 * np.abs(ref) + 2.0 ** -24
    err = np.abs(y.astype(np.float64) - ref)
    assert np.all(err <= tol), f"max err {err.max():.3e}, max tol-ratio {(err / tol).max():.2f}"


TCQ_COMBOS = [(9, kv) for kv in range(2, 11)] + [(10, 8), (10, 9), (10, 10), (11, 9), (11, 10)]


@pytest.mark.parametrize("S,KV", TCQ_COMBOS)
def test_tcq_golden(qp, oracle, S, KV):
    g = _g("tcq.npz")
    m, k = int(g["m"]), int(g["k"])
    tr, tl, W = g[f"tcq_S{S}_KV{KV}_trellis"], g[f"tcq_S{S}_KV{KV}_tlut"], g[f"tcq_S{S}_KV{KV}_W"]
    dq = qp.ops.get_op(f"decompress_tcq_{S}_{KV}")(_cuda(tr), _cuda(tl), m, k)
    assert np.array_equal(_bits(dq), W.view(np.uint16))
    rng = np.random.default_rng(KV)
    for n in (1, 3, 8):
        x = rng.standard_normal((n, k)).astype(np.float16)
        y = qp.ops.get_op(f"decompress_gemm_tcq_{m}_{n}_{k}_{S}_{KV}")(_cuda(tr), _cuda(x), _cuda(tl))
        assert y.dtype == torch.float32 and tuple(y.shape) == (n, m)
        _check_gemv(y.cpu().numpy(), W, x, oracle)


@pytest.mark.parametrize("S,KV", [(9, 2), (9, 3), (9, 5), (9, 6), (9, 7), (9, 9), (10, 8), (10, 9), (11, 9)])
def test_tcq_comb_combt_golden(qp, oracle, S, KV):
    g = _g("tcq.npz")
    m, k = int(g["m"]), int(g["k"])
    t1, t2 = g[f"tcq_S{S}_KV{KV}_trellis"], g[f"tcq_S{S}_KV{KV + 1}_trellis"]
    tl = g[f"tcq_S{S}_KV{KV}_tlut"]
    W1, W2 = g[f"tcq_S{S}_KV{KV}_W"], g[f"tcq_S{S}_KV{KV + 1}_W"]
    rng = np.random.default_rng(100 + KV)
    # combt: column halves
    Wt = np.concatenate([W1, W2], 1)
    dq = qp.ops.get_op(f"decompress_tcq_combt_{S}_{KV}_{KV + 1}")(_cuda(t1), _cuda(t2), _cuda(tl), m, 2 * k)
    assert np.array_equal(_bits(dq), Wt.view(np.uint16))
    for n in (1, 4):
        x = rng.standard_normal((n, 2 * k)).astype(np.float16)
        y = qp.ops.get_op(f"decompress_gemm_tcq_combt_{m}_{n}_{2 * k}_{S}_{KV}_{KV + 1}")(
            _cuda(t1), _cuda(t2), _cuda(x), _cuda(tl))
        _check_gemv(y.cpu().numpy(), Wt, x, oracle)
    # comb: row halves
    Wr = np.concatenate([W1, W2], 0)
    dq = qp.ops.get_op(f"decompress_tcq_comb_{S}_{KV}_{KV + 1}")(_cuda(t1), _cuda(t2), _cuda(tl), 2 * m, k)
    assert np.array_equal(_bits(dq), Wr.view(np.uint16))
    x = rng.standard_normal((2, k)).astype(np.float16)
    y = qp.ops.get_op(f"decompress_gemm_tcq_comb_{2 * m}_2_{k}_{S}_{KV}_{KV + 1}")(_cuda(t1), _cuda(t2), _cuda(x), _cuda(tl))
    _check_gemv(y.cpu().numpy(), Wr, x, oracle)


@pytest.mark.parametrize("vec,bits", [(1, b) for b in range(2, 9)] + [(2, b) for b in range(2, 13)])
def test_lut_tc_golden(qp, oracle, vec, bits):
    g = _g("lut_tc.npz")
    m, k = int(g["m"]), int(g["k"])
    q, lut, W = g[f"tc_v{vec}_b{bits}_qweight"], g[f"tc_v{vec}_b{bits}_lut"], g[f"tc_v{vec}_b{bits}_W"]
    vtypes = ["vq2"] if vec == 2 else (["sq_dup", "sq"] if bits <= 4 else ["sq"])
    rng = np.random.default_rng(bits * 3 + vec)
    for vt in vtypes:
        dq = qp.ops.get_op(f"decompress_{bits}_{vt}")(_cuda(q), _cuda(lut), m, k)
        assert np.array_equal(_bits(dq), W.view(np.uint16))
        for n in (1, 2, 5):
            x = rng.standard_normal((n, k)).astype(np.float16)
            y = qp.ops.get_op(f"decompress_gemm_{m}_{n}_{k}_{bits}_{vt}")(_cuda(q), _cuda(x), _cuda(lut))
            _check_gemv(y.cpu().numpy(), W, x, oracle)
    out = torch.full((1, m), float("nan"), device="cuda")
    x = rng.standard_normal((1, k)).astype(np.float16)
    qp.ops.get_op(f"decompress_gemv_{m}_{k}_{bits}_{vtypes[-1]}")(_cuda(q), _cuda(x), _cuda(lut), out)
    _check_gemv(out.cpu().numpy(), W, x, oracle)


@pytest.mark.parametrize("vec,bits", [(1, b) for b in range(2, 9)] + [(2, b) for b in range(3, 13)]
                         + [(4, b) for b in range(6, 13)])
def test_simt_golden(qp, oracle, vec, bits):
    g = _g("simt.npz")
    m, k = int(g["m"]), int(g[f"simt_v{vec}_k"])
    q, idx = g[f"simt_v{vec}_b{bits}_qweight"], g[f"simt_v{vec}_b{bits}_idx"]
    rng = np.random.default_rng(bits * 5 + vec)
    lut = rng.standard_normal((1 << bits, vec)).astype(np.float16)
    W = lut[idx].reshape(m, k)
    qd, ld = _cuda(q.view(np.int32)), _cuda(lut)
    if vec == 1:
        dq = qp.ops.get_op("sq_pack_dequant_simt")(qd, ld, bits, m, k)
    else:
        dq = qp.ops.get_op(f"vq_pack_dequant_simt_{vec}_{bits}")(qd, ld, m, k)
    assert np.array_equal(_bits(dq), W.view(np.uint16))
    for n in (1, 3, 8):
        x = rng.standard_normal((n, 1, k)).astype(np.float16)
        if vec == 1:
            y = qp.ops.get_op("sq_pack_gemm_simt")(_cuda(x), qd, ld, bits)
        else:
            y = qp.ops.get_op(f"vq_pack_gemm_simt_{n}_{vec}_{bits}")(_cuda(x), qd, ld)
        assert y.dtype == torch.float16 and tuple(y.shape) == (n, 1, m)
        _check_gemv(y.float().cpu().numpy().reshape(n, m), W, x.reshape(n, k), oracle, fp16_out=True)


@pytest.mark.parametrize("bits", [2, 4, 5, 8])
def test_sq_pack_gemm_inplace_simt(qp, oracle, bits):
    """sq_pack_gemm_inplace_simt (reference: lib/linear/__init__.py:372-378 — pack_gemm into a caller-owned `output`): the op
    mutates `output` and returns nothing; NaN-filled before the call, every element must come back written and agree with the
    oracle and, bit for bit, with the allocating twin sq_pack_gemm_simt."""
    g = _g("simt.npz")
    m, k = int(g["m"]), int(g["simt_v1_k"])
    q, idx = g[f"simt_v1_b{bits}_qweight"], g[f"simt_v1_b{bits}_idx"]
    rng = np.random.default_rng(bits * 11 + 1)
    lut = rng.standard_normal((1 << bits, 1)).astype(np.float16)
    W = lut[idx].reshape(m, k)
    qd, ld = _cuda(q.view(np.int32)), _cuda(lut)
    inplace, alloc = qp.ops.get_op("sq_pack_gemm_inplace_simt"), qp.ops.get_op("sq_pack_gemm_simt")
    for n in (1, 3, 8):
        x = rng.standard_normal((n, 1, k)).astype(np.float16)
        out = torch.full((n, 1, m), float("nan"), dtype=torch.float16, device="cuda")
        assert inplace(_cuda(x), qd, ld, out, bits) is None
        assert not torch.isnan(out).any()
        _check_gemv(out.float().cpu().numpy().reshape(n, m), W, x.reshape(n, k), oracle, fp16_out=True)
        assert torch.equal(out, alloc(_cuda(x), qd, ld, bits))
    with pytest.raises(RuntimeError):
        inplace(_cuda(rng.standard_normal((1, 1, k)).astype(np.float16)), qd, ld, torch.empty((1, 1, m), dtype=torch.float16, device="cuda"), 9)


def test_opcheck_one_op_of_each_family(qp):
    """torch.library.opcheck over one operator of every family of the ours_lib surface (schema, fake-tensor / meta agreement,
    mutation annotations, AOT dispatch): what torch.compile relies on when the reference's modules call these names."""
    from torch.library import opcheck
    k, m = 256, 64
    rng = np.random.default_rng(3)
    x1 = torch.randn(1, k, device="cuda").half()
    utils = ("test_schema", "test_faketensor", "test_aot_dispatch_static", "test_aot_dispatch_dynamic")
    # TCQ, single stream and column split
    for qstr in ("tcq_4_none_0.9", "tcomb_6_7_0.5_none_0.9"):
        info = qp.mem_op.dummy_linear_info(k, m, qstr, seed=1)
        layer = qp.make_linear_from_info(qstr, info).cuda()
        if "tcomb" in qstr:
            name = f"decompress_gemm_tcq_combt_{m}_1_{k}_{layer.tlut_bits}_{layer.KV[0]}_{layer.KV[1]}"
            opcheck(qp.ops.get_op(name).default, (layer.trellis1, layer.trellis2, x1, layer.tlut), test_utils=utils)
        else:
            name = f"decompress_gemm_tcq_{m}_1_{k}_{layer.tlut_bits}_{layer.KV}"
            opcheck(qp.ops.get_op(name).default, (layer.trellis, x1, layer.tlut), test_utils=utils)
            opcheck(qp.ops.get_op(f"decompress_tcq_{layer.tlut_bits}_{layer.KV}").default, (layer.trellis, layer.tlut, m, k), test_utils=utils)
    # VQ / SQ, tensor-core packing: allocating GEMV and the gemv that mutates `out`
    info = qp.mem_op.dummy_linear_info(k, m, "ldlq_2_6_none_1.0", seed=2)
    layer = qp.VQLinearPackTensorCore.gen_layer_from_info(info).cuda()
    opcheck(qp.ops.get_op(f"decompress_gemm_{m}_1_{k}_6_vq2").default, (layer.qweight, x1, layer.lut), test_utils=utils)
    out = torch.zeros(m, dtype=torch.float32, device="cuda")
    opcheck(qp.ops.get_op(f"decompress_gemv_{m}_{k}_6_vq2").default, (layer.qweight, x1, layer.lut, out), test_utils=utils)
    # SIMT packings: allocating, in-place
    idx = torch.from_numpy(rng.integers(0, 16, size=(m, k), dtype=np.int64))
    q = qp.packers.pack_qweight_sq_simt(idx, 4).cuda()
    lut = torch.randn(16, 1, device="cuda").half()
    x3 = x1.reshape(1, 1, k)
    opcheck(qp.ops.get_op("sq_pack_gemm_simt").default, (x3, q, lut, 4), test_utils=utils)
    opcheck(qp.ops.get_op("sq_pack_gemm_inplace_simt").default, (x3, q, lut, torch.zeros(1, 1, m, dtype=torch.float16, device="cuda"), 4),
            test_utils=utils)
    idx = torch.from_numpy(rng.integers(0, 64, size=(m, k // 2), dtype=np.int64))
    q2 = qp.packers.pack_qweight_vq_simt(idx, 6, 2).cuda()
    lut2 = torch.randn(64, 2, device="cuda").half()
    opcheck(qp.ops.get_op("vq_pack_gemm_simt_1_2_6").default, (x3, q2, lut2), test_utils=utils)
    opcheck(qp.ops.get_op("vq_pack_dequant_simt_2_6").default, (q2, lut2, m, k), test_utils=utils)


@pytest.mark.parametrize("vec,bits,m,k", [(1, 2, 8199, 2560), (1, 7, 8199, 2560), (2, 3, 8199, 2560), (2, 9, 8199, 6144),
                                          (4, 6, 8199, 5120), (4, 12, 8199, 4096), (2, 5, 1024, 14336), (4, 8, 4096, 14336),
                                          (1, 4, 1023, 4096)])
def test_simt_gemv_geometries(qp, oracle, vec, bits, m, k):
    """Every launch geometry of the SIMT GEMV (csrc/simt_kernels.h simt_gemv_geometry): two rows per wave with several
    blocks per row and rows strided over a persistent grid (m > 2 * resident waves), a row per wave with the halves on
    alternate blocks (small m), odd row counts, trailing partial blocks, every workgroup size (table 0.5 .. 128 KiB)."""
    rng = np.random.default_rng(bits * 7 + vec + m)
    idx = torch.from_numpy(rng.integers(0, 1 << bits, size=(m, k // vec), dtype=np.int64))
    lut = rng.standard_normal((1 << bits, vec)).astype(np.float16)
    q = qp.packers.pack_qweight_sq_simt(idx, bits) if vec == 1 else qp.packers.pack_qweight_vq_simt(idx, bits, vec)
    assert np.array_equal(oracle.simt_indices(q.numpy().view(np.uint32), m, k, bits, vec), idx.numpy())
    W = lut[idx.numpy()].reshape(m, k)
    qd, ld = q.cuda(), _cuda(lut)
    for n in (1, 2):
        x = rng.standard_normal((n, 1, k)).astype(np.float16)
        if vec == 1:
            y = qp.ops.get_op("sq_pack_gemm_simt")(_cuda(x), qd, ld, bits)
        else:
            y = qp.ops.get_op(f"vq_pack_gemm_simt_{n}_{vec}_{bits}")(_cuda(x), qd, ld)
        _check_gemv(y.float().cpu().numpy().reshape(n, m), W, x.reshape(n, k), oracle, fp16_out=True)


@pytest.mark.parametrize("vec,bits", [(1, 3), (1, 4), (1, 8), (2, 5), (2, 8), (2, 12)])
def test_tc_to_simt_golden(qp, vec, bits):
    g = _g("simt.npz")
    m, k = int(g["conv_m"]), int(g["conv_k"])
    got = qp.ops.tc_to_simt(_cuda(g[f"conv_v{vec}_b{bits}_tc"]), m, k, bits, vec)
    assert np.array_equal(got.cpu().numpy().view(np.uint32), g[f"conv_v{vec}_b{bits}_simt"])


# ------------------------------------------------------------------ Llama shapes, synthetic weights
LLAMA = [("tcomb_6_7_0.5_none_0.9", 4096, 4096), ("tcomb_6_7_0.5_none_0.9", 4096, 1024),
         ("tcomb_6_7_0.5_none_0.9", 4096, 14336), ("tcomb_6_7_0.5_none_0.9", 14336, 4096),
         ("tcq_6_none_0.9", 4096, 6144), ("tcq_3_none_0.9", 4096, 4096), ("tcq_8_none_0.9", 14336, 4096),
         ("tcq_9_none_0.9", 4096, 1024), ("tcq_10_none_0.9", 4096, 2048), ("tcomb_9_10_0.5_none_0.9", 4096, 1024),
         ("ldlq_1_4_none_1.0", 4096, 4096), ("ldlq_1_6_none_1.0", 4096, 1024), ("ldlq_2_12_none_1.0", 4096, 1024),
         ("ldlq_2_8_none_1.0", 4096, 28672), ("ldlq_1_8_none_1.0", 4096, 4096)]


def _oracle_weight(oracle, qstr, info, m, k):
    if "tcomb" in qstr:
        return oracle.tcq_dequant(info["trellis1"].numpy(), info["tlut"].numpy(), m, k, info["tlut_bits"], info["KV"][0],
                                  c2=info["trellis2"].numpy(), KV2=info["KV"][1], split=2)
    if "tcq" in qstr:
        return oracle.tcq_dequant(info["trellis"].numpy(), info["tlut"].numpy(), m, k, info["tlut_bits"], info["KV"])
    return oracle.lut_tc_dequant(info["qweight"].numpy(), info["lut"].numpy(), m, k, info["lut_bits"], info["vec_sz"])


@pytest.mark.parametrize("qstr,k,m", LLAMA)
def test_llama_shapes_modules(qp, oracle, qstr, k, m):
    info = qp.mem_op.dummy_linear_info(k, m, qstr, seed=m + k, device="cpu")
    layer = qp.make_linear_from_info(qstr, info).cuda()
    W = _oracle_weight(oracle, qstr, info, m, k)
    assert np.array_equal(_bits(layer.get_weight()), W.view(np.uint16))
    gen = torch.Generator().manual_seed(k)
    for n in (1, 8):
        x = torch.randn(n, k, generator=gen).half()
        y = layer(x.cuda().float())  # fp32 in -> fp32 out keeps the kernel's fp32 result
        assert y.dtype == torch.float32 and tuple(y.shape) == (n, m)
        # (few-row VQ / SQ layers run the SIMT-order kernel on a re-packed copy of their codes at batch <= 8: fp16 output)
        simt_twin = isinstance(layer, qp.VQLinearPackTensorCore) and m <= layer.SIMT_TWIN_MAX_ROWS
        assert simt_twin == (getattr(layer, "_simt_qweight", None) is not None)
        _check_gemv(y.cpu().numpy(), W, x.numpy(), oracle, fp16_out=simt_twin)
    # bs > max_chunked_batch (2 x max_fused_batch = 256; SIMT packings: 8): the module's decode-to-fp16 + fp16 GEMM path — what the perplexity eval exercises
    # (eval_qdict.py:17-38 at bs = 8192; lib/linear/tcq_linear.py:75-84); tighter check: test_module_path_above_the_fused_batch
    x = torch.randn(272, k, generator=gen).half()
    assert x.shape[0] > max(layer.max_fused_batch, layer.max_chunked_batch)
    y = layer(x.cuda()).float().cpu().numpy()
    ref = (x.float() @ torch.from_numpy(W).float().T).numpy()
    assert np.allclose(y, ref, rtol=2e-2, atol=2e-2 * np.abs(ref).max())


def test_merge_infos_is_row_concat(qp, oracle):
    qstr, k = "tcomb_6_7_0.5_none_0.9", 4096
    a = qp.mem_op.dummy_linear_info(k, 1024, qstr, seed=1)
    b = qp.mem_op.dummy_linear_info(k, 1024, qstr, seed=2)
    b["tlut"] = a["tlut"].clone()
    merged = qp.CombtLinearTCQ.merge_infos(a, b)
    fused = qp.CombtLinearTCQ.gen_layer_from_info(merged).cuda()
    la, lb = qp.CombtLinearTCQ.gen_layer_from_info(a).cuda(), qp.CombtLinearTCQ.gen_layer_from_info(b).cuda()
    x = torch.randn(1, k, generator=torch.Generator().manual_seed(3)).cuda()
    y = fused(x)
    assert torch.equal(y[:, :1024], la(x)) and torch.equal(y[:, 1024:], lb(x))


def test_simt_module_loads_tensor_core_file(qp, oracle):
    k, m = 4096, 1024
    for qstr in ("ldlq_1_4_none_1.0", "ldlq_2_8_none_1.0"):
        info = qp.mem_op.dummy_linear_info(k, m, qstr, seed=7)
        layer = qp.VQLinearPackSIMT.gen_layer_from_info(info, device="cuda")
        W = _oracle_weight(oracle, qstr, info, m, k)
        assert np.array_equal(_bits(layer.get_weight()), W.view(np.uint16))
        x = torch.randn(2, k, generator=torch.Generator().manual_seed(5)).half()
        y = layer(x.cuda())
        _check_gemv(y.float().cpu().numpy(), W, x.numpy(), oracle, fp16_out=True)


def test_simt_twin_follows_reloaded_codes(qp, oracle):
    """A few-row VQ / SQ layer in tensor-core packing answers batch <= 8 from a SIMT-order twin of its codes (vq_linear.py).  The twin
    must follow the codes: forward, then load_state_dict() / an in-place update / a device round trip, then forward again — every
    answer against the oracle on the codes the layer holds at that moment (round 4 cached the twin of the FIRST codes for good)."""
    k, m = 4096, 1024
    for qstr in ("ldlq_1_4_none_1.0", "ldlq_2_8_none_1.0"):
        infos = [qp.mem_op.dummy_linear_info(k, m, qstr, seed=70 + i) for i in range(3)]
        Ws = [_oracle_weight(oracle, qstr, info, m, k) for info in infos]
        assert not np.array_equal(Ws[0], Ws[1])
        layer = qp.VQLinearPackTensorCore(k, m, infos[0]["lut_bits"], infos[0]["vec_sz"], device="cuda")
        x = torch.randn(2, k, generator=torch.Generator().manual_seed(5)).half()

        def check(W):
            y = layer._gemv(x.cuda(), 2)
            assert y.dtype == torch.float16 and getattr(layer, "_simt_qweight", None) is not None   # the twin answered
            _check_gemv(y.float().cpu().numpy(), W, x.numpy(), oracle, fp16_out=True)
            assert np.array_equal(_bits(layer.get_weight()), W.view(np.uint16))

        layer._gemv(x.cuda(), 2)  # (random initial codes: a twin exists before anything is loaded)
        src = qp.VQLinearPackTensorCore.gen_layer_from_info(infos[0])
        layer.load_state_dict(src.state_dict())
        check(Ws[0])
        with torch.no_grad():
            layer.qweight.copy_(infos[1]["qweight"].cuda())  # in-place update: the version counter moves
            layer.lut.copy_(infos[1]["lut"].cuda())
        check(Ws[1])
        layer = layer.cpu().cuda()  # a move: new storage
        check(Ws[1])
        layer.qweight = infos[2]["qweight"].cuda()  # re-assignment of the buffer
        layer.lut = infos[2]["lut"].cuda()
        check(Ws[2])
        layer.qweight.data.copy_(infos[0]["qweight"].cuda())  # invisible to the version counter: prepare() is the documented call
        layer.lut.data.copy_(infos[0]["lut"].cuda())
        layer.prepare()
        check(Ws[0])


def test_graph_capture_and_determinism(qp):
    qstr, k, m = "tcomb_6_7_0.5_none_0.9", 4096, 14336
    layer = qp.make_linear_from_info(qstr, qp.mem_op.dummy_linear_info(k, m, qstr, seed=9)).cuda()
    x = torch.randn(1, k, device="cuda")
    eager = layer(x).clone()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        layer(x)
        with torch.cuda.graph(g, stream=s):
            y = layer(x)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    assert torch.equal(y, eager)


def test_errors_are_exceptions(qp):
    with pytest.raises(AttributeError):
        qp.ops.get_op("decompress_gemm_tcq_4096_1_4096_9_11")  # KV outside the table
    op = qp.ops.get_op("decompress_gemm_tcq_64_1_64_9_4")
    with pytest.raises(RuntimeError):
        op(torch.zeros(3, dtype=torch.int16, device="cuda"), torch.zeros(1, 64, device="cuda"),
           torch.zeros(512, 2, dtype=torch.float16, device="cuda"))


def test_multi_job_launch_equals_single_launches(qp):
    """q|k|v and gate|up as one multi-job launch agree with one launch per linear (the launch planner may
    cut K differently, so the fp32 summation order - not the arithmetic - can differ: tolerance, not bits)."""
    k = 4096
    for qstr, ms in (("tcomb_6_7_0.5_none_0.9", [4096, 1024, 1024]), ("tcq_6_none_0.9", [14336, 14336]),
                     ("ldlq_2_8_none_1.0", [1024, 4096]), ("tcq_10_none_0.9", [1024, 2048, 512])):
        layers = [qp.make_linear_from_info(qstr, qp.mem_op.dummy_linear_info(k, m, qstr, seed=m + i)).cuda()
                  for i, m in enumerate(ms)]
        for n in (1, 3):
            x = torch.randn(n, k, generator=torch.Generator().manual_seed(n)).cuda()
            ys = qp.multi_gemv(layers, x)
            for layer, y in zip(layers, ys):
                ref = layer._gemv(x, n)
                # (a few-row VQ / SQ layer answers a single launch from its SIMT-order twin: fp16 output)
                rtol = 1e-4 if ref.dtype == torch.float32 else 2.0 ** -10
                assert torch.allclose(y, ref.float(), rtol=rtol, atol=rtol * float(ref.abs().max()))


@pytest.mark.parametrize("njobs", [2, 3, 4, 5, 8])
def test_multi_job_launch_job_counts(qp, oracle, njobs):
    """Launches of 2..8 jobs against the oracle: up to 4 jobs a workgroup knows the job of its first item from a preloaded
    packed item table (tc_kernels.h `eie`), beyond that it scans the table in the kernel-argument block; unequal row counts put
    the job boundaries at odd workgroup indices, and 8 x 1536 rows are more items than one round of workgroups."""
    k = 4096
    qstr = "tcq_6_none_0.9"
    ms = ([1536] * 8 if njobs == 8 else [96, 2048, 32, 1024, 544][:njobs])
    infos = [qp.mem_op.dummy_linear_info(k, m, qstr, seed=50 + i, codebook_seed=2) for i, m in enumerate(ms)]
    layers = [qp.make_linear_from_info(qstr, info).cuda() for info in infos]
    qp.share_codebooks(layers)
    assert [len(g) for g in qp.linear.launch_groups(layers)] == [njobs]
    for n in (1, 5):
        x = torch.randn(n, k, generator=torch.Generator().manual_seed(njobs * 10 + n)).half()
        ys = qp.multi_gemv(layers, x.cuda())
        for info, y, m in zip(infos, ys, ms):
            _check_gemv(y.cpu().numpy(), _oracle_weight(oracle, qstr, info, m, k), x.numpy(), oracle)


def test_early_and_late_staging_agree(qp):
    """When all jobs of a launch share x and the codebook tensor, the kernel stages them from PRELOADED kernel arguments
    (before its argument block has arrived); with distinct codebook tensors of equal content it stages them the ordinary
    way.  Same arithmetic either way: bit-identical outputs."""
    k = 4096
    for qstr, ms in (("tcomb_6_7_0.5_none_0.9", [1024, 512, 512]), ("ldlq_2_8_none_1.0", [512, 1024]), ("tcq_4_none_0.9", [256])):
        layers = [qp.make_linear_from_info(qstr, qp.mem_op.dummy_linear_info(k, m, qstr, seed=m + i, codebook_seed=5)).cuda()
                  for i, m in enumerate(ms)]
        for n in (1, 2, 8, 11):
            x = torch.randn(n, k, generator=torch.Generator().manual_seed(n)).cuda()
            assert qp.share_codebooks(layers) == 1          # one tensor -> early staging (n <= 8)
            shared = [y.clone() for y in qp.multi_gemv(layers, x)]
            for layer in layers:                            # equal values, distinct tensors -> ordinary staging
                name = "tlut" if hasattr(layer, "tlut") else "lut"
                t = getattr(layer, name)
                if isinstance(t, torch.nn.Parameter):
                    t.data = t.data.clone()
                else:
                    setattr(layer, name, t.clone())
            if len(layers) > 1:
                assert len({getattr(l, "tlut" if hasattr(l, "tlut") else "lut").data_ptr() for l in layers}) == len(layers)
            for a, b in zip(shared, qp.multi_gemv(layers, x)):
                if n < 4:
                    assert torch.equal(a, b), (qstr, n)
                else:  # batched kernel (csrc/tc_gemm.h, batch >= 4): K split over up to 32 workgroups, fp32 atomics -> summation order varies
                    assert torch.allclose(a, b, rtol=1e-5, atol=1e-5 * float(a.abs().max())), (qstr, n)


def test_prezero_and_out_zeroed(qp, oracle):
    """A multi-job launch can pre-zero the output of a later split-K launch (gate|up zeroes down_proj's out)."""
    qstr = "tcomb_6_7_0.5_none_0.9"
    gate, up = (qp.make_linear_from_info(qstr, qp.mem_op.dummy_linear_info(4096, 14336, qstr, seed=s)).cuda()
                for s in (1, 2))
    info = qp.mem_op.dummy_linear_info(14336, 4096, qstr, seed=3)
    down = qp.make_linear_from_info(qstr, info).cuda()
    x = torch.randn(1, 4096, generator=torch.Generator().manual_seed(1)).cuda()
    xd = torch.randn(1, 14336, generator=torch.Generator().manual_seed(2)).half()
    out = torch.full((1, 4096), float("nan"), device="cuda")
    qp.multi_gemv([gate, up], x, prezero=out)
    assert torch.count_nonzero(out).item() == 0
    (y,) = qp.multi_gemv([down], xd.cuda(), outs=[out], outs_zeroed=True)
    assert y.data_ptr() == out.data_ptr()
    W = _oracle_weight(oracle, qstr, info, 4096, 14336)
    _check_gemv(y.cpu().numpy(), W, xd.numpy(), oracle)
    assert torch.equal(y, down._gemv(xd.cuda(), 1))


RAGGED = [(32, 64), (32, 32), (96, 160), (160, 96), (32, 4096), (4096, 32), (64, 14336), (2048, 2560), (992, 1056)]


@pytest.mark.parametrize("m,k", RAGGED)
def test_ragged_shapes(qp, oracle, m, k):
    """Partial steps (k/32 not a multiple of 4), single supertile rows, more waves than steps, tiny and long K."""
    gen = torch.Generator().manual_seed(m * 7 + k)
    cases = ["tcq_6_none_0.9", "tcq_7_none_0.9", "ldlq_1_4_none_1.0", "ldlq_2_9_none_1.0"]
    if k % 64 == 0:
        cases.append("tcomb_6_7_0.5_none_0.9")
    for qstr in cases:
        if "ldlq_2_9" in qstr and (9 * k) % 64:
            continue
        info = qp.mem_op.dummy_linear_info(k, m, qstr, seed=m + k)
        layer = qp.make_linear_from_info(qstr, info).cuda()
        W = _oracle_weight(oracle, qstr, info, m, k)
        assert np.array_equal(_bits(layer.get_weight()), W.view(np.uint16)), qstr
        for n in (1, 2, 8):
            x = torch.randn(n, k, generator=gen).half()
            y = layer._gemv(x.cuda(), n)
            # (a few-row VQ / SQ layer in tensor-core packing answers batch <= 8 from its SIMT-order twin: fp16 output)
            twin = isinstance(layer, qp.VQLinearPackTensorCore) and getattr(layer, "_simt_qweight", None) is not None
            assert y.dtype == (torch.float16 if twin else torch.float32)
            _check_gemv(y.cpu().numpy(), W, x.numpy(), oracle, fp16_out=twin)


@pytest.mark.parametrize("m,k", RAGGED)
def test_ragged_shapes_batched(qp, oracle, m, k):
    """The same ragged shapes through the lockstep skinny-GEMM kernels (batches 9..16: csrc/tc_gemm.h; 17..128: csrc/tc_gemm16.h —
    16 real rows of W per MFMA): partial steps, row groups with dead waves (m / 32 not a multiple of 8), K ranges shorter than the
    K split, batch rows that do not fill a group of 16; then the epilogue forms at a wide batch — per-row scale, output scale,
    residual add — on the multi-job entry point."""
    gen = torch.Generator().manual_seed(m * 11 + k)
    cases = ["tcq_6_none_0.9", "ldlq_2_8_none_1.0"]
    if k % 64 == 0:
        cases.append("tcomb_6_7_0.5_none_0.9")
    for qstr in cases:
        info = qp.mem_op.dummy_linear_info(k, m, qstr, seed=m + k + 1)
        layer = qp.make_linear_from_info(qstr, info).cuda()
        W = _oracle_weight(oracle, qstr, info, m, k)
        for n in (12, 17, 40, 128):
            x = torch.randn(n, k, generator=gen).half()
            y = layer._gemv(x.cuda(), n)
            assert y.dtype == torch.float32 and tuple(y.shape) == (n, m)
            _check_gemv(y.cpu().numpy(), W, x.numpy(), oracle)
        n = 33
        x = torch.randn(n, k, generator=gen).half()
        ws = (torch.rand(m, generator=gen) + 0.5).half()
        base = torch.randn(n, m, generator=gen)
        out = base.clone().cuda()
        qp.multi_gemv([layer], x.cuda(), outs=[out], wscales=[ws.cuda()], oscale=0.75, accumulate=True)
        ref, scale = oracle.gemv(W, x.numpy())
        want = base.numpy().astype(np.float64) + ref * ws.float().numpy().astype(np.float64)[None, :] * 0.75
        tol = GEMV_RTOL_ABS * scale * ws.float().numpy()[None, :] * 0.75 + 1e-6 * np.abs(want) + 1e-30
        assert np.all(np.abs(out.cpu().numpy().astype(np.float64) - want) <= tol), qstr


def test_random_shapes_fuzz(qp, oracle):
    """48 seeded random (family, m, k, batch) cases against the oracle: whatever the launch planner (batch <= 8) or the lockstep
    kernels' K split and batch slices (batch > 8) make of an unusual shape must still be the same GEMV."""
    import random
    rng = random.Random(5)
    fams = ["tcq_3_none_0.9", "tcq_5_none_0.9", "tcq_6_none_0.9", "tcq_8_none_0.9", "tcq_10_none_0.9", "tcomb_3_4_0.5_none_0.9",
            "tcomb_6_7_0.5_none_0.9", "tcomb_9_10_0.5_none_0.9", "ldlq_1_3_none_1.0", "ldlq_1_8_none_1.0", "ldlq_2_6_none_1.0", "ldlq_2_11_none_1.0"]
    done = 0
    while done < 48:
        qstr = rng.choice(fams)
        m = 32 * rng.choice((1, 2, 3, 5, 8, 9, 17, 31, 40, 64, 96))
        k = 64 * rng.choice((1, 2, 3, 5, 8, 13, 16, 33, 64, 100, 128))
        n = rng.choice((1, 2, 3, 5, 8, 9, 13, 16, 17, 33, 48, 64, 65, 100, 128))
        if m * k * n > 3e8:
            continue
        if qstr.startswith("ldlq_2_") and (int(qstr.split("_")[2]) * k) % 64:
            continue
        info = qp.mem_op.dummy_linear_info(k, m, qstr, seed=1000 + done)
        layer = qp.make_linear_from_info(qstr, info).cuda()
        if n > layer.max_fused_batch:
            continue
        W = _oracle_weight(oracle, qstr, info, m, k)
        x = torch.randn(n, k, generator=torch.Generator().manual_seed(done)).half()
        y = layer._gemv(x.cuda(), n)
        try:
            _check_gemv(y.float().cpu().numpy(), W, x.numpy(), oracle, fp16_out=y.dtype == torch.float16)
        except AssertionError as e:
            raise AssertionError(f"{qstr} m {m} k {k} n {n}: {e}") from e
        done += 1


def test_unsupported_shapes_raise(qp):
    with pytest.raises(AttributeError):
        qp.ops.get_op("decompress_gemm_tcq_48_1_64_9_6")      # m % 32
    with pytest.raises(AttributeError):
        qp.ops.get_op("decompress_gemm_tcq_64_0_64_9_6")      # n = 0
    with pytest.raises(AttributeError):
        qp.ops.get_op("decompress_gemm_tcq_combt_64_1_96_9_6_7")  # combt needs k % 64
    op = qp.ops.get_op("decompress_gemm_tcq_64_2_64_9_6")
    with pytest.raises(RuntimeError):   # x has the wrong batch
        op(torch.zeros(16 * 48, dtype=torch.int16, device="cuda"), torch.zeros(1, 64, device="cuda"),
           torch.zeros(512, 2, dtype=torch.float16, device="cuda"))


@pytest.mark.parametrize("qstr,k,m", [("tcq_6_none_0.9", 8192, 57344), ("tcomb_6_7_0.5_none_0.9", 28672, 8192),
                                      ("ldlq_2_8_none_1.0", 8192, 10240)])
def test_llama70b_shapes(qp, oracle, qstr, k, m):
    """Largest shapes of BASELINE.json configs[4] (70B: fused up|gate 57344x8192, down 8192x28672, qkv 10240x8192)."""
    info = qp.mem_op.dummy_linear_info(k, m, qstr, seed=1)
    layer = qp.make_linear_from_info(qstr, info).cuda()
    x = torch.randn(1, k, generator=torch.Generator().manual_seed(4)).half()
    y = layer._gemv(x.cuda(), 1).cpu().numpy()
    W = _oracle_weight(oracle, qstr, info, m, k)
    _check_gemv(y, W, x.numpy(), oracle)
    rows = slice(0, m, 997)   # dequant spot rows (bit-exact)
    assert np.array_equal(_bits(layer.get_weight()[rows]), W[rows].view(np.uint16))


@pytest.mark.parametrize("qstr,k,m", [("tcomb_6_7_0.5_none_0.9", 4096, 4096), ("tcq_8_none_0.9", 4096, 1024),
                                      ("ldlq_1_4_none_1.0", 4096, 2048), ("ldlq_2_12_none_1.0", 4096, 1024),
                                      ("tcq_6_none_0.9", 14336, 4096), ("tcq_6_none_0.9", 8192, 1024),
                                      ("tcomb_6_7_0.5_none_0.9", 8192, 8192), ("ldlq_2_8_none_1.0", 4096, 1024)])
def test_fused_skinny_gemm_batch_9_to_128(qp, oracle, qstr, k, m):
    """Batches 9..128 run in ONE fused launch: a decoded step feeds 1 / 2 / 4 / 8 groups of 16 batch rows, the MFMA's 16 rows being 16
    real rows of W after the lane-pair exchange (csrc/tc_gemm16.h, round 5; rounds 2-4: 64 at most, 65..128 as two passes) — SURVEY N1 /
    north_star "batched dequant-then-GEMM tensor path", Llama-8B and 70B shapes.  Beyond the module's max_fused_batch (128; 64 where a
    128 KiB codebook image leaves the LDS no room for more x tiles) the module decodes to fp16 and calls the fp16 GEMM, as the reference
    does for bs > 8 (lib/linear/tcq_linear.py:75-84)."""
    info = qp.mem_op.dummy_linear_info(k, m, qstr, seed=5)
    layer = qp.make_linear_from_info(qstr, info).cuda()
    assert layer.max_fused_batch == (64 if qstr.startswith("ldlq_2_12") else 128)
    W = _oracle_weight(oracle, qstr, info, m, k)
    gen = torch.Generator().manual_seed(9)
    # (from 32 on a few-row launch runs as two slices of the batch; 65 / 100 / 128: one pass where rounds 2-4 needed two)
    for n in (9, 13, 16, 17, 31, 32, 33, 37, 50, 64, 65, 100, 128):
        if n > layer.max_fused_batch:
            continue
        x = torch.randn(n, k, generator=gen).half()
        y = layer(x.cuda().float())
        assert y.dtype == torch.float32 and tuple(y.shape) == (n, m)
        _check_gemv(y.cpu().numpy(), W, x.numpy(), oracle)


def test_batch_slices_of_a_multi_job_launch(qp, oracle):
    """q | k | v at a wide batch: the lockstep kernel's launch carries every projection twice, once per slice of the batch
    (csrc/qpal_capi.hip slice_gemm_batch: 3 jobs -> 6), each slice writing its own rows of the [n, m] outputs."""
    k, qstr = 4096, "tcomb_6_7_0.5_none_0.9"
    infos = [qp.mem_op.dummy_linear_info(k, m, qstr, seed=40 + i, codebook_seed=11) for i, m in enumerate((1024, 256, 256))]
    mods = [qp.make_linear_from_info(qstr, info).cuda() for info in infos]
    qp.share_codebooks(mods)
    Ws = [_oracle_weight(oracle, qstr, info, m, k) for info, m in zip(infos, (1024, 256, 256))]
    gen = torch.Generator().manual_seed(12)
    for n in (32, 40, 64, 96, 128):
        x = torch.randn(n, k, generator=gen).half()
        ys = qp.multi_gemv(mods, x.cuda())
        for y, W, m in zip(ys, Ws, (1024, 256, 256)):
            assert tuple(y.shape) == (n, m)
            _check_gemv(y.cpu().numpy(), W, x.numpy(), oracle)


def test_mixed_kv_projections_share_one_launch(qp, oracle):
    """q, k, v of a mixed-scheme model have different bit widths: single-stream TCQ layers of one codebook size go out as
    ONE launch (any-KV kernel, one codebook image) and agree with their own single launches and with the oracle."""
    k = 4096
    for kvs, ms in (((3, 6, 8), (4096, 1024, 1024)), ((2, 7), (14336, 14336)), ((8, 9, 10), (512, 256, 256)), ((9, 10), (1024, 32))):
        layers = []
        for i, (kv, m) in enumerate(zip(kvs, ms)):
            qstr = f"tcq_{kv}_none_0.9"
            info = qp.mem_op.dummy_linear_info(k, m, qstr, seed=100 + i, codebook_seed=3 if kv <= 8 else 30 + kv)
            layers.append((qp.make_linear_from_info(qstr, info).cuda(), qstr, info))
        mods = [l for l, _, _ in layers]
        qp.share_codebooks(mods)
        groups = qp.linear.launch_groups(mods, mixed_kv=True)
        sizes = sorted(len(g) for g in groups)
        # S = 9 (KV <= 8), 10 (KV 9) and 11 (KV 10) are different codebooks: (8, 9, 10) stays three launches
        assert sizes == ([len(kvs)] if max(kvs) <= 8 else [1] * len(kvs)), (kvs, sizes)
        for n in (1, 3, 8):
            x = torch.randn(n, k, generator=torch.Generator().manual_seed(n)).cuda()
            ys = qp.multi_gemv(mods, x)
            for (mod, qstr, info), y in zip(layers, ys):
                ref = mod._gemv(x, n)   # (its own launch may cut K differently: summation order, not arithmetic)
                assert torch.allclose(y, ref, rtol=1e-4, atol=1e-4 * float(ref.abs().max())), (kvs, qstr, n)
            if n == 3:
                for (mod, qstr, info), y, m in zip(layers, ys, ms):
                    _check_gemv(y.cpu().numpy(), _oracle_weight(oracle, qstr, info, m, k), x.half().cpu().numpy(), oracle)
        # batch 12 falls back to one launch per codec (the any-KV kernel has no two-batch-group variant)
        x = torch.randn(12, k, generator=torch.Generator().manual_seed(12)).cuda()
        for (mod, _, _), y in zip(layers, qp.multi_gemv(mods, x)):
            assert torch.allclose(y, mod._gemv(x, 12), rtol=1e-4, atol=1e-4 * float(y.abs().max()))


def test_tcomb_and_tcq_projections_share_one_launch(qp, oracle):
    """Mixed-scheme q | k | v where some projections are column-split (tcomb_x_y: two streams, two bit widths) and some plain
    TCQ: one any-KV launch (per-job KV and KV2; the waves of a row pick their decode loop by the stream their chunk lies in),
    equal to the layers' own launches and to the oracle.  A pair the any-KV kernel cannot hold (KV2 = 9 at S = 9) stays apart."""
    k = 4096
    cases = ((("tcomb_3_4_0.5_none_0.9", "tcq_6_none_0.9", "tcq_3_none_0.9"), (4096, 1024, 1024)),
             (("tcomb_6_7_0.5_none_0.9", "tcomb_2_3_0.5_none_0.9"), (14336, 14336)),
             (("tcq_5_none_0.9", "tcomb_7_8_0.5_none_0.9", "tcomb_4_5_0.5_none_0.9", "tcq_8_none_0.9"), (512, 2048, 96, 32)))
    for qstrs, ms in cases:
        layers = []
        for i, (qstr, m) in enumerate(zip(qstrs, ms)):
            info = qp.mem_op.dummy_linear_info(k, m, qstr, seed=200 + i, codebook_seed=3)
            layers.append((qp.make_linear_from_info(qstr, info).cuda(), qstr, info))
        mods = [l for l, _, _ in layers]
        qp.share_codebooks(mods)
        assert [len(g) for g in qp.linear.launch_groups(mods, mixed_kv=True)] == [len(mods)], qstrs
        for n in (1, 5, 8):
            x = torch.randn(n, k, generator=torch.Generator().manual_seed(40 + n)).cuda()
            ys = qp.multi_gemv(mods, x)
            for (mod, qstr, info), y, m in zip(layers, ys, ms):
                ref = mod._gemv(x, n)
                assert torch.allclose(y, ref, rtol=1e-4, atol=1e-4 * float(ref.abs().max())), (qstrs, qstr, n)
                if n != 8:
                    _check_gemv(y.cpu().numpy(), _oracle_weight(oracle, qstr, info, m, k), x.half().cpu().numpy(), oracle)
        # fused epilogue and a pre-zeroed split-K output through the same launch
        x = torch.randn(1, k, generator=torch.Generator().manual_seed(7)).cuda()
        wsc = [(0.5 + torch.rand(m, generator=torch.Generator().manual_seed(m))).half().cuda() for m in ms]
        outs = [torch.zeros(1, m, device="cuda") for m in ms]
        ys = qp.multi_gemv(mods, x, outs=outs, outs_zeroed=True, wscales=wsc, oscale=0.25)
        for (mod, _, _), y, w in zip(layers, ys, wsc):
            r0 = mod._gemv(x, 1)
            rt = 2e-3 if r0.dtype == torch.float16 else 1e-4   # (fp16: the layer answered from its SIMT-order twin)
            ref = r0.float() * w.float() * 0.25
            assert torch.allclose(y, ref, rtol=rt, atol=rt * float(ref.abs().max()))
    info = qp.mem_op.dummy_linear_info(k, 256, "tcomb_8_9_0.5_none_0.9", seed=1, codebook_seed=3)
    far = qp.make_linear_from_info("tcomb_8_9_0.5_none_0.9", info).cuda()
    assert sorted(len(g) for g in qp.linear.launch_groups([mods[0], far], mixed_kv=True)) == [1, 1]


def test_mixed_family_projections_one_launch_per_family(qp, oracle):
    """Mixed-FAMILY q | k | v (the reference's MSQ results give every projection its own quantizer): TCQ layers of one codebook
    size (any KV, column-split tcomb layers too) share ONE any-KV launch; every VQ/SQ codec is a launch of its own (a kernel that
    switched families per job measured slower than separate launches and was removed in round 5)."""
    k = 4096
    cases = ((("tcq_3_none_0.9", "ldlq_2_6_none_1.0", "tcomb_4_5_0.5_none_0.9"), (4096, 1024, 1024), [1, 2]),
             (("ldlq_2_8_none_1.0", "tcq_7_none_0.9", "ldlq_1_4_none_1.0", "ldlq_1_8_none_1.0", "tcomb_6_7_0.5_none_0.9"),
              (14336, 512, 4096, 96, 2048), [1, 1, 1, 2]),
             (("ldlq_2_3_none_1.0", "ldlq_1_7_none_1.0", "ldlq_2_5_none_1.0"), (1024, 1024, 4096), [1, 1, 1]),
             (("tcq_4_none_0.9", "ldlq_2_10_none_1.0", "ldlq_1_6_none_1.0", "ldlq_2_4_none_1.0"), (1024, 512, 512, 256), [1, 1, 1, 1]))
    for qstrs, ms, want_sizes in cases:
        layers = []
        for i, (qstr, m) in enumerate(zip(qstrs, ms)):
            info = qp.mem_op.dummy_linear_info(k, m, qstr, seed=300 + i, codebook_seed=3)
            layers.append((qp.make_linear_from_info(qstr, info).cuda(), qstr, info))
        mods = [l for l, _, _ in layers]
        qp.share_codebooks(mods)
        assert sorted(len(g) for g in qp.linear.launch_groups(mods, mixed_kv=True)) == want_sizes, qstrs
        for n in (1, 4, 8):
            x = torch.randn(n, k, generator=torch.Generator().manual_seed(60 + n)).cuda()
            ys = qp.multi_gemv(mods, x)
            for (mod, qstr, info), y, m in zip(layers, ys, ms):
                ref = mod._gemv(x, n)
                # (a few-row VQ / SQ layer answers a single launch from its SIMT-order twin: fp16 output)
                rt = 2e-3 if ref.dtype == torch.float16 else 1e-4
                assert torch.allclose(y, ref.float(), rtol=rt, atol=rt * float(ref.abs().max())), (qstrs, qstr, n)
                if n == 4:
                    _check_gemv(y.cpu().numpy(), _oracle_weight(oracle, qstr, info, m, k), x.half().cpu().numpy(), oracle)
        x = torch.randn(1, k, generator=torch.Generator().manual_seed(9)).cuda()
        wsc = [(0.5 + torch.rand(m, generator=torch.Generator().manual_seed(m))).half().cuda() for m in ms]
        buf = torch.zeros(1, sum(ms), device="cuda")
        spare = torch.full((1, 4096), 7.0, device="cuda")
        ys = qp.multi_gemv(mods, x, outs=list(buf.split(list(ms), dim=1)), outs_zeroed=True, wscales=wsc, oscale=0.25, prezero=spare)
        assert float(spare.abs().max()) == 0.0
        for (mod, _, _), y, w in zip(layers, ys, wsc):
            r0 = mod._gemv(x, 1)
            rt = 2e-3 if r0.dtype == torch.float16 else 1e-4   # (fp16: the layer answered from its SIMT-order twin)
            ref = r0.float() * w.float() * 0.25
            assert torch.allclose(y, ref, rtol=rt, atol=rt * float(ref.abs().max()))


class _RefStyleTCQ(torch.nn.Module):
    """The reference's QTIPLinearTCQ.forward pattern (lib/linear/tcq_linear.py:64-85): op looked up by name inside forward."""

    def __init__(self, layer):
        super().__init__()
        self.l = layer

    def forward(self, inp):
        l = self.l
        x = inp.view(-1, l.in_features)
        bs = x.shape[0]
        if bs <= 8:
            y = getattr(torch.ops.ours_lib, f"decompress_gemm_tcq_{l.out_features}_{bs}_{l.in_features}_{l.tlut_bits}_{l.KV}")(
                l.trellis, x.to(torch.float16), l.tlut)
        else:
            dq = getattr(torch.ops.ours_lib, f"decompress_tcq_{l.tlut_bits}_{l.KV}")(l.trellis, l.tlut, l.out_features, l.in_features)
            y = x.to(dq.dtype) @ dq.T
        return y.view(*inp.shape[:-1], l.out_features).to(inp.dtype)


def test_reference_calling_pattern_under_torch_compile_and_graph_capture(qp, oracle):
    """eval/measure_latency.py:220-225 compiles the decode step (fullgraph=True) and runs it under CUDA graphs: the ops must
    trace (register_fake), compile without graph breaks (backend aot_eager — no Triton in this build) and be capturable with
    the reference's own calling pattern."""
    qstr, k, m = "tcq_6_none_0.9", 1024, 512
    info = qp.mem_op.dummy_linear_info(k, m, qstr, seed=3, device="cpu")
    layer = qp.make_linear_from_info(qstr, info).cuda()
    W = _oracle_weight(oracle, qstr, info, m, k)
    ref_mod = _RefStyleTCQ(layer)
    ours_c = torch.compile(layer, fullgraph=True, backend="aot_eager")
    ref_c = torch.compile(ref_mod, fullgraph=True, backend="aot_eager")
    gen = torch.Generator().manual_seed(1)
    for n in (1, 4, 12):
        x = torch.randn(n, k, generator=gen).half()
        y_e = layer(x.cuda().float())
        y_o = ours_c(x.cuda().float())
        y_r = ref_c(x.cuda().float())
        assert torch.equal(y_e, y_o)
        if n <= 8:
            assert torch.equal(y_e, y_r)
            _check_gemv(y_r.cpu().numpy(), W, x.numpy(), oracle)
        else:  # the reference's bs > 8 path: decode + fp16 GEMM
            ref = (x.float() @ torch.from_numpy(W).float().T).numpy()
            assert np.allclose(y_r.cpu().numpy(), ref, rtol=2e-2, atol=2e-2 * np.abs(ref).max())
    # capture + replay of the compiled reference-style module
    xs = torch.randn(1, k, generator=gen).half().cuda().float()
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        ref_c(xs)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            out = ref_c(xs)
    xn = torch.randn(1, k, generator=gen).half()
    xs.copy_(xn.cuda().float())
    g.replay()
    torch.cuda.synchronize()
    _check_gemv(out.cpu().numpy(), W, xn.numpy(), oracle)


@pytest.mark.parametrize("qstr,k,m,part", [("comb_6_7_0.5_none_0.9", 512, 256, (96, 160)), ("comb_7_8_0.5_none_0.9", 1024, 128, (32, 96)),
                                            ("tcomb_6_7_0.5_none_0.9", 512, 256, (192, 320)), ("tcomb_3_4_0.5_none_0.9", 1024, 64, (768, 256))])
def test_comb_layers_with_unequal_parts(qp, oracle, qstr, k, m, part):
    """Comb / Combt with unequal halves cannot use the fused two-stream kernel (use_comb_kernel is False): the modules run two
    single-stream ops and concatenate / add, like the reference (lib/linear/comb_linear.py:91-102, 234-245)."""
    info = qp.mem_op.dummy_linear_info(k, m, qstr, seed=3, part=part)
    layer = qp.make_linear_from_info(qstr, info).cuda()
    assert layer.use_comb_kernel is False
    kv1, kv2 = info["KV"]
    S = info["tlut_bits"]
    tl = info["tlut"].numpy()
    if "out_part" in info:
        W = np.concatenate([oracle.tcq_dequant(info["trellis1"].numpy(), tl, part[0], k, S, kv1),
                            oracle.tcq_dequant(info["trellis2"].numpy(), tl, part[1], k, S, kv2)], axis=0)
    else:
        W = np.concatenate([oracle.tcq_dequant(info["trellis1"].numpy(), tl, m, part[0], S, kv1),
                            oracle.tcq_dequant(info["trellis2"].numpy(), tl, m, part[1], S, kv2)], axis=1)
    assert np.array_equal(_bits(layer.get_weight()), W.view(np.uint16))
    gen = torch.Generator().manual_seed(4)
    for n in (1, 5, 16):
        x = torch.randn(n, k, generator=gen).half()
        y = layer(x.cuda().float())
        # the column-split form adds two fp32 partial results: one more fp32 rounding than a single kernel
        _check_gemv(y.cpu().numpy(), W, x.numpy(), oracle)
    # above the fused batch these layers have no multi-job form: passes of their own launches up to max_chunked_batch (this range
    # used to recurse forever: forward -> multi_gemv -> forward), decode + GEMM beyond
    for n in (65, 128, 200):
        x = torch.randn(n, k, generator=gen).half()
        y = layer(x.cuda())
        assert y.dtype == torch.float16 and tuple(y.shape) == (n, m)
        rows = np.array([0, 63, 64, n - 1])
        _check_gemv(y[rows].float().cpu().numpy(), W, x[rows].numpy(), oracle, fp16_out=True)


def test_comb_layer_row_shards(qp, oracle):
    """shard_linear_info of a CombLinearTCQ: a row shard takes its rows from whichever halves it overlaps (possibly one);
    the shards' outputs concatenate to the full layer's."""
    qstr, k, m = "comb_6_7_0.5_none_0.9", 512, 256
    info = qp.mem_op.dummy_linear_info(k, m, qstr, seed=8)
    full = qp.make_linear_from_info(qstr, info).cuda()
    x = torch.randn(2, k, generator=torch.Generator().manual_seed(1)).cuda()
    for world in (2, 4, 3):
        parts = []
        for r in range(world):
            sh = qp.shard.shard_linear_info(info, r, world)
            layer = qp.CombLinearTCQ.gen_layer_from_info(sh).cuda()
            assert layer.out_features == qp.shard.shard_rows(m, world)[r]
            parts.append(layer(x))
        y = torch.cat(parts, dim=1)
        ref = full(x)
        assert torch.allclose(y, ref, rtol=1e-5, atol=1e-5 * float(ref.abs().max())), world


@pytest.mark.parametrize("qstr,k,m,n", [("tcomb_6_7_0.5_none_0.9", 4096, 4096, 256), ("tcq_6_none_0.9", 14336, 4096, 200),
                                        ("ldlq_2_12_none_1.0", 4096, 1024, 96), ("ldlq_1_4_none_1.0", 4096, 2048, 130),
                                        ("tcomb_6_7_0.5_none_0.9", 4096, 4096, 257), ("tcq_6_none_0.9", 14336, 4096, 300),
                                        ("ldlq_2_12_none_1.0", 4096, 1024, 129), ("ldlq_1_4_none_1.0", 4096, 2048, 512),
                                        ("tcomb_6_7_0.5_none_0.9", 4096, 14336, 1024)])
def test_module_path_above_the_fused_batch(qp, oracle, qstr, k, m, n):
    """Above max_fused_batch the module runs passes of the fused kernel up to max_chunked_batch (2 x 128 rows) and beyond that the
    path the perplexity eval takes (bs = 8192 there: eval_qdict.py:17-38): W decoded to fp16 (bit-exact, checked elsewhere)
    times the fp16 GEMM, as the reference does for bs > 8 (lib/linear/tcq_linear.py:75-84).  Both against the oracle's fp64 GEMM
    on the oracle's W: fp32 accumulation + ONE fp16 rounding of the output."""
    info = qp.mem_op.dummy_linear_info(k, m, qstr, seed=11)
    layer = qp.make_linear_from_info(qstr, info).cuda()
    assert n > layer.max_fused_batch
    assert layer.max_chunked_batch == 2 * layer.max_fused_batch
    W = _oracle_weight(oracle, qstr, info, m, k)
    x = torch.randn(n, k, generator=torch.Generator().manual_seed(n)).half()
    y = layer(x.cuda())
    assert y.dtype == torch.float16 and tuple(y.shape) == (n, m)
    rows = np.arange(0, n, max(1, n // 16))  # the oracle's float64 GEMV on 16 batch rows spread over the batch
    _check_gemv(y[rows].float().cpu().numpy(), W, x[rows].numpy(), oracle, fp16_out=True)


L70 = {"qkv": (8192, 10240), "o": (8192, 8192), "ug": (8192, 57344), "d": (28672, 8192)}


@pytest.mark.parametrize("kind", sorted(L70))
@pytest.mark.parametrize("rank", [0, 5])
def test_llama70b_row_shard_of_8(qp, oracle, kind, rank):
    """BASELINE configs[4] in its sharded form: rank r of 8 of the Llama-3.1-70B q|k|v, o, gate|up and down layers (tcq_6,
    3.0 b/w; shard_linear_info: rows of W in units of one supertile row — /root/reference/lib/utils/mem_op.py:65-95 for the
    shapes) through the HIP path at batch 1 and batch 16, against the oracle on that shard."""
    qstr = "tcq_6_none_0.9"
    k, m = L70[kind]
    full = qp.mem_op.dummy_linear_info(k, m, qstr, seed=70 + len(kind), device="cpu")
    info = qp.shard.shard_linear_info(full, rank, 8)
    r0, r1 = qp.shard.shard_bounds(m, 8, rank)
    ms = r1 - r0
    assert info["out_features"] == ms == m // 8
    per_row16 = k // 16
    assert torch.equal(info["trellis"], full["trellis"][r0 // 16 * per_row16: r1 // 16 * per_row16])
    layer = qp.make_linear_from_info(qstr, info).cuda()
    W = _oracle_weight(oracle, qstr, info, ms, k)
    if kind == "o":  # and the shard IS rows [r0, r1) of the unsharded layer's weight
        Wf = _oracle_weight(oracle, qstr, full, m, k)
        assert np.array_equal(Wf[r0:r1].view(np.uint16), W.view(np.uint16))
    assert np.array_equal(_bits(layer.get_weight()), W.view(np.uint16))
    gen = torch.Generator().manual_seed(rank)
    for n in (1, 16):
        x = torch.randn(n, k, generator=gen).half()
        (y,) = qp.multi_gemv([layer], x.cuda())
        assert y.dtype == torch.float32 and tuple(y.shape) == (n, ms)
        _check_gemv(y.cpu().numpy(), W, x.numpy(), oracle)


@pytest.mark.parametrize("qstr,m,k", [("tcomb_6_7_0.5_none_0.9", 8960, 4096), ("tcq_6_none_0.9", 8960, 4096), ("tcq_5_none_0.9", 5120, 4096),
                                      ("ldlq_1_4_none_1.0", 8960, 4096), ("ldlq_2_9_none_1.0", 5120, 4096),
                                      ("tcq_6_none_0.9", 28672, 2048), ("tcq_4_none_0.9", 57344, 1024)])
def test_pair_mode_rows_shared_by_two_workgroups(qp, oracle, qstr, m, k):
    """Pair mode of the GEMV planner (csrc/tc_kernels.h TcParams, qpal_capi.hip plan_launch): two projections of one input whose
    outputs the caller declares zeroed — 2 x 280 rows at 4 rows per workgroup pair up as 7 rows per two workgroups (160 items
    instead of 140), 2 x 160 rows at 2 rows per workgroup as 3 per two, 2 x 896 rows at 8 per workgroup (Llama-70B's gate | up
    geometry) as 14 per two with TWO shared rows, 2 x 1792 at 16 as 28 per two with four; the shared rows' halves meet by
    atomics.  Against the oracle, against the unpaired launch of the same layers, bit-identical between repeats, and with
    accumulate."""
    import subprocess, textwrap
    infos = [qp.mem_op.dummy_linear_info(k, m, qstr, seed=70 + i, codebook_seed=13) for i in range(2)]
    mods = [qp.make_linear_from_info(qstr, info).cuda() for info in infos]
    qp.share_codebooks(mods)
    Ws = [_oracle_weight(oracle, qstr, info, m, k) for info in infos]
    gen = torch.Generator().manual_seed(21)
    for n in (1, 3):
        x = torch.randn(n, k, generator=gen).half()
        plain = qp.multi_gemv(mods, x.cuda())
        outs = [torch.zeros(n, m, dtype=torch.float32, device="cuda") for _ in mods]
        got = qp.multi_gemv(mods, x.cuda(), outs=outs, outs_zeroed=True)
        again = qp.multi_gemv(mods, x.cuda(), outs=[torch.zeros_like(o) for o in outs], outs_zeroed=True)
        for y, y2, p_, W in zip(got, again, plain, Ws):
            _check_gemv(y.cpu().numpy(), W, x.numpy(), oracle)
            assert torch.equal(y, y2)  # two adders per element: order-independent
            assert torch.allclose(y, p_, rtol=1e-4, atol=1e-4 * float(p_.abs().max()))
        res = [torch.randn(n, m, generator=gen).cuda() for _ in mods]
        acc = [r.clone() for r in res]
        qp.multi_gemv(mods, x.cuda(), outs=acc, accumulate=True)
        for a, r, p_ in zip(acc, res, plain):
            assert torch.allclose(a, r + p_, rtol=1e-4, atol=1e-4 * float(p_.abs().max()))
    # the planner really let workgroups share rows in these launches (its log line says so), and QPAL_PAIR=0 (QPAL_SHARE=0) does not
    code = textwrap.dedent(f"""
        import torch, qpalette_amd as qp
        infos = [qp.mem_op.dummy_linear_info({k}, {m}, "{qstr}", seed=70 + i, codebook_seed=13) for i in range(2)]
        mods = [qp.make_linear_from_info("{qstr}", info).cuda() for info in infos]
        qp.share_codebooks(mods)
        x = torch.randn(1, {k}).half().cuda()
        qp.multi_gemv(mods, x, outs=[torch.zeros(1, {m}, device="cuda") for _ in mods], outs_zeroed=True)
        torch.cuda.synchronize()
    """)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import re
    busiest = {}
    for pair in ("1", "0"):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, QPAL_PLAN_LOG="1", QPAL_PAIR=pair, PYTHONPATH=root),
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-1500:]
        lines = [l for l in r.stderr.splitlines() if l.startswith("[qpal plan]")]
        assert lines, r.stderr[-1500:]
        if pair == "0":
            assert "shared rows" not in lines[-1], lines
        busiest[pair] = max(int(v) for v in re.findall(r"busiest SIMD (\d+) steps", lines[-1]))
    # sharing rows is a means, not an end: the planner takes it where it lowers the busiest SIMD's steps (round 5: some of these
    # shapes balance as well with whole rows in uneven groups, e.g. 7 rows per workgroup)
    assert busiest["1"] <= busiest["0"], busiest


def test_lane_xor_forms_match_the_shuffle(tmp_path):
    """qpal_common.h lane_xor<MASK> / group_sum / wave_sum / wave_max (DPP quad_perm, row mirrors, gfx950 permlane swaps: what the
    epilogues and the glue kernels' reductions use instead of __shfl_xor's LDS permute) agree with __shfl_xor lane for lane:
    perf/lane_xor_test.hip, compiled here with hipcc and run on the GPU."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "lane_xor_test.bin")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++20", "-I", os.path.join(root, "include"),
                    "-I", os.path.join(root, "q-palette_amd", "csrc"), os.path.join(root, "perf", "lane_xor_test.hip"), "-o", exe],
                   check=True, capture_output=True, timeout=300)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "MISMATCH" not in r.stdout, r.stdout + r.stderr
