"""Incoherence wrapper: Hadamard factors, the one-launch rotation kernel, the fused
`* Wscale * scale` GEMV epilogue and the IncoherentLinear / IncoherentMLP / IncoherentSdpaAttention modules.

CPU part: the generated Paley factors equal the reference's get_hadK outputs (golden), the numpy oracle reproduces
the reference's matmul_hadU / matmul_hadUt golden outputs, host logic and C-ABI argument checks.
GPU part (``-m gpu``): qpal_hadamard vs the oracle — at most ONE fp16 ulp apart (fp32 butterflies in a different
order than the float64 oracle, then the same fp16 roundings); whole modules vs the oracle chain with the
reference's fp16 rounding points — |err| <= 2^-9 * (sum_k |w_k x_k| * Wscale * scale) per output, which covers the
handful of fp16 roundings along the chain (each 2^-11 relative) that the fused fp32 epilogue does not repeat.
"""
import os
import types

import numpy as np
import pytest
import torch

from oracle import incoherent as oi

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _gold():
    return np.load(os.path.join(GOLD, "hadamard.npz"))


# ------------------------------------------------------------------------------------------------ CPU
@pytest.mark.parametrize("K", [12, 20, 28, 36, 60, 108, 140, 52, 116, 124, 156, 172])
def test_generated_factor_equals_reference(K):
    import qpalette_amd as qp
    ref = oi.unpack_hadk(_gold()[f"hadK_{K}"], K)
    mine = qp.hadamard.hadK_matrix(K).numpy().astype(np.float64)
    assert np.array_equal(ref, mine)
    assert np.array_equal(mine @ mine.T, K * np.eye(K))


def test_oracle_matches_reference_matmul_hadU():
    import qpalette_amd as qp
    g = _gold()
    for n in (int(v) for v in g["sizes"]):
        hadK, K = qp.hadamard.get_hadK(n)
        h = None if K == 1 else hadK.numpy().astype(np.float64)
        x = g[f"x_{n}"].astype(np.float64)
        # the reference divides by a float32 sqrt(n): 1e-7 relative
        assert np.allclose(oi.had_blocks(x, n, h), g[f"hadU_{n}"], rtol=0, atol=5e-7), n
        assert np.allclose(oi.had_blocks(x, n, None if h is None else h.T), g[f"hadUt_{n}"], rtol=0, atol=5e-7), n


def test_get_hadK_grammar():
    import qpalette_amd as qp
    had = qp.hadamard
    assert had.get_hadK(4096) == (None, 1)
    for n, K in ((14336, 28), (28672, 28), (3072, 12), (5120, 20), (13824, 108), (8192, 1), (1152, 36)):
        h, k = had.get_hadK(n)
        assert k == K and (h is None) == (K == 1)
    h, _ = had.get_hadK(3072)
    ht, _ = had.get_hadK(3072, transpose=True)
    assert torch.equal(h.T, ht) and not torch.equal(h, ht)  # Paley I factors are not symmetric
    assert had.get_hadK(11008)[1] == 172 and had.get_hadK(13312)[1] == 52  # Williamson factors (Llama-2-7B / -13B sizes)
    with pytest.raises(NotImplementedError):
        had.hadK_matrix(44)  # not one of the reference's factors: register_hadK is the way in
    with pytest.raises(AssertionError):
        had.get_hadK(4097 * 3)
    with pytest.raises(ValueError):
        had.register_hadK(52, torch.ones(52, 52))
    fake = torch.kron(had.hadK_matrix(12), torch.tensor([[1.0, 1.0], [1.0, -1.0]]))  # a 24 x 24 Hadamard matrix
    with pytest.raises(ValueError):
        had.register_hadK(52, fake)


def test_capi_hadamard_argument_checks():
    import ctypes
    import qpalette_amd as qp
    lib = qp._native.lib()
    buf = (ctypes.c_uint16 * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    E_SHAPE, E_PARAM, E_NULL = -1, -2, -3
    call = lib.qpal_hadamard
    assert call(None, p, None, None, None, 1, 64, 64, 1, 0, 1, 1.0, None) == E_NULL
    buf2 = (ctypes.c_uint16 * 64)()
    p2 = ctypes.cast(buf2, ctypes.c_void_p)
    assert call(p, p, None, None, None, 1, 64, 64, 1, 0, 1, 1.0, None) == E_PARAM      # in place is refused
    assert call(p, p2, None, None, None, 1, 64, 64, 12, 0, 1, 1.0, None) == E_NULL     # K > 1 needs hadk
    assert call(p, p2, None, None, None, 1, 96, 96, 1, 0, 1, 1.0, None) == E_SHAPE      # 96 is not a power of two
    assert call(p, p2, None, None, p, 1, 96, 96, 12, 0, 1, 1.0, None) == E_SHAPE        # P = 8 < 16
    assert call(p, p2, None, None, None, 1, 64, 48, 1, 0, 1, 1.0, None) == E_SHAPE      # hd does not divide n
    assert call(p, p2, None, None, None, 1, 64, 64, 1, 7, 1, 1.0, None) == E_PARAM
    assert call(p, p2, None, None, None, 1, 65536, 65536, 1, 0, 1, 1.0, None) == E_SHAPE  # does not fit the LDS


def _cfg(hidden=1024, inter=3584, heads=8, kv_heads=2):
    return types.SimpleNamespace(hidden_size=hidden, intermediate_size=inter, hidden_act="silu",
                                 num_attention_heads=heads, num_key_value_heads=kv_heads, attention_dropout=0.0)


def _info(qp, k, m, qstr, seed, with_scales=True):
    gen = torch.Generator().manual_seed(1000 + seed)
    info = {"quant_info": qp.mem_op.get_quant_info(qstr), "in_features": k, "out_features": m, "dtype": torch.float16,
            "bias": None, "linear_info": qp.mem_op.dummy_linear_info(k, m, qstr, seed=seed, codebook_seed=7)}
    if with_scales:
        info["SU"] = (torch.randint(0, 2, (k,), generator=gen) * 2 - 1).to(torch.float16)
        info["SV"] = (torch.randint(0, 2, (m,), generator=gen) * 2 - 1).to(torch.float16)
        info["Wscale"] = (0.01 + 0.02 * torch.rand(m, generator=gen)).to(torch.float16)
    return info


def test_modules_build_on_cpu():
    import qpalette_amd as qp
    cfg = _cfg()
    H, I = cfg.hidden_size, cfg.intermediate_size
    up, gate, down = _info(qp, H, I, "tcq_4_0_1", 1), _info(qp, H, I, "tcq_4_0_1", 2), _info(qp, I, H, "tcq_4_0_1", 3)
    for merge in (False, True):
        mlp = qp.IncoherentMLP.gen_layer_from_info(cfg, up, gate, down, merge_ug=merge)
        assert mlp.inter_K == 28 and mlp.hidden_K == 1 and mlp.had_left_ug_T is None
        assert tuple(mlp.had_left_dp_T.shape) == (28, 28) and mlp.had_left_dp_T.dtype == torch.float16
        assert torch.equal(mlp.Wscale_ug, torch.cat([up["Wscale"], gate["Wscale"]]))
        assert (mlp.ug_proj is not None) == merge and (mlp.up_proj is None) == merge
        if merge:
            assert mlp.ug_proj.out_features == 2 * I
    kvo = H * cfg.num_key_value_heads // cfg.num_attention_heads
    q, k, v, o = (_info(qp, H, H, "tcq_4_0_1", 4), _info(qp, H, kvo, "tcq_4_0_1", 5), _info(qp, H, kvo, "tcq_4_0_1", 6),
                  _info(qp, H, H, "ldlq_2_7_0_1", 7))
    for flags in ({}, {"merge_qkv": True}, {"merge_qk": True}, {"merge_kv": True}, {"merge_qv": True}):
        attn = qp.IncoherentSdpaAttention.gen_layer_from_info(cfg, 0, q, k, v, o, **flags)
        layers, wscales, blocks = attn._qkv_layout()
        assert sum(l.out_features for l in layers) == H + 2 * kvo == sum(w.numel() for w in wscales)
        assert [b[0] for b in blocks] == (["q", "v", "k"] if flags.get("merge_qv") else ["q", "k", "v"])
        assert isinstance(attn.o_proj, qp.VQLinearPackTensorCore)
    with pytest.raises(AssertionError):
        qp.IncoherentSdpaAttention(cfg, merge_qk=True, merge_kv=True)
    lin = qp.IncoherentLinear.gen_layer_from_info(dict(_info(qp, H, H, "tcomb_3_4_0.5_0_1", 8), hadU=128, hadV=128,
                                                       rot_info="skip_r"), merge_layers=True)
    assert lin.skip_r and not lin.skip_l and lin.K_left == 1 and isinstance(lin.linear, qp.CombtLinearTCQ)
    dummy = qp.IncoherentMLP.gen_layer_from_quantizer_str_and_key(
        types.SimpleNamespace(hidden_size=4096, intermediate_size=14336, hidden_act="silu",
                              _name_or_path="meta-llama/Llama-3.1-8B"),
        None, "tcq_2_0_1", "tcq_2_0_1", "tcq_2_0_1", "0_up", "0_gate", "0_down", merge_ug=True, dummy=True)
    assert dummy.ug_proj.out_features == 28672 and dummy.down_proj.in_features == 14336
    assert isinstance(qp.make_linear({"quant_info": {"quantizer_str": "default"}, "in_features": 64, "out_features": 32,
                                      "linear_info": None}), torch.nn.Linear)


# ------------------------------------------------------------------------------------------------ GPU
gpu = pytest.mark.gpu


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


def _ulp16(v):
    a = np.maximum(np.abs(v), 2.0 ** -14)
    return 2.0 ** (np.floor(np.log2(a)) - 10)


def _assert_ulp(got, want, ulps=1):
    got = got.astype(np.float64)
    err = np.abs(got - want)
    bad = err > ulps * _ulp16(want) * 1.0001
    assert not bad.any(), f"{bad.sum()} of {bad.size} off by more than {ulps} ulp; max err {err.max():.3e}"


def _signs(rng, n):
    return (rng.integers(0, 2, n) * 2 - 1).astype(np.float16)


@gpu
@pytest.mark.parametrize("rows,n,hd", [(1, 64, 64), (3, 128, 128), (1, 1024, 1024), (1, 4096, 4096), (16, 4096, 4096),
                                       (2, 8192, 8192), (1, 32768, 32768), (1, 14336, 14336), (5, 14336, 14336),
                                       (1, 28672, 28672), (3, 3072, 3072), (2, 5120, 5120), (2, 13824, 13824),
                                       (4, 4096, 128), (2, 3584, 448), (3, 2048, 64),
                                       (2, 11008, 11008), (1, 13312, 13312), (1, 7424, 7424), (1, 9984, 9984)])  # K = 172, 52, 116, 156 (Williamson)
@pytest.mark.parametrize("round_mid", [True, False])
def test_rotate_vs_oracle(qp, rows, n, hd, round_mid):
    had = qp.hadamard
    rng = np.random.default_rng(n + rows + hd)
    hadK, K = had.get_hadK(hd)
    hT = None if K == 1 else hadK.T.contiguous()
    x = rng.standard_normal((rows, n)).astype(np.float16)
    su = _signs(rng, n)
    y = had.rotate(torch.from_numpy(x).cuda(), hd=hd, hadK=None if K == 1 else hT.half().cuda(), K=K,
                   su=torch.from_numpy(su).cuda(), post_scale=1.0 / 64, round_mid=round_mid)
    assert y.dtype == torch.float16 and tuple(y.shape) == (rows, n)
    xs = oi.f16(x.astype(np.float64) * su)
    want = oi.f16(oi.had_blocks(xs, hd, None if K == 1 else hT.numpy().astype(np.float64), round_mid=round_mid)) / 64
    if K > 1 and round_mid:
        # the fp16 rounding BETWEEN the two factors: where the fp32 and the float64 butterflies land on different
        # sides of a rounding boundary, one intermediate t_i moves by one of ITS ulps, and the +-1 sum passes that on
        # unchanged — more than an ulp of the output wherever the sum cancels.  Allow it on top of the output's ulp.
        t = oi.f16(oi.wht(xs.reshape(rows, n // hd, K, hd // K)) * float(hd) ** -0.5)
        extra = np.broadcast_to(_ulp16(np.abs(t).max(axis=-2, keepdims=True)), t.shape).reshape(rows, n) / 64
        got = y.cpu().numpy().astype(np.float64)
        if hd > 4096 and K <= 32:
            # large blocks take the one-launch mix-first kernel (csrc/hadamard.hip had_mixfirst_kernel): hadK before the
            # butterflies, fp32 throughout, ONE rounding — the reference's fp16 intermediate t is never formed.  The result
            # is the correctly rounded exact transform (to an ulp), and it stays inside the reference pipeline's own
            # rounding noise around the reference-rounded value: K terms, each off by at most half an ulp of t.
            exact = oi.f16(oi.had_blocks(xs, hd, hT.numpy().astype(np.float64), round_mid=False)) / 64
            _assert_ulp(got, exact)
            assert np.all(np.abs(got - want) <= (1.5 * _ulp16(want) + 0.5 * K * extra) * 1.0001)
        else:
            assert np.all(np.abs(got - want) <= (_ulp16(want) + extra) * 1.0001)
            assert (np.abs(got - want) > _ulp16(want) * 1.0001).mean() < 5e-4 + 2e-5 * K  # and it is rare (one chance per term)
    else:
        _assert_ulp(y.cpu().numpy(), want)


@gpu
def test_rotate_modes_vs_oracle(qp):
    had = qp.hadamard
    rng = np.random.default_rng(5)
    n = 3584
    hadK, K = had.get_hadK(n)
    h16 = hadK.half().cuda()
    hn = hadK.numpy().astype(np.float64)
    # fp32 input is rounded to fp16 first; sv is an fp16 multiply after the transform (IncoherentLinear's right side)
    x32 = (rng.standard_normal((3, n)) * 3).astype(np.float32)
    sv = (_signs(rng, n).astype(np.float32) * 32).astype(np.float16)
    y = had.rotate(torch.from_numpy(x32).cuda(), hadK=h16, K=K, sv=torch.from_numpy(sv).cuda(), in_mode=had.IN_F32,
                   round_mid=False)
    want = oi.f16(oi.f16(oi.had_blocks(oi.f16(x32), n, hn)) * sv.astype(np.float64))
    _assert_ulp(y.cpu().numpy(), want)
    # SwiGLU of up | gate (fp32, as the fused GEMV epilogue leaves them)
    ug = (rng.standard_normal((2, 2 * n)) * 2).astype(np.float32)
    su = _signs(rng, n)
    y = had.rotate(torch.from_numpy(ug).cuda(), hadK=h16, K=K, su=torch.from_numpy(su).cuda(), post_scale=1 / 64,
                   in_mode=had.IN_SWIGLU_F32)
    act = oi.swiglu(oi.f16(ug[:, :n]), oi.f16(ug[:, n:]))
    want = oi.f16(oi.had_blocks(oi.f16(act * su), n, hn, round_mid=True)) / 64
    got = y.cpu().numpy().astype(np.float64)
    # the device exp differs from numpy's in the last fp32 bits: an activation may round to the neighbouring fp16
    # value, which the transform spreads over the block -> allow 2 ulp of the block's rms instead of 1 ulp pointwise
    assert np.abs(got - want).max() <= 2 * _ulp16(np.sqrt((want ** 2).mean()))
    # reference API names
    x = rng.standard_normal((2, n)).astype(np.float16)
    y = had.matmul_hadU_cuda(torch.from_numpy(x).cuda(), hadK.cuda(), K)
    _assert_ulp(y.cpu().numpy(), oi.matmul_hadU_cuda(x, hn))
    yt = had.matmul_hadUt_cuda(torch.from_numpy(x[:, :3072]).contiguous().cuda(), *[t for t in had.get_hadK(3072)])
    _assert_ulp(yt.cpu().numpy(), oi.matmul_hadU_cuda(x[:, :3072], had.get_hadK(3072)[0].numpy().T.astype(np.float64)))
    yh = had.matmul_hadU_head_cuda(torch.from_numpy(x[:, :2048]).contiguous().cuda(), None, 1, 128)
    _assert_ulp(yh.cpu().numpy(), oi.matmul_hadU_head_cuda(x[:, :2048], 128))
    had.ensure_hadamard_op()
    yo = torch.ops.hadamard.hadamard(torch.from_numpy(x[:, :2048]).contiguous().cuda(), 2048 ** -0.5)
    _assert_ulp(yo.cpu().numpy(), oi.f16(oi.had_blocks(x[:, :2048], 2048)))
    with pytest.raises(RuntimeError):
        had.rotate(torch.from_numpy(x), K=1)  # CPU tensor: no CPU implementation


def _dequant(oracle, lin_info, qstr):
    li = lin_info
    m, k = li["out_features"], li["in_features"]
    if "trellis" in li:
        return oracle.tcq_dequant(li["trellis"].numpy(), li["tlut"].numpy(), m, k, li["tlut_bits"], li["KV"])
    if "in_part" in li:
        return oracle.tcq_dequant(li["trellis1"].numpy(), li["tlut"].numpy(), m, k, li["tlut_bits"], li["KV"][0],
                                  c2=li["trellis2"].numpy(), KV2=li["KV"][1], split=2)
    if "out_part" in li:
        return oracle.tcq_dequant(li["trellis1"].numpy(), li["tlut"].numpy(), m, k, li["tlut_bits"], li["KV"][0],
                                  c2=li["trellis2"].numpy(), KV2=li["KV"][1], split=1)
    return oracle.lut_tc_dequant(li["qweight"].numpy(), li["lut"].numpy(), m, k, li["lut_bits"], li["vec_sz"])


def _lin(oracle, info, x16):
    """(reference-rounded y, magnitude sum |w x| * Wscale * scale) of `linear(x) * Wscale * scale`."""
    W = _dequant(oracle, info["linear_info"], info["quant_info"]["quantizer_str"]).astype(np.float64)
    acc = x16 @ W.T
    mag = np.abs(x16) @ np.abs(W.T)
    return acc, mag


def _close(got, want, mag, what):
    err = np.abs(got.astype(np.float64) - want)
    tol = 2.0 ** -9 * mag + 2.0 ** -20
    assert np.all(err <= tol), f"{what}: max err/tol {(err / tol).max():.2f}"


@gpu
@pytest.mark.parametrize("qstr,merge", [("tcq_4_0_1", False), ("tcq_4_0_1", True), ("tcomb_5_6_0.5_0_1", True),
                                        ("ldlq_2_7_0_1", False)])
@pytest.mark.parametrize("n,hidden", [(1, 1024), (3, 1024), (1, 2048), (3, 2048)])
def test_incoherent_mlp_vs_oracle(qp, oracle, qstr, merge, n, hidden):
    # hidden = 2048, n = 1: the up|gate rotation runs inside the GEMV launch (x_rot); else: separate qpal_hadamard launch
    cfg = _cfg(hidden=hidden)
    H, I = cfg.hidden_size, cfg.intermediate_size
    up, gate, down = _info(qp, H, I, qstr, 11), _info(qp, H, I, qstr, 12), _info(qp, I, H, qstr, 13)
    mlp = qp.IncoherentMLP.gen_layer_from_info(cfg, up, gate, down, merge_ug=merge).cuda()
    rng = np.random.default_rng(n)
    x = rng.standard_normal((n, H)).astype(np.float16)
    y = mlp(torch.from_numpy(x).cuda())
    assert y.dtype == torch.float16 and tuple(y.shape) == (n, H)
    scale = mlp.scale
    hT = qp.hadamard.get_hadK(I)[0].numpy().T.astype(np.float64)
    xr = oi.left_input(x, up["SU"].numpy(), None, scale)
    au, mu = _lin(oracle, up, xr)
    ag, mg = _lin(oracle, gate, xr)
    u = oi.linear_post(au, up["Wscale"].numpy(), scale)
    g = oi.linear_post(ag, gate["Wscale"].numpy(), scale)
    # stage check through the module's two-step interface (reference semantics of compute_ug)
    act = mlp.compute_ug(torch.from_numpy(x).cuda()).cpu().numpy()
    want_act = oi.swiglu(u, g)
    wu, wg = up["Wscale"].numpy().astype(np.float64) * scale, gate["Wscale"].numpy().astype(np.float64) * scale
    _close(act, want_act, np.abs(oi.silu(g)) * mu * wu + np.abs(u) * mg * wg + np.abs(want_act), "compute_ug")
    xd = oi.left_input(want_act, down["SU"].numpy(), hT, scale)
    ad, md = _lin(oracle, down, xd)
    want = oi.linear_post(ad, down["Wscale"].numpy(), scale)
    # the activation error (above) passes through an orthogonal transform and down_proj: bound it by the same
    # magnitude sum, one more factor 2
    _close(y.cpu().numpy(), want, 2 * md * down["Wscale"].numpy().astype(np.float64) * scale + np.abs(want), "forward")
    # compute_dp on the oracle's activation isolates the second half exactly
    yd = mlp.compute_dp(torch.from_numpy(want_act.astype(np.float16)).cuda()).cpu().numpy()
    _close(yd, want, md * down["Wscale"].numpy().astype(np.float64) * scale, "compute_dp")


@gpu
@pytest.mark.parametrize("flags", [{}, {"merge_qkv": True}, {"merge_qk": True}, {"merge_kv": True}, {"merge_qv": True}])
def test_incoherent_attention_projections_vs_oracle(qp, oracle, flags):
    cfg = _cfg(hidden=1024, heads=8, kv_heads=2)
    H = cfg.hidden_size
    kvo = 256
    q, k, v, o = (_info(qp, H, H, "tcq_5_0_1", 21), _info(qp, H, kvo, "tcq_5_0_1", 22), _info(qp, H, kvo, "tcq_5_0_1", 23),
                  _info(qp, H, H, "ldlq_1_4_0_1", 24))
    attn = qp.IncoherentSdpaAttention.gen_layer_from_info(cfg, 0, q, k, v, o, **flags).cuda()
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 1, H)).astype(np.float16)
    got = attn.compute_qkv(torch.from_numpy(x).cuda())
    xr = oi.left_input(x.reshape(2, H), q["SU"].numpy(), None, attn.scale)
    for name, info, y in zip("qkv", (q, k, v), got):
        acc, mag = _lin(oracle, info, xr)
        want = oi.linear_post(acc, info["Wscale"].numpy(), attn.scale)
        assert tuple(y.shape) == (2, 1, info["out_features"]) and y.dtype == torch.float16
        _close(y.reshape(2, -1).cpu().numpy(), want, mag * info["Wscale"].numpy().astype(np.float64) * attn.scale
               + np.abs(want), f"compute_qkv[{name}] {flags}")
    a = rng.standard_normal((2, 1, H)).astype(np.float16)
    yo = attn.compute_o(torch.from_numpy(a).cuda())
    xo = oi.left_input(a.reshape(2, H), o["SU"].numpy(), None, attn.scale)
    acc, mag = _lin(oracle, o, xo)
    want = oi.linear_post(acc, o["Wscale"].numpy(), attn.scale)
    _close(yo.reshape(2, -1).cpu().numpy(), want, mag * o["Wscale"].numpy().astype(np.float64) * attn.scale + np.abs(want),
           "compute_o")
    # whole forward runs (SDPA in between is torch's; covered for shape/dtype and determinism only)
    hs = torch.from_numpy(rng.standard_normal((1, 4, H)).astype(np.float16)).cuda()
    y1, _, _ = attn(hs)
    y2, _, _ = attn(hs)
    assert tuple(y1.shape) == (1, 4, H) and torch.equal(y1, y2) and torch.isfinite(y1).all()


@gpu
@pytest.mark.parametrize("rot_info", ["all", "skip_l", "skip_r", "skip_lr"])
def test_incoherent_linear_vs_oracle(qp, oracle, rot_info):
    k, m, hadU, hadV = 3584, 1024, 448, 128
    info = dict(_info(qp, k, m, "tcq_6_0_1", 31), hadU=hadU, hadV=hadV, rot_info=rot_info)
    info["bias"] = None
    layer = qp.IncoherentLinear.gen_layer_from_info(info, merge_layers=True).cuda()
    rng = np.random.default_rng(9)
    x = rng.standard_normal((3, k)).astype(np.float16)
    y = layer(torch.from_numpy(x).cuda())
    scale = layer.scale
    su, sv, ws = (info[n].numpy().astype(np.float64) for n in ("SU", "SV", "Wscale"))
    hl = qp.hadamard.get_hadK(hadU)[0].numpy().T.astype(np.float64)
    if rot_info in ("all", "skip_r"):
        xr = oi.f16(oi.matmul_hadU_head_cuda(oi.f16(x * su), hadU, hl) / scale)
    else:
        xr = oi.f16(x.astype(np.float64) / scale)
    acc, mag = _lin(oracle, info, xr)
    z = oi.f16(oi.f16(acc) * ws)
    if rot_info in ("all", "skip_l"):
        want = oi.f16(oi.matmul_hadU_head_cuda(z, hadV) * oi.f16(sv * scale))
        # a rotation of the output mixes hadV errors: bound by the block rms of the magnitude
        bound = np.sqrt(((mag * ws) ** 2).reshape(3, m // hadV, hadV).mean(-1, keepdims=True)).repeat(hadV, -1)
        tolmag = bound.reshape(3, m) * scale * 4 + np.abs(want)
    else:
        want = oi.f16(z * scale)
        tolmag = mag * ws * scale + np.abs(want)
    assert y.dtype == torch.float16 and tuple(y.shape) == (3, m)
    _close(y.cpu().numpy(), want, tolmag, f"IncoherentLinear {rot_info}")


@gpu
def test_gemv_epilogue_scale_and_strided_out(qp, oracle):
    """wscale / oscale / ldo of the multi-job entry points: out[:, block] = acc * wscale * oscale."""
    k, n = 1024, 5
    infos = [_info(qp, k, m, "tcq_3_0_1", 40 + i) for i, m in enumerate((1024, 256, 256))]
    layers = [qp.QTIPLinearTCQ.gen_layer_from_info(i["linear_info"]).cuda() for i in infos]
    qp.share_codebooks(layers)
    rng = np.random.default_rng(1)
    x = rng.standard_normal((n, k)).astype(np.float16)
    buf = torch.full((n, 1536 + 64), -7.0, dtype=torch.float32, device="cuda")
    outs = list(buf[:, :1536].split([1024, 256, 256], dim=1))
    ws = [i["Wscale"].cuda() for i in infos]
    qp.multi_gemv(layers, torch.from_numpy(x).cuda(), outs=outs, wscales=ws, oscale=64.0)
    assert torch.all(buf[:, 1536:] == -7.0)  # the padding columns of the wider buffer are untouched
    col = 0
    for info, w in zip(infos, ws):
        acc, mag = _lin(oracle, info, x.astype(np.float64))
        wsn = w.cpu().numpy().astype(np.float64) * 64.0
        got = buf[:, col:col + info["out_features"]].cpu().numpy().astype(np.float64)
        assert np.all(np.abs(got - acc * wsn) <= 1e-5 * mag * wsn + 1e-30)
        col += info["out_features"]
    # LUT family, no wscale but an oscale, single job
    info = _info(qp, k, 512, "ldlq_2_9_0_1", 50)
    layer = qp.VQLinearPackTensorCore.gen_layer_from_info(info["linear_info"]).cuda()
    (y,) = qp.multi_gemv([layer], torch.from_numpy(x).cuda(), oscale=0.5)
    acc, mag = _lin(oracle, info, x.astype(np.float64))
    assert np.all(np.abs(y.cpu().numpy() - acc * 0.5) <= 1e-5 * mag + 1e-30)


@gpu
@pytest.mark.parametrize("k", [2048, 4096])
def test_fused_rotation_equals_separate_launch(qp, oracle, k, n=1):
    """x_rot of the GEMV entry points (rotation inside the kernel's x staging) vs qpal_hadamard + plain GEMV: the
    same transform code on the same data -> identical staged x -> bit-identical outputs; plus the oracle bound."""
    rng = np.random.default_rng(k + n)
    x = rng.standard_normal((n, k)).astype(np.float16)
    su = _signs(rng, k)
    xd, sud = torch.from_numpy(x).cuda(), torch.from_numpy(su).cuda()
    xr = qp.hadamard.rotate(xd, su=sud, post_scale=1 / 64)
    want_x = oi.left_input(x, su, None, 64.0)
    _assert_ulp(xr.cpu().numpy(), want_x)
    for qstr, ms in (("tcq_4_0_1", (512, 256)), ("tcomb_6_7_0.5_0_1", (512,)), ("ldlq_2_8_0_1", (256, 256, 128))):
        infos = [_info(qp, k, m, qstr, 60 + i) for i, m in enumerate(ms)]
        layers = [qp.make_linear(i).cuda() for i in infos]
        qp.share_codebooks(layers)
        assert qp.linear.rotation_fusable(layers, n)
        ws = [i["Wscale"].cuda() for i in infos]
        sep = qp.multi_gemv(layers, xr, wscales=ws, oscale=64.0)
        fused = qp.multi_gemv(layers, xd, wscales=ws, oscale=64.0, x_rot=(sud, 1 / 64))
        for a, b, info in zip(sep, fused, infos):
            assert torch.equal(a, b), qstr
            acc, mag = _lin(oracle, info, xr.cpu().numpy().astype(np.float64))
            wsn = info["Wscale"].numpy().astype(np.float64) * 64.0
            assert np.all(np.abs(b.cpu().numpy() - acc * wsn) <= 1e-5 * mag * wsn + 1e-30)
    # not fusable: SIMT packing, k without the fast transform, batch beyond the LDS staging
    assert not qp.ops.can_fuse_rotation(1, 14336) and not qp.ops.can_fuse_rotation(2, 4096)
    assert not qp.ops.can_fuse_rotation(1, 8192) and not qp.ops.can_fuse_rotation(1, 1024)
    with pytest.raises(RuntimeError):
        qp.multi_gemv([qp.make_linear(_info(qp, 1024, 256, "tcq_4_0_1", 70)).cuda()],
                      torch.zeros(1, 1024, dtype=torch.float16, device="cuda"), x_rot=(None, 1.0))


@gpu
def test_fused_14336_rotation_in_the_down_proj_staging(qp, oracle):
    """x_rot = (su, post, hadK, 28) on a k = 14336 layer: the GEMV's x staging applies (hadK(28) (x) H_512) / sqrt(k) with the
    reference's fp16 rounding between the factors (csrc/rot_k28.h; lib/utils/matmul_had.py:137-148) instead of a qpal_hadamard
    launch in front of every down_proj.
    (1) The staged vector itself, read back through a SELECTION layer (5-bit scalar codes, lut = {0, 1, ...}: row r of W is the
        unit vector e_sel[r], so y[r] = x'[sel[r]] exactly): within 1 fp16 ulp of the oracle's restatement of the reference
        pipeline, almost everywhere identical.
    (2) Real layers: equal to the plain GEMV on the oracle-rotated input (the same decode and summation order) up to those rare
        ulps, and within the module bound of the oracle's float64 result."""
    k, K = 14336, 28
    rng = np.random.default_rng(28)
    act = (rng.standard_normal((1, k)) * 0.7).astype(np.float16)
    su = _signs(rng, k)
    hk, KK = qp.hadamard.get_hadK(k)
    assert KK == K
    hadk_T = hk.T.contiguous().half()          # what IncoherentMLP passes (had_left_dp_T)
    scale = 48.0
    want = oi.left_input(act, su, hadk_T.numpy().astype(np.float64), scale)   # y[j] = sum_i M[j][i] t[i], M as passed
    actd, sud, hkd = torch.from_numpy(act).cuda(), torch.from_numpy(su).cuda(), hadk_T.cuda()
    assert qp.ops.can_fuse_rotation(1, k, K) and not qp.ops.can_fuse_rotation(2, k, K) and not qp.ops.can_fuse_rotation(1, 4096, K)
    # (1) selection layer: blocks j = 0, 13, 27 whole + 512 random positions
    sel = np.concatenate([np.arange(0, 512), np.arange(13 * 512, 14 * 512), np.arange(27 * 512, 28 * 512),
                          rng.choice(k, 512, replace=False)]).astype(np.int64)
    m = sel.size
    idx = torch.zeros((m, k), dtype=torch.int32)
    idx[torch.arange(m), torch.from_numpy(sel)] = 1
    lut = torch.zeros(32, dtype=torch.float16)
    lut[1] = 1.0
    lut[2:] = torch.linspace(-3, 3, 30).half()
    info = {"in_features": k, "out_features": m, "lut_bits": 5, "vec_sz": 1, "bias": None, "dtype": torch.float16,
            "qweight": qp.packers.pack_qweight(idx, 1, 5), "lut": lut.view(32, 1)}
    layer = qp.VQLinearPackTensorCore.gen_layer_from_info(info).cuda()
    (y,) = qp.multi_gemv([layer], actd, x_rot=(sud, 1 / scale, hkd, K))
    got = y.cpu().numpy()[0]
    _assert_ulp(got, want[0, sel])
    assert np.mean(got == want[0, sel]) > 0.999   # measured: every element identical (fp32 vs float64 sums differ at exact ties only)
    # the sign vector may also have been applied by the producer (act_su of the gate|up epilogue): same staged vector
    (y2,) = qp.multi_gemv([layer], (actd * sud), x_rot=(None, 1 / scale, hkd, K))
    assert torch.equal(y, y2)
    # (the separate launch, qpal_hadamard, keeps fp32 between the two factors: it differs from the reference pipeline's rounding
    # of the intermediate in ~40 % of the elements, by design — test_rotate_vs_oracle holds it to its own bound)
    # (2) real down_proj layers
    x_or = torch.from_numpy(want.astype(np.float16)).cuda()
    for qstr in ("tcomb_6_7_0.5_0_1", "tcq_6_0_1", "ldlq_2_12_0_1"):
        linfo = _info(qp, k, 1024, qstr, 90)
        lay = qp.make_linear(linfo).cuda()
        ws = linfo["Wscale"].cuda()
        (fused,) = qp.multi_gemv([lay], actd, wscales=[ws], oscale=scale, x_rot=(sud, 1 / scale, hkd, K))
        (plain,) = qp.multi_gemv([lay], x_or, wscales=[ws], oscale=scale)
        acc, mag = _lin(oracle, linfo, want)
        wsn = linfo["Wscale"].numpy().astype(np.float64) * scale
        _close(fused.cpu().numpy(), acc * wsn, mag * wsn, qstr)
        assert torch.equal(fused, plain), qstr   # the same staged vector, the same decode and summation order
        # accumulate (the residual add of the decoder block) through the same launch
        h = torch.randn(1, 1024, device="cuda")
        acc_buf = h.clone()
        qp.multi_gemv([lay], actd, outs=[acc_buf], wscales=[ws], oscale=scale, x_rot=(sud, 1 / scale, hkd, K), accumulate=True)
        assert torch.allclose(acc_buf, h + fused, atol=1e-4 * float(fused.abs().max()) + 1e-6)
    # codecs whose image is smaller than the rotation's 40 KiB of scratch are refused (callers rotate with qpal_hadamard)
    small = qp.make_linear(_info(qp, k, 256, "ldlq_2_8_0_1", 91)).cuda()
    with pytest.raises(RuntimeError):
        qp.multi_gemv([small], actd, x_rot=(sud, 1 / scale, hkd, K))


@gpu
@pytest.mark.parametrize("n", [1, 20])
def test_incoherent_mlp_70b_shapes_and_large_batch(qp, oracle, n):
    """Llama-70B sizes (hidden 8192: rotation in a launch of its own; intermediate 28672 = 28 * 1024: two-launch K > 1 path)
    and a batch beyond the fused GEMV (n = 20: decode-to-fp16 + GEMM path inside the wrapper)."""
    cfg = _cfg(hidden=8192, inter=28672)
    H, I = cfg.hidden_size, cfg.intermediate_size
    qstr = "tcq_3_0_1"
    up, gate, down = _info(qp, H, I, qstr, 81), _info(qp, H, I, qstr, 82), _info(qp, I, H, qstr, 83)
    mlp = qp.IncoherentMLP.gen_layer_from_info(cfg, up, gate, down, merge_ug=True).cuda()
    rng = np.random.default_rng(n)
    x = rng.standard_normal((n, H)).astype(np.float16)
    y = mlp(torch.from_numpy(x).cuda())
    scale = mlp.scale
    hT = qp.hadamard.get_hadK(I)[0].numpy().T.astype(np.float64)
    xr = oi.left_input(x, up["SU"].numpy(), None, scale)
    au, mu = _lin(oracle, up, xr)
    ag, mg = _lin(oracle, gate, xr)
    act = oi.swiglu(oi.linear_post(au, up["Wscale"].numpy(), scale), oi.linear_post(ag, gate["Wscale"].numpy(), scale))
    xd = oi.left_input(act, down["SU"].numpy(), hT, scale)
    ad, md = _lin(oracle, down, xd)
    want = oi.linear_post(ad, down["Wscale"].numpy(), scale)
    assert y.dtype == torch.float16 and tuple(y.shape) == (n, H)
    _close(y.cpu().numpy(), want, 2 * md * down["Wscale"].numpy().astype(np.float64) * scale + np.abs(want), "70B-shaped MLP")


@gpu
def test_incoherent_modules_under_hip_graph(qp):
    """A decoder block's projections captured once and replayed (how a decode loop runs them): same bits as eager,
    including the pre-zero chain between the launches and the two-launch K > 1 rotation."""
    cfg = _cfg(hidden=2048, inter=3584)
    H, I = cfg.hidden_size, cfg.intermediate_size
    qstr = "tcomb_6_7_0.5_0_1"
    mlp = qp.IncoherentMLP.gen_layer_from_info(cfg, _info(qp, H, I, qstr, 91), _info(qp, H, I, qstr, 92),
                                               _info(qp, I, H, qstr, 93)).cuda()
    attn = qp.IncoherentSdpaAttention.gen_layer_from_info(cfg, 0, _info(qp, H, H, qstr, 94), _info(qp, H, 512, qstr, 95),
                                                          _info(qp, H, 512, qstr, 96), _info(qp, H, H, qstr, 97)).cuda()
    x = torch.randn(1, 1, H, device="cuda").half()

    def block(inp):
        a, _, _ = attn(inp)
        return mlp(inp + a)

    eager = block(x).clone()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        block(x)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            y = block(x)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        assert torch.equal(y, eager)
        x.copy_(torch.randn(1, 1, H, device="cuda").half())   # new input, same graph
        g.replay()
        torch.cuda.synchronize()
        replayed = y.clone()
    assert torch.equal(replayed, block(x))
