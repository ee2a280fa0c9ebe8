"""GPU tests of the decoder-block glue (csrc/decoder_glue.hip) and of the fused decode step of perf/decode_llama.py:
rotary embedding + KV-cache write and single-token attention against plain torch, RMSNorm / fp32-residual fusion of the GEMV
launches against the modular Incoherent* path.  Reference counterparts: model/llama.py apply_rotary_pos_emb + StaticCache
update + SDPA, lib/linear/incoherent_linear.py:76-108, 317-338."""
import math
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def qp():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import qpalette_amd
    qpalette_amd._native.lib()
    return qpalette_amd


def _rotate_half(x):
    x1, x2 = x[..., : x.shape[-1] // 2], x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


@pytest.mark.parametrize("nq,nkv,hd,ctx,pos", [(32, 8, 128, 256, 17), (8, 8, 64, 64, 0), (64, 8, 128, 2048, 2047)])
def test_rope_kv_and_attn_decode_match_torch(qp, nq, nkv, hd, ctx, pos):
    dev = torch.device("cuda", 0)
    nat = qp._native
    gen = torch.Generator(device=dev).manual_seed(nq + pos)
    q32 = torch.randn(nq * hd, device=dev, generator=gen)
    k32 = torch.randn(nkv * hd, device=dev, generator=gen)
    v32 = torch.randn(nkv * hd, device=dev, generator=gen)
    kc = (torch.randn(nkv, ctx, hd, device=dev, generator=gen) * 0.5).half()
    vc = (torch.randn(nkv, ctx, hd, device=dev, generator=gen) * 0.5).half()
    kc_ref, vc_ref = kc.clone(), vc.clone()
    inv_freq = 1.0 / (500000.0 ** (torch.arange(0, hd, 2, device=dev).float() / hd))
    pos_t = torch.tensor([pos], dtype=torch.long, device=dev)
    q16 = torch.empty(nq * hd, dtype=torch.float16, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    nat.check(nat.lib().qpal_rope_kv(q32.data_ptr(), k32.data_ptr(), v32.data_ptr(), q16.data_ptr(), kc.data_ptr(), vc.data_ptr(),
                                     pos_t.data_ptr(), inv_freq.data_ptr(), nq, nkv, hd, ctx, stream), "qpal_rope_kv")
    # torch reference: the reference's fp16 pipeline (model/llama.py apply_rotary_pos_emb on .half() inputs)
    ang = pos_t.float()[:, None] * inv_freq[None, :]
    emb = torch.cat((ang, ang), dim=-1)
    cos, sin = emb.cos().half(), emb.sin().half()
    qh, kh = q32.half().view(nq, hd), k32.half().view(nkv, hd)
    q_ref = qh * cos + _rotate_half(qh) * sin
    k_ref = kh * cos + _rotate_half(kh) * sin
    kc_ref[:, pos] = k_ref
    vc_ref[:, pos] = v32.half().view(nkv, hd)
    torch.cuda.synchronize()
    assert torch.allclose(q16.view(nq, hd).float(), q_ref.float(), atol=4e-3, rtol=2e-3)  # sincos in fp32 vs torch's: <= 1 fp16 ulp
    assert torch.allclose(kc.float(), kc_ref.float(), atol=4e-3, rtol=2e-3) and torch.equal(vc, vc_ref)
    out = torch.empty(nq * hd, dtype=torch.float16, device=dev)
    nat.check(nat.lib().qpal_attn_decode(q16.data_ptr(), kc.data_ptr(), vc.data_ptr(), out.data_ptr(), pos_t.data_ptr(), nq, nkv, hd,
                                         ctx, 1.0 / math.sqrt(hd), stream), "qpal_attn_decode")
    qf = q16.view(1, nq, 1, hd).float()
    kf = kc[:, : pos + 1].float().repeat_interleave(nq // nkv, dim=0)[None]
    vf = vc[:, : pos + 1].float().repeat_interleave(nq // nkv, dim=0)[None]
    ref = torch.softmax(qf @ kf.transpose(-1, -2) / math.sqrt(hd), dim=-1) @ vf
    torch.cuda.synchronize()
    assert torch.allclose(out.view(nq, hd).float(), ref.view(nq, hd), atol=2e-3, rtol=2e-3)


@pytest.mark.parametrize("nq,nkv,hd,ctx,pos,split", [
    (32, 8, 128, 256, 17, False), (8, 8, 64, 64, 0, False), (64, 8, 128, 4096, 4095, False), (16, 4, 256, 512, 100, False),
    (32, 8, 128, 1024, 1, False), (32, 8, 128, 2048, 1300, False),
    # split-context form (workspace given): first / middle / last chunk owning the new position, empty chunks, chunk boundaries
    (32, 8, 128, 2048, 0, True), (32, 8, 128, 2048, 63, True), (32, 8, 128, 2048, 64, True), (32, 8, 128, 2048, 1300, True),
    (32, 8, 128, 4096, 4095, True), (64, 8, 128, 8192, 5000, True), (8, 8, 64, 4096, 2049, True), (16, 4, 256, 2048, 2047, True),
    (32, 8, 128, 32768, 31000, True), (32, 8, 128, 512, 100, True),
    # below 512 positions the split launch runs the one-head body; above, the existing context is cut evenly over the splits
    (32, 8, 128, 2048, 511, True), (32, 8, 128, 2048, 512, True), (32, 8, 128, 2048, 768, True), (8, 1, 128, 8192, 5000, True), (32, 8, 128, 32768, 800, True), (64, 8, 128, 32768, 1023, True),
    (32, 8, 128, 32768, 2047, True), (32, 8, 128, 32768, 2048, True)])
def test_attn_rope_decode_one_launch_matches_torch(qp, nq, nkv, hd, ctx, pos, split):
    """qpal_attn_rope_decode = rope + cache append + attention in one launch: equals the two-launch pair's math (torch
    reference as above), leaves the cache exactly as qpal_rope_kv would, and does not depend on the cache row it writes."""
    dev = torch.device("cuda", 0)
    nat = qp._native
    gen = torch.Generator(device=dev).manual_seed(nq * 7 + pos)
    q32 = torch.randn(nq * hd, device=dev, generator=gen)
    k32 = torch.randn(nkv * hd, device=dev, generator=gen)
    v32 = torch.randn(nkv * hd, device=dev, generator=gen)
    kc = (torch.randn(nkv, ctx, hd, device=dev, generator=gen) * 0.5).half()
    vc = (torch.randn(nkv, ctx, hd, device=dev, generator=gen) * 0.5).half()
    kc[:, pos] = float("nan")  # whatever is in the cache at the new position must not matter
    vc[:, pos] = float("nan")
    kc_ref, vc_ref = kc.clone(), vc.clone()
    inv_freq = 1.0 / (500000.0 ** (torch.arange(0, hd, 2, device=dev).float() / hd))
    pos_t = torch.tensor([pos], dtype=torch.long, device=dev)
    out = torch.empty(nq * hd, dtype=torch.float16, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    ws_bytes = nat.lib().qpal_attn_ws_bytes(nq, nkv, hd, ctx) if split else 0
    assert (ws_bytes > 0) == (split and ctx >= 2048)
    ws = torch.zeros(max(ws_bytes, 4) // 4, dtype=torch.float32, device=dev)
    for rep in range(2):  # twice: the split form's tickets must be back at zero after a launch
        if rep:
            kc[:, pos] = float("nan")
            vc[:, pos] = float("nan")
            out.fill_(float("nan"))
        nat.check(nat.lib().qpal_attn_rope_decode(q32.data_ptr(), k32.data_ptr(), v32.data_ptr(), kc.data_ptr(), vc.data_ptr(),
                                                  out.data_ptr(), pos_t.data_ptr(), inv_freq.data_ptr(), nq, nkv, hd, ctx,
                                                  1.0 / math.sqrt(hd), ws.data_ptr() if ws_bytes else None, ws_bytes, stream),
                  "qpal_attn_rope_decode")
    ang = pos_t.float()[:, None] * inv_freq[None, :]
    emb = torch.cat((ang, ang), dim=-1)
    cos, sin = emb.cos().half(), emb.sin().half()
    qh, kh = q32.half().view(nq, hd), k32.half().view(nkv, hd)
    q_ref = qh * cos + _rotate_half(qh) * sin
    kc_ref[:, pos] = kh * cos + _rotate_half(kh) * sin
    vc_ref[:, pos] = v32.half().view(nkv, hd)
    torch.cuda.synchronize()
    assert torch.allclose(kc.float(), kc_ref.float(), atol=4e-3, rtol=2e-3) and torch.equal(vc, vc_ref)
    qf = q_ref.view(1, nq, 1, hd).float()
    kf = kc_ref[:, : pos + 1].float().repeat_interleave(nq // nkv, dim=0)[None]
    vf = vc_ref[:, : pos + 1].float().repeat_interleave(nq // nkv, dim=0)[None]
    ref = torch.softmax(qf @ kf.transpose(-1, -2) / math.sqrt(hd), dim=-1) @ vf
    # q / k rotations differ from torch's by <= 1 fp16 ulp of cos / sin (fp32 sincos here): a few 1e-3 on a score of O(1)
    assert torch.allclose(out.view(nq, hd).float(), ref.view(nq, hd), atol=4e-3, rtol=4e-3)


@pytest.mark.parametrize("ctx,split", [(256, False), (4096, True)])
def test_attention_position_outside_the_cache_touches_nothing(qp, ctx, split):
    """*pos lives on the device, so the host cannot validate it: a position outside [0, max_len) must not write anywhere."""
    dev = torch.device("cuda", 0)
    nat = qp._native
    nq, nkv, hd = 32, 8, 128
    q32, k32, v32 = (torch.randn(n * hd, device=dev) for n in (nq, nkv, nkv))
    kc = torch.randn(nkv, ctx, hd, device=dev).half()
    vc = torch.randn(nkv, ctx, hd, device=dev).half()
    guard = torch.zeros(2, 4096, dtype=torch.float16, device=dev)  # allocated right after the caches
    kc0, vc0 = kc.clone(), vc.clone()
    inv_freq = 1.0 / (500000.0 ** (torch.arange(0, hd, 2, device=dev).float() / hd))
    out = torch.full((nq * hd,), 3.0, dtype=torch.float16, device=dev)
    wsb = nat.lib().qpal_attn_ws_bytes(nq, nkv, hd, ctx) if split else 0
    ws = torch.zeros(max(wsb, 4) // 4, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    for bad in (ctx, ctx + 5, -1, 1 << 40):
        pos_t = torch.tensor([bad], dtype=torch.long, device=dev)
        nat.check(nat.lib().qpal_attn_rope_decode(q32.data_ptr(), k32.data_ptr(), v32.data_ptr(), kc.data_ptr(), vc.data_ptr(), out.data_ptr(),
                                                  pos_t.data_ptr(), inv_freq.data_ptr(), nq, nkv, hd, ctx, 0.1, ws.data_ptr() if wsb else None,
                                                  wsb, stream), "qpal_attn_rope_decode")
        q16 = torch.zeros(nq * hd, dtype=torch.float16, device=dev)
        nat.check(nat.lib().qpal_rope_kv(q32.data_ptr(), k32.data_ptr(), v32.data_ptr(), q16.data_ptr(), kc.data_ptr(), vc.data_ptr(),
                                         pos_t.data_ptr(), inv_freq.data_ptr(), nq, nkv, hd, ctx, stream), "qpal_rope_kv")
        nat.check(nat.lib().qpal_attn_decode(q16.data_ptr(), kc.data_ptr(), vc.data_ptr(), out.data_ptr(), pos_t.data_ptr(), nq, nkv, hd,
                                             ctx, 0.1, stream), "qpal_attn_decode")
    torch.cuda.synchronize()
    assert torch.equal(kc, kc0) and torch.equal(vc, vc0) and float(guard.abs().max()) == 0.0
    assert bool((out == 3.0).all()) and float(ws.abs().max()) == 0.0


@pytest.mark.parametrize("qstr", ["tcomb_6_7_0.5_none_0.9", "ldlq_2_8_none_1.0"])
def test_rmsnorm_fp32_stream_and_accumulate_in_the_gemv_launch(qp, qstr):
    """multi_gemv(..., x = fp32 residual stream, x_rot, x_rms) == rotate(rmsnorm(x).half()) then GEMV; accumulate adds into out."""
    dev = torch.device("cuda", 0)
    k, m = 4096, 1024
    layer = qp.make_linear_from_info(qstr, qp.mem_op.dummy_linear_info(k, m, qstr, seed=2)).to(dev)
    gen = torch.Generator(device=dev).manual_seed(3)
    h32 = torch.randn(1, k, device=dev, generator=gen) * 3.0
    w_ln = (1.0 + 0.1 * torch.randn(k, device=dev, generator=gen)).half()
    su = (torch.randint(0, 2, (k,), device=dev, generator=gen) * 2 - 1).half()
    wsc = (0.01 + 0.01 * torch.rand(m, device=dev, generator=gen)).half()
    eps, scale = 1e-5, 64.0
    x_norm = torch.nn.functional.rms_norm(h32, (k,), w_ln.float(), eps).half()   # fp32 norm, one fp16 rounding (the reference)
    xr = qp.hadamard.rotate(x_norm, su=su, post_scale=1.0 / scale)
    (ref,) = qp.multi_gemv([layer], xr, wscales=[wsc], oscale=scale)
    (got,) = qp.multi_gemv([layer], h32, wscales=[wsc], oscale=scale, x_rot=(su, 1.0 / scale), x_rms=(eps, w_ln))
    torch.cuda.synchronize()
    tol = 2.0 ** -9 * float(ref.abs().max())  # rsqrt / summation-order differences move a few inputs by one fp16 ulp
    assert torch.allclose(got, ref, atol=tol, rtol=2e-3), float((got - ref).abs().max())
    acc = torch.full((1, m), 0.5, device=dev)
    qp.multi_gemv([layer], h32, outs=[acc], wscales=[wsc], oscale=scale, x_rot=(su, 1.0 / scale),
                  x_rms=(eps, w_ln), accumulate=True)
    torch.cuda.synchronize()
    assert torch.allclose(acc, got + 0.5, atol=tol, rtol=2e-3)


@pytest.mark.parametrize("qstr", ["tcomb_6_7_0.5_none_0.9", "tcq_6_none_0.9", "ldlq_2_8_none_1.0"])
@pytest.mark.parametrize("n", [1, 3])
def test_accumulate_into_a_live_buffer_under_split_k(qp, qstr, n):
    """accumulate=True with the truthful outs_zeroed=False on a few-rows x long-K layer (Llama-8B down_proj: the launch planner
    splits K over workgroups there): out must end up as h + y.  A memset in front of the split-K atomics would leave y alone
    (qpal_capi.hip zero_if_split)."""
    dev = torch.device("cuda", 0)
    k, m = 14336, 4096
    layer = qp.make_linear_from_info(qstr, qp.mem_op.dummy_linear_info(k, m, qstr, seed=5)).to(dev)
    gen = torch.Generator(device=dev).manual_seed(11)
    x = torch.randn(n, k, device=dev, generator=gen).half()
    (y,) = qp.multi_gemv([layer], x)
    h = torch.randn(n, m, device=dev, generator=gen) * 100.0
    acc = h.clone()
    qp.multi_gemv([layer], x, outs=[acc], accumulate=True)
    torch.cuda.synchronize()
    assert float((acc - h).abs().max()) > 1.0, "nothing was added"
    tol = 1e-5 * float(y.abs().max()) * 64 + 2.0 ** -17 * 400.0   # fp32 order of summation + the fp32 add onto |h| <= ~400
    assert torch.allclose(acc, h + y, atol=tol, rtol=1e-5), float((acc - (h + y)).abs().max())
    # and the flag the harness used to pass by mistake changes nothing
    acc2 = h.clone()
    qp.multi_gemv([layer], x, outs=[acc2], outs_zeroed=True, accumulate=True)
    torch.cuda.synchronize()
    assert torch.allclose(acc2, h + y, atol=tol, rtol=1e-5)


@pytest.mark.parametrize("n", [8192, 5120, 4096, 2048, 11008])
def test_rmsnorm_inside_the_rotation_launch(qp, n):
    """rotate(x fp32, rms=(eps, w)) == rotate(rmsnorm(x).half()): the widths whose rotation cannot run inside the GEMV staging
    (8192 = Llama-70B hidden, 5120 = 13B: K = 20, 11008 = 172 * 64) and, for comparison, the ones that can."""
    dev = torch.device("cuda", 0)
    had = qp.hadamard
    gen = torch.Generator(device=dev).manual_seed(n)
    x = torch.randn(3, n, device=dev, generator=gen) * torch.tensor([[0.02], [3.0], [400.0]], device=dev)  # residual-stream magnitudes
    w = (1.0 + 0.1 * torch.randn(n, device=dev, generator=gen)).half()
    su = (torch.randint(0, 2, (n,), device=dev, generator=gen) * 2 - 1).half()
    hadK, K = had.get_hadK(n)
    hk = None if K == 1 else hadK.T.contiguous().half().to(dev)
    eps = 1e-5
    xn = torch.nn.functional.rms_norm(x, (n,), w.float(), eps).half()
    ref = had.rotate(xn, hadK=hk, K=K, su=su, post_scale=1 / 32).float()
    got = had.rotate(x, hadK=hk, K=K, su=su, post_scale=1 / 32, in_mode=had.IN_F32, rms=(eps, w)).float()
    torch.cuda.synchronize()
    tol = 2.0 ** -9 * float(ref.abs().max())  # the norm's scalar is applied after the transform: inputs round at another point
    assert torch.allclose(got, ref, atol=tol, rtol=4e-3), float((got - ref).abs().max())


@pytest.mark.parametrize("qstr,inter", [("tcomb_6_7_0.5_none_0.9", 14336), ("tcq_4_none_0.9", 1024), ("ldlq_2_8_none_1.0", 2048)])
def test_swiglu_in_the_gemv_epilogue(qp, qstr, inter):
    """An up | gate pair with interleaved supertile rows (linear.interleave_up_gate): the rotating GEMV launch writes
    fp16 silu(gate) * up itself (act_out) — equal to the two projections + torch SwiGLU on the reference's fp16 rounding points."""
    dev = torch.device("cuda", 0)
    k = 4096
    up = qp.make_linear_from_info(qstr, qp.mem_op.dummy_linear_info(k, inter, qstr, seed=5, codebook_seed=3)).to(dev)
    gate = qp.make_linear_from_info(qstr, qp.mem_op.dummy_linear_info(k, inter, qstr, seed=6, codebook_seed=3)).to(dev)
    qp.share_codebooks([up, gate])
    il = qp.linear.interleave_up_gate(up, gate)
    assert il.out_features == 2 * inter
    gen = torch.Generator(device=dev).manual_seed(7)
    x = torch.randn(1, k, device=dev, generator=gen).half()
    su = (torch.randint(0, 2, (k,), device=dev, generator=gen) * 2 - 1).half()
    wu = (0.02 + 0.02 * torch.rand(inter, device=dev, generator=gen)).half()
    wg = (0.02 + 0.02 * torch.rand(inter, device=dev, generator=gen)).half()
    scale = 32.0
    u, g = qp.multi_gemv([up, gate], x, wscales=[wu, wg], oscale=scale, x_rot=(su, 1.0 / scale))
    ref = (torch.nn.functional.silu(g.half().float()).half().float() * u.half().float()).half()
    act = torch.full((1, inter), float("nan"), dtype=torch.float16, device=dev)
    (none,) = qp.multi_gemv([il], x, wscales=[qp.linear.interleave_rows(wu, wg)], oscale=scale, x_rot=(su, 1.0 / scale), act_out=act)
    # act_su: the NEXT projection's sign vector applied by this epilogue (exact: a sign flip)
    sdn = (torch.randint(0, 2, (inter,), device=dev, generator=gen) * 2 - 1).half()
    act_s = torch.empty_like(act)
    qp.multi_gemv([il], x, wscales=[qp.linear.interleave_rows(wu, wg)], oscale=scale, x_rot=(su, 1.0 / scale), act_out=act_s, act_su=sdn)
    torch.cuda.synchronize()
    assert torch.equal(act_s, act * sdn)
    torch.cuda.synchronize()
    assert none is None and bool(torch.isfinite(act).all())
    # same fp32 sums (the interleaved layer is planned like the pair), silu through the hardware reciprocal: <= 1-2 fp16 ulps
    err = (act.float() - ref.float()).abs()
    tol = 2.0 ** -9 * ref.float().abs() + 2.0 ** -9 * float(ref.float().abs().max()) * 2.0 ** -6
    assert bool((err <= tol).all()), float((err / tol).max())
    # and the rotation that follows reads it as plain fp16
    hk, K = qp.hadamard.get_hadK(inter)
    hkT = None if hk is None else hk.T.contiguous().half().to(dev)
    sd = (torch.randint(0, 2, (inter,), device=dev, generator=gen) * 2 - 1).half()
    a = qp.hadamard.rotate(act, hadK=hkT, K=K, su=sd, post_scale=1 / 8)
    ug = torch.cat([u, g], dim=1)
    b = qp.hadamard.rotate(ug, hadK=hkT, K=K, su=sd, post_scale=1 / 8, in_mode=qp.hadamard.IN_SWIGLU_F32)
    torch.cuda.synchronize()
    assert torch.allclose(a.float(), b.float(), atol=2.0 ** -8 * float(b.float().abs().max()), rtol=0)


@pytest.mark.parametrize("vocab,k,eps", [(128256, 4096, 1e-5), (32000, 2048, 0.0), (1000, 8192, 1e-5), (33, 4096, 1e-5)])
def test_lm_head_argmax_one_launch(qp, vocab, k, eps):
    """qpal_lm_head_argmax: final RMSNorm + fp16 lm_head GEMV + argmax (+ optional logits) against plain torch; twice (the
    ticket of the last-arriver reduction must be back at zero); ties take the lowest index like torch.argmax."""
    dev = torch.device("cuda", 0)
    nat = qp._native
    gen = torch.Generator(device=dev).manual_seed(vocab)
    h = torch.randn(k, device=dev, generator=gen) * 2.0
    w_ln = (1.0 + 0.1 * torch.randn(k, device=dev, generator=gen)).half()
    W = (torch.randn(vocab, k, device=dev, generator=gen) * 0.05).half()
    wsb = nat.lib().qpal_lm_head_ws_bytes(vocab)
    ws = torch.zeros(wsb // 4, device=dev)
    logits = torch.empty(vocab, device=dev)
    tok = torch.full((1,), -1, dtype=torch.long, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    if eps > 0:
        x = (h * torch.rsqrt((h * h).mean() + eps)).half() * w_ln
    else:
        x = h.half()
    ref = W.float() @ x.float()
    for rep in range(2):
        tok.fill_(-1)
        nat.check(nat.lib().qpal_lm_head_argmax(h.data_ptr(), w_ln.data_ptr() if eps > 0 else None, eps, W.data_ptr(),
                                                logits.data_ptr() if rep == 0 else None, tok.data_ptr(), ws.data_ptr(), wsb, vocab, k,
                                                stream), "qpal_lm_head_argmax")
        torch.cuda.synchronize()
        t = int(tok[0])
        assert 0 <= t < vocab
        tol = 2e-3 * float(ref.abs().max())  # x rounds once more or less than torch's fp16 pipeline; fp32 dot vs fp32 matmul
        assert float(ref[t]) >= float(ref.max()) - tol
        if rep == 0:
            assert torch.allclose(logits, ref, atol=tol, rtol=1e-3)
            assert t == int(logits.argmax())
    # exact ties: rows 7 and 500 (and the last row) are copies of the winner
    if vocab > 600:
        W2 = W.clone()
        W2[7] = W[t]; W2[500] = W[t]; W2[vocab - 1] = W[t]
        nat.check(nat.lib().qpal_lm_head_argmax(h.data_ptr(), w_ln.data_ptr() if eps > 0 else None, eps, W2.data_ptr(), None, tok.data_ptr(),
                                                ws.data_ptr(), wsb, vocab, k, stream), "qpal_lm_head_argmax")
        torch.cuda.synchronize()
        assert int(tok[0]) == min(7, t)


@pytest.mark.parametrize("case", ["nan", "neg_inf"])
def test_lm_head_argmax_degenerate_logits_give_a_valid_token(qp, case):
    """A diverged step (NaN in the residual stream, or every logit -inf) selects no row by comparison; the token written must
    still be a valid row index — the harness uses it as an embedding index on the device in the next graph replay
    (ADVICE r2).  torch.argmax returns an in-range index in both cases as well (0 for all -inf)."""
    dev = torch.device("cuda", 0)
    nat = qp._native
    vocab, k = 5000, 4096
    gen = torch.Generator(device=dev).manual_seed(1)
    W = (torch.randn(vocab, k, device=dev, generator=gen) * 0.05).half()
    h = torch.zeros(k, device=dev)
    if case == "nan":
        h[:] = torch.randn(k, device=dev, generator=gen)
        h[17] = float("nan")
    else:  # x = e_0: every logit is W[r][0] = -inf
        h[0] = 1.0
        W[:, 0] = float("-inf")
        W[:, 1:] = 0
    wsb = nat.lib().qpal_lm_head_ws_bytes(vocab)
    ws = torch.zeros(wsb // 4, device=dev)
    tok = torch.full((1,), -1, dtype=torch.long, device=dev)
    for rep in range(2):
        nat.check(nat.lib().qpal_lm_head_argmax(h.data_ptr(), None, 0.0, W.data_ptr(), None, tok.data_ptr(), ws.data_ptr(), wsb, vocab, k,
                                                torch.cuda.current_stream(dev).cuda_stream), "qpal_lm_head_argmax")
        torch.cuda.synchronize()
        assert 0 <= int(tok[0]) < vocab
    if case == "neg_inf":
        assert int(tok[0]) == int((W.float() @ h.half().float()).argmax()) == 0


@pytest.mark.parametrize("model", ["3_8b", "3_70b"])
def test_fused_decode_step_matches_modular_step(qp, model):
    """perf/decode_llama.py: the fused-glue step (6 launches per layer; 9 where the hidden width's rotation cannot run inside the
    GEMV staging: 70B's 8192 takes RMSNorm + rotation as one launch of its own) and the modular Incoherent* step (torch glue)
    produce the same normalised hidden state for the same random 2-layer model."""
    sys.path.insert(0, os.path.join(ROOT, "perf"))
    import decode_llama
    res = decode_llama.main(["--model", model, "--layers", "2", "--tokens", "4", "--context", "128", "--vocab", "4096"])
    chk = res["check"]
    assert chk is not None and res["ms_fused_glue"] is not None
    assert chk["max_abs_diff_final_norm"] <= 2.0 ** -7 * max(1.0, chk["max_abs_ref"]), chk
