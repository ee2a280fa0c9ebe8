#!/usr/bin/env python3
"""End-to-end decode loop of a Llama-shaped model built from THIS package's modules.

Counterpart of the reference's eval/measure_latency.py (HF model + StaticCache + CUDA graphs): every decoder layer is
``IncoherentSdpaAttention`` + ``IncoherentMLP`` (qpalette_amd, dummy packed weights of the reference's shapes: quantizer
string per linear or a published qdict with its merge_info), with a static KV cache, rotary embedding, RMSNorm, residuals,
embedding and an fp16 lm_head.  One decode step is captured in a HIP graph and replayed.

What is NOT this package's work and is left as plain torch ops (the reference gets them fused by torch.compile, which is
Triton and therefore not used here): RMSNorm, rotary embedding, KV-cache update, SDPA, residual adds, argmax — a few
hundred small launches per token.  The script reports the whole-step rate and, next to it, the rate of the quantized
projections alone (same graph without the glue), so the two are not confused.

    python perf/decode_llama.py [--quantizer tcomb_6_7_0.5_none_0.9 | --qdict figure1d] [--context 1024] [--tokens 64]
"""
import argparse
import json
import math
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn

import qpalette_amd as qp

LINEARS = ["self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj",
           "mlp.gate_proj", "mlp.up_proj", "mlp.down_proj"]


class RMSNorm(nn.Module):
    def __init__(self, n, eps=1e-5):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(n, dtype=torch.float16), requires_grad=False)
        self.eps = eps

    def forward(self, x):
        return torch.nn.functional.rms_norm(x, (x.shape[-1],), self.weight, self.eps)  # one launch


class StaticKV:
    """Fixed-size key/value cache with the `update(k, v, layer_idx, kwargs)` method the attention module calls."""

    def __init__(self, nlayers, kv_heads, head_dim, max_len, device):
        shape = (1, kv_heads, max_len, head_dim)
        self.k = [torch.zeros(shape, dtype=torch.float16, device=device) for _ in range(nlayers)]
        self.v = [torch.zeros(shape, dtype=torch.float16, device=device) for _ in range(nlayers)]

    def update(self, k, v, layer_idx, kwargs):
        pos = kwargs["cache_position"]
        self.k[layer_idx].index_copy_(2, pos, k)
        self.v[layer_idx].index_copy_(2, pos, v)
        return self.k[layer_idx], self.v[layer_idx]


def _info(model_key, layer, key, qstr, device, gen):
    li = qp.mem_op.get_layer_info(model_key)[key]
    k, m = li["in_features"], li["out_features"]
    return {"quant_info": qp.mem_op.get_quant_info(qstr), "in_features": k, "out_features": m, "dtype": torch.float16,
            "bias": None,
            "linear_info": qp.mem_op.dummy_linear_info(k, m, qstr, seed=layer * 16 + LINEARS.index(key), device=device,
                                                       codebook_seed=777),
            "SU": (torch.randint(0, 2, (k,), device=device, generator=gen) * 2 - 1).half(),
            # small output scales keep the random model's residual stream finite in fp16 over 32 layers
            "Wscale": (0.001 + 0.001 * torch.rand(m, device=device, generator=gen)).half()}


class DecoderLayer(nn.Module):
    def __init__(self, cfg, model_key, layer, qof, merges, device, gen):
        super().__init__()
        inf = {key: _info(model_key, layer, key, *qof(layer, key)[:1], device, gen) for key in LINEARS}
        simt = {key: qof(layer, key)[1] for key in LINEARS}
        q, k, v, o, g, u, d = LINEARS
        self.self_attn = qp.IncoherentSdpaAttention.gen_layer_from_info(
            cfg, layer, inf[q], inf[k], inf[v], inf[o], merge_qk="merge_qk" in merges, merge_qv="merge_qv" in merges,
            merge_kv="merge_kv" in merges, merge_qkv="merge_qkv" in merges, use_simt_q=simt[q], use_simt_k=simt[k],
            use_simt_v=simt[v], use_simt_o=simt[o]).to(device)
        self.mlp = qp.IncoherentMLP.gen_layer_from_info(cfg, inf[u], inf[g], inf[d], merge_ug="merge_ug" in merges,
                                                        use_simt_u=simt[u], use_simt_g=simt[g], use_simt_d=simt[d]).to(device)
        self.input_layernorm = RMSNorm(cfg.hidden_size).to(device)
        self.post_attention_layernorm = RMSNorm(cfg.hidden_size).to(device)

    def forward(self, h, rope, mask, cache, pos, glue=True):
        if not glue:  # the quantized projections alone: q|k|v, o, up|gate (+SwiGLU rotation), down
            q, _, _ = self.self_attn.compute_qkv(h)
            self.self_attn.compute_o(q)
            self.mlp(h)
            return h
        a, _, _ = self.self_attn(self.input_layernorm(h), attention_mask=mask, past_key_value=cache, cache_position=pos,
                                 position_embeddings=rope)
        h = h + a
        return h + self.mlp(self.post_attention_layernorm(h))


def main(argv=None, quiet=False):
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="3_8b", choices=sorted(qp.mem_op.LAYER_INFO))
    ap.add_argument("--quantizer", default="tcomb_6_7_0.5_none_0.9")
    ap.add_argument("--qdict", default=None, help="perf/qdicts/<name>.json (figure1c, figure1d) instead of --quantizer")
    ap.add_argument("--layers", type=int, default=0)
    ap.add_argument("--context", type=int, default=1024, help="static KV-cache length attended over")
    ap.add_argument("--tokens", type=int, default=64)
    ap.add_argument("--start-pos", type=int, default=8, help="position of the first timed token (the cache below it is attended over)")
    ap.add_argument("--vocab", type=int, default=128256)
    ap.add_argument("--no-fused", action="store_true", help="skip the fused-glue step (third figure)")
    ap.add_argument("--no-swiglu-epilogue", action="store_true", help="fused step: up|gate as fp32 outputs + SwiGLU inside the rotation launch")
    ap.add_argument("--torch-lm-head", action="store_true", help="fused step: final norm, lm_head and argmax as torch ops (hipBLASLt GEMV)")
    ap.add_argument("--no-split-attention", action="store_true", help="one workgroup per query head at every context length")
    ap.add_argument("--no-modular", action="store_true", help="time the fused-glue step only (profiling)")
    ap.add_argument("--no-k28-fusion", action="store_true", help="fused step: down_proj's 28 x 512 rotation as a launch of its own (round 2)")
    args = ap.parse_args(argv)
    if not torch.cuda.is_available():
        raise SystemExit("needs a GPU")
    dev = torch.device("cuda", 0)
    li = qp.mem_op.get_layer_info(args.model)
    H, I = li["mlp.gate_proj"]["in_features"], li["mlp.gate_proj"]["out_features"]
    kv_out = li["self_attn.k_proj"]["out_features"]
    head_dim = 128
    cfg = types.SimpleNamespace(hidden_size=H, intermediate_size=I, hidden_act="silu", num_attention_heads=H // head_dim,
                                num_key_value_heads=kv_out // head_dim, head_dim=head_dim, attention_dropout=0.0)
    nlayers = args.layers or li["nlayers"]
    qdict, merge_info = None, None
    if args.qdict:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "qdicts", args.qdict + ".json")) as f:
            data = json.load(f)
        qdict, merge_info = data["qdict"], data["merge_info"]

    def qof(layer, key):
        if qdict is None:
            return args.quantizer, False
        q, simt = qdict[f"{layer}_{key}"]
        return q, simt == "1"

    gen = torch.Generator(device=dev).manual_seed(1234)
    layers = nn.ModuleList([DecoderLayer(cfg, args.model, i, qof, set(merge_info[i]) if merge_info else set(), dev, gen)
                            for i in range(nlayers)])
    qp.share_codebooks([m for m in layers.modules() if hasattr(m, "tlut") or hasattr(m, "lut")])
    embed = (torch.randn(args.vocab, H, device=dev, generator=gen) * 0.5).half()
    lm_head = (torch.randn(args.vocab, H, device=dev, generator=gen) * 0.02).half()
    norm = RMSNorm(H).to(dev)
    cache = StaticKV(nlayers, cfg.num_key_value_heads, head_dim, args.context, dev)
    inv_freq = 1.0 / (500000.0 ** (torch.arange(0, head_dim, 2, device=dev).float() / head_dim))
    tok = torch.zeros(1, dtype=torch.long, device=dev)
    pos = torch.zeros(1, dtype=torch.long, device=dev)
    out_tok = torch.zeros(1, dtype=torch.long, device=dev)
    ar = torch.arange(args.context, device=dev)

    def step(glue=True):
        h = embed[tok].view(1, 1, H)
        rope = mask = None
        if glue:
            ang = pos.float()[:, None] * inv_freq[None, :]
            emb = torch.cat((ang, ang), dim=-1)[None]                       # [1, 1, head_dim]
            rope = (emb.cos().half(), emb.sin().half())
            mask = torch.where(ar <= pos, 0.0, float("-inf")).half().view(1, 1, 1, -1)
        for layer in layers:
            h = layer(h, rope, mask, cache, pos, glue=glue)
        if glue:
            logits = norm(h).view(1, H) @ lm_head.T
            out_tok.copy_(logits.argmax(-1))
        return h

    def _time_tokens(g, feed):
        """ms-free seconds per token of `args.tokens` replays of the captured step.  The loop that is timed is the loop that is warmed
        up: with `feed` every token first copies the sampled token in and sets the position (two tiny launches whose FIRST use
        loads their code objects — ~28 ms that round 3 had inside the timed region: 284 tok/s at 20 tokens against 472 at 64).
        Timed with events on the replay stream around the whole loop, the host loop runs ahead of the GPU."""
        def one(i):
            if feed:                           # the next step consumes the sampled token at the next position
                tok.copy_(out_tok)
                pos.fill_(min(args.context - 1, args.start_pos + i))
            g.replay()
        for i in range(8):
            one(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(args.tokens):
            one(i)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / args.tokens

    def timed(glue):
        s = torch.cuda.Stream(dev)
        with torch.cuda.stream(s):
            step(glue)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                step(glue)
            return _time_tokens(g, glue)

    # ---- fused glue (MI355X decoder block): the residual stream stays fp32; RMSNorm + sign flip + Hadamard run inside the
    # q|k|v and up|gate launches (x_rms / x_rot on the fp32 stream), o_proj and down_proj ADD into the stream (accumulate),
    # rotary embedding + KV-cache append + attention over the cache are one launch (qpal_attn_rope_decode), the SwiGLU rotation
    # one: 6 launches per layer instead of ~41.
    nat = qp._native
    h32 = torch.zeros(1, H, dtype=torch.float32, device=dev)
    nq, nkv = cfg.num_attention_heads, cfg.num_key_value_heads
    q16 = torch.zeros(1, nq, 1, head_dim, dtype=torch.float16, device=dev)
    a16 = torch.zeros(1, H, dtype=torch.float16, device=dev)
    qkv32 = torch.zeros(1, H + 2 * kv_out, dtype=torch.float32, device=dev)
    ug32 = torch.zeros(1, 2 * I, dtype=torch.float32, device=dev)
    act16 = torch.zeros(1, I, dtype=torch.float16, device=dev)
    eps = layers[0].input_layernorm.eps
    # long caches: split-context attention (one workspace serves every layer: launches are stream-ordered)
    attn_ws_bytes = 0 if args.no_split_attention else nat.lib().qpal_attn_ws_bytes(nq, nkv, head_dim, args.context)
    attn_ws = torch.zeros(max(attn_ws_bytes, 4) // 4, dtype=torch.float32, device=dev)
    lm_ws_bytes = nat.lib().qpal_lm_head_ws_bytes(args.vocab)
    lm_ws = torch.zeros(lm_ws_bytes // 4, dtype=torch.float32, device=dev)
    want_hidden = [False]  # the parity check below wants the normalised hidden state back (an extra torch norm, not timed)

    rot_in_gemv = bool(qp.ops.can_fuse_rotation(1, H))  # k in {2048, 4096}: the GEMV staging rotates x itself

    def _hid(mod, name):
        """(hadK^T fp16 on the device or None, K) of a module's rotation of the hidden width (cached on the module)"""
        if not hasattr(mod, name):
            hk, K = qp.hadamard.get_hadK(H)
            setattr(mod, name, (None if hk is None else hk.T.contiguous().half().to(dev), K))
        return getattr(mod, name)

    def k28_in_gemv(mlp):
        """down_proj's rotation inside its own launch: k = 14336 = 28 x 512, a tensor-core-order layer whose codebook image can lend
        the rotation its scratch (every TCQ codec; --no-k28-fusion: the qpal_hadamard launch of round 2)."""
        if args.no_k28_fusion or mlp.inter_K <= 1 or not qp.ops.can_fuse_rotation(1, mlp.intermediate_size, mlp.inter_K):
            return False
        d = mlp.down_proj
        if isinstance(d, qp.VQLinearPackTensorCore):
            idx = d.lut_bits if d.vec_sz == 2 else (2 * d.lut_bits if d.lut_bits <= 6 else d.lut_bits)
            return (4 << (idx + min(15 - idx, 5))) >= 40 * 1024
        return isinstance(d, (qp.QTIPLinearTCQ, qp.CombtLinearTCQ)) and qp.linear._codec_key(d)[0] != "single"

    def fused_layer(idx, layer, mask):
        att, mlp = layer.self_attn, layer.mlp
        proj, wsc, blocks = att._qkv_layout()
        widths = [l.out_features for l in proj]
        if rot_in_gemv:
            qp.multi_gemv(proj, h32, outs=list(qkv32.split(widths, dim=1)), wscales=wsc, oscale=att.scale,
                          x_rot=(att.SU_qkv, 1.0 / att.scale), x_rms=(eps, layer.input_layernorm.weight))
        else:  # wider hidden sizes (70B: 8192): RMSNorm + rotation as ONE launch of their own, then the plain GEMV launch
            hk, K = _hid(att, "_hadk_hidden")
            xr = qp.hadamard.rotate(h32, hadK=hk, K=K, su=att.SU_qkv, post_scale=1.0 / att.scale, in_mode=qp.hadamard.IN_F32,
                                    rms=(eps, layer.input_layernorm.weight))
            qp.multi_gemv(proj, xr, outs=list(qkv32.split(widths, dim=1)), wscales=wsc, oscale=att.scale)
        parts = dict(zip([b[0] for b in blocks], qkv32.split([b[1] for b in blocks], dim=1)))
        with torch.cuda.device(dev):
            rc = nat.lib().qpal_attn_rope_decode(parts["q"].data_ptr(), parts["k"].data_ptr(), parts["v"].data_ptr(),
                                                 cache.k[idx].data_ptr(), cache.v[idx].data_ptr(), a16.data_ptr(), pos.data_ptr(),
                                                 inv_freq.data_ptr(), nq, nkv, head_dim, args.context, 1.0 / math.sqrt(head_dim),
                                                 attn_ws.data_ptr() if attn_ws_bytes else None, attn_ws_bytes,
                                                 torch.cuda.current_stream(dev).cuda_stream)
        nat.check(rc, "qpal_attn_rope_decode")
        if rot_in_gemv:
            qp.multi_gemv([att.o_proj], a16, outs=[h32], wscales=[att.Wscale_o], oscale=att.scale,
                          x_rot=(att.SU_o, 1.0 / att.scale), accumulate=True)
        else:
            hk, K = _hid(att, "_hadk_hidden")
            xr = qp.hadamard.rotate(a16, hadK=hk, K=K, su=att.SU_o, post_scale=1.0 / att.scale)
            qp.multi_gemv([att.o_proj], xr, outs=[h32], wscales=[att.Wscale_o], oscale=att.scale, accumulate=True)
        inter = mlp.intermediate_size
        if mlp.merge_ug:
            ugl, ugw = [mlp.ug_proj], [mlp.Wscale_ug]
        else:
            ugl, ugw = [mlp.up_proj, mlp.gate_proj], [mlp.Wscale_ug[:inter], mlp.Wscale_ug[inter:]]
        ug_outs = list(ug32.split([l.out_features for l in ugl], dim=1))
        if rot_in_gemv and not args.no_swiglu_epilogue:
            # up | gate as ONE layer with interleaved supertile rows: the launch's epilogue writes fp16 silu(gate) * up itself
            if not hasattr(mlp, "_ug_il"):
                il = qp.linear.interleave_up_gate(mlp.ug_proj, None) if mlp.merge_ug else qp.linear.interleave_up_gate(mlp.up_proj, mlp.gate_proj)
                qp.share_codebooks([il, mlp.down_proj] + ugl)
                mlp._ug_il = il
                mlp._ug_il_w = qp.linear.interleave_rows(mlp.Wscale_ug[:inter], mlp.Wscale_ug[inter:])
            fuse28 = k28_in_gemv(mlp)
            # with the fused rotation the gate|up epilogue also applies down_proj's sign vector (a sign flip: exact), so the rotation
            # inside every down_proj workgroup reads one 28 KiB vector instead of two
            qp.multi_gemv([mlp._ug_il], h32, wscales=[mlp._ug_il_w], oscale=mlp.scale, x_rot=(mlp.SU_ug, 1.0 / mlp.scale),
                          x_rms=(eps, layer.post_attention_layernorm.weight), act_out=act16, act_su=mlp.SU_dp if fuse28 else None)
            if fuse28:  # the 28 x 512 rotation inside down_proj's x staging (csrc/rot_k28.h): no launch of its own
                qp.multi_gemv([mlp.down_proj], act16, outs=[h32], wscales=[mlp.Wscale_dp], oscale=mlp.scale,
                              x_rot=(None, 1.0 / mlp.scale, mlp.had_left_dp_T, mlp.inter_K), accumulate=True)
                return
            xr = qp.hadamard.rotate(act16, hadK=mlp.had_left_dp_T, K=mlp.inter_K, su=mlp.SU_dp, post_scale=1.0 / mlp.scale)
            qp.multi_gemv([mlp.down_proj], xr, outs=[h32], wscales=[mlp.Wscale_dp], oscale=mlp.scale,
                          accumulate=True)
            return
        if rot_in_gemv:
            qp.multi_gemv(ugl, h32, outs=ug_outs, wscales=ugw, oscale=mlp.scale, x_rot=(mlp.SU_ug, 1.0 / mlp.scale),
                          x_rms=(eps, layer.post_attention_layernorm.weight))
        else:
            hk, K = _hid(mlp, "_hadk_hidden")
            xr = qp.hadamard.rotate(h32, hadK=hk, K=K, su=mlp.SU_ug, post_scale=1.0 / mlp.scale, in_mode=qp.hadamard.IN_F32,
                                    rms=(eps, layer.post_attention_layernorm.weight))
            qp.multi_gemv(ugl, xr, outs=ug_outs, wscales=ugw, oscale=mlp.scale)
        xr = qp.hadamard.rotate(ug32, hadK=mlp.had_left_dp_T, K=mlp.inter_K, su=mlp.SU_dp, post_scale=1.0 / mlp.scale,
                                in_mode=qp.hadamard.IN_SWIGLU_F32)
        qp.multi_gemv([mlp.down_proj], xr, outs=[h32], wscales=[mlp.Wscale_dp], oscale=mlp.scale,
                      accumulate=True)

    def fused_step():
        h32.copy_(embed[tok].view(1, H))
        mask = torch.where(ar <= pos, 0.0, float("-inf")).half().view(1, 1, 1, -1)
        for idx, layer in enumerate(layers):
            fused_layer(idx, layer, mask)
        if args.torch_lm_head:
            hn = norm(h32.half().view(1, 1, H))
            logits = hn.view(1, H) @ lm_head.T
            out_tok.copy_(logits.argmax(-1))
            return hn
        with torch.cuda.device(dev):  # final RMSNorm + lm_head GEMV + argmax: one launch
            rc = nat.lib().qpal_lm_head_argmax(h32.data_ptr(), norm.weight.data_ptr(), norm.eps, lm_head.data_ptr(), None,
                                               out_tok.data_ptr(), lm_ws.data_ptr(), lm_ws_bytes, args.vocab, H,
                                               torch.cuda.current_stream(dev).cuda_stream)
        nat.check(rc, "qpal_lm_head_argmax")
        return norm(h32.half().view(1, 1, H)) if want_hidden[0] else None

    def _tc(l):
        return isinstance(l, qp.linear._base.PackedLinearBase) and not isinstance(l, qp.VQLinearPackSIMT)

    fusable = (not args.no_fused and
               all(all(_tc(p_) for p_ in l.self_attn._qkv_layout()[0]) and _tc(l.self_attn.o_proj) and _tc(l.mlp.down_proj)
                   and all(_tc(p_) for p_ in ([l.mlp.ug_proj] if l.mlp.merge_ug else [l.mlp.up_proj, l.mlp.gate_proj]))
                   and (not rot_in_gemv or (qp.linear.rotation_fusable(l.self_attn._qkv_layout()[0], 1)
                                            and qp.linear.rotation_fusable([l.self_attn.o_proj], 1)))
                   for l in layers))

    def timed_fused():
        s = torch.cuda.Stream(dev)
        with torch.cuda.stream(s):
            fused_step()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                fused_step()
            return _time_tokens(g, True)

    # the fused step computes what the modular step computes (fp32 residual stream instead of fp16: small differences)
    check = None
    if fusable and not args.no_modular:
        tok.zero_()
        pos.fill_(3)
        ref_h = norm(step(True).view(1, 1, H)).float()
        tok.zero_()
        pos.fill_(3)
        want_hidden[0] = True
        got_h = fused_step().float()
        want_hidden[0] = False
        check = {"max_abs_diff_final_norm": float((ref_h - got_h).abs().max()), "max_abs_ref": float(ref_h.abs().max())}
    if args.no_modular:
        if not fusable:
            raise SystemExit("--no-modular: this configuration has no fused-glue step")
        t_fused = timed_fused()
        # kernel launches of one fused step: per layer q|k|v (+ its rotation where the GEMV cannot rotate), attention, o (+ rotation),
        # up|gate (+ rotation), SwiGLU rotation, down; + the norm / lm_head / argmax launch (the embedding row copy is a memcpy node)
        per_layer = (5 if k28_in_gemv(layers[0].mlp) else 6) if rot_in_gemv else 9
        if not quiet:
            print(json.dumps({"model": args.model, "layers": nlayers, "quantizer": args.qdict or args.quantizer, "context": args.context,
                              "tokens_per_s_fused_glue": 1.0 / t_fused, "ms_fused_glue": t_fused * 1e3}))
        return {"ms_whole_step": None, "ms_fused_glue": t_fused * 1e3, "check": None, "launches_per_token": per_layer * nlayers + 1}
    t_full = timed(True)
    t_proj = timed(False)
    t_fused = timed_fused() if fusable else None
    finite = bool(torch.isfinite(step(True)).all())
    packed = sum(t.numel() * t.element_size() for m in layers.modules() for name in ("trellis", "trellis1", "trellis2", "qweight")
                 if (t := getattr(m, name, None)) is not None)
    print(json.dumps({
        "what": "decode step of a Llama-shaped model: incoherent quantized projections (this package) + torch glue",
        "model": args.model, "layers": nlayers, "quantizer": args.qdict or args.quantizer, "context": args.context,
        "tokens_per_s_whole_step": 1.0 / t_full, "ms_whole_step": t_full * 1e3,
        "tokens_per_s_projections_only": 1.0 / t_proj, "ms_projections_only": t_proj * 1e3,
        "glue_ms": (t_full - t_proj) * 1e3, "packed_weight_GB": packed / 1e9, "lm_head_GB": lm_head.numel() * 2 / 1e9,
        "tokens_per_s_fused_glue": (1.0 / t_fused) if t_fused else None, "ms_fused_glue": t_fused * 1e3 if t_fused else None,
        "fused_vs_modular": check, "finite": finite}))
    return {"ms_whole_step": t_full * 1e3, "ms_fused_glue": t_fused * 1e3 if t_fused else None, "check": check}


if __name__ == "__main__":
    main()
