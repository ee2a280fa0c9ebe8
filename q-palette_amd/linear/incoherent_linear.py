"""Incoherence wrappers around the packed linears (reference: lib/linear/incoherent_linear.py).

Same classes, constructor arguments, buffers and ``gen_layer_from_info`` loaders as the reference's
``IncoherentLinear`` (l.397-558), ``IncoherentMLP`` (l.275-394) and ``IncoherentSdpaAttention`` (l.28-271);
what differs is how a forward is launched on MI355X:

  reference (per projection group)              here
  x.half() * SU            elementwise          \\
  hadamard_transform       third-party kernel    |  ONE launch: qpal_hadamard (sign flip, both Hadamard factors,
  hadK @ x (K = 28 sizes)  cuBLAS                |  1/scale, and SwiGLU of up|gate for down_proj's input)
  / scale                  elementwise          /
  q/k/v (or up/gate)       1 launch per linear  \\  ONE multi-job launch; `* Wscale * scale` is the GEMV epilogue
  * Wscale * scale         2 elementwise each   /   (fp32, not re-rounded to fp16 three times)

The HF-model side of the attention module (rotary embedding, KV cache classes) belongs to the reference's
model fork, which is out of scope; ``forward`` takes ``position_embeddings`` = (cos, sin) and any
cache object with the ``update(k, v, layer_idx, kwargs)`` method and runs torch SDPA in between.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import hadamard as had
from .. import mem_op
from . import linear_class_for, make_linear_from_info, multi_gemv, rotation_fusable
from ._base import PackedLinearBase

_ACTS = {"silu": F.silu, "swish": F.silu, "gelu": F.gelu, "relu": F.relu}


def make_linear(info, use_simt=False):
    """reference incoherent_linear.py:13-26 (the substring order of the quantizer string matters)."""
    qstr = info["quant_info"]["quantizer_str"]
    try:
        return make_linear_from_info(qstr, info["linear_info"], use_simt=use_simt)
    except ValueError:
        return nn.Linear(info["in_features"], info["out_features"], bias=False)


def _had_buffer(n):
    """(fp16 transposed factor or None, K) — what the reference keeps as `had_left_*_T`."""
    hadK, K = had.get_hadK(n)
    return (hadK.T.contiguous().to(torch.float16) if hadK is not None else None), K


def _scaled_linears(layers, x16, wscales, scale, out=None, out_zeroed=False, prezero=None, x_rot=None):
    """[layer(x) * wscale * scale for layer in layers] as column blocks of ONE fp32 [n, sum m] buffer.
    out / out_zeroed: destination (already all zeros: a split-K layer then needs no memset of its own);
    prezero: a buffer the launch also zeroes for the NEXT projection of the block (o_proj after q|k|v, down_proj after
    up|gate) — the memset-free chain of multi_gemv."""
    n = x16.shape[0]
    widths = [l.out_features for l in layers]
    if out is None:
        out = torch.empty((n, sum(widths)), dtype=torch.float32, device=x16.device)
        out_zeroed = False
    if all(isinstance(l, PackedLinearBase) for l in layers) and n <= 16:
        outs = list(out.split(widths, dim=1))
        multi_gemv(layers, x16, outs=outs, outs_zeroed=out_zeroed, prezero=prezero, wscales=wscales, oscale=scale, x_rot=x_rot)
    else:  # unquantized ("default") layers or large batches: plain torch
        assert x_rot is None
        for l, w, o in zip(layers, wscales, out.split(widths, dim=1)):
            o.copy_(l(x16).float() * w.float() * scale)
        if prezero is not None:
            prezero.zero_()
    return out


def _rotated_linears(layers, x16, su, hadK, K, wscales, scale, **kw):
    """_scaled_linears(layers, rotate(x16 * su) / scale): where the sizes allow (k = 2048 / 4096, batch 1,
    tensor-core-order layers) the rotation runs inside the GEMV launches and costs no launch of its own."""
    n = x16.shape[0]
    if K == 1 and n <= 16 and all(isinstance(l, PackedLinearBase) for l in layers) and rotation_fusable(layers, n):
        return _scaled_linears(layers, x16, wscales, scale, x_rot=(su, 1.0 / scale), **kw)
    xr = had.rotate(x16, hadK=hadK, K=K, su=su, post_scale=1.0 / scale)
    return _scaled_linears(layers, xr, wscales, scale, **kw)


def _next_out(x16, layer):
    """Output buffer of the next projection of a block, to be zeroed by the launch before it (None if not useful)."""
    if isinstance(layer, PackedLinearBase) and x16.shape[0] <= 16:
        return torch.empty((x16.shape[0], layer.out_features), dtype=torch.float32, device=x16.device)
    return None


class IncoherentMLP(nn.Module):
    """Left-rotation-only MLP with one SU for up|gate (reference l.275-394)."""

    def __init__(self, hidden_size, intermediate_size, hidden_act, merge_ug=False, bias=False, dtype=torch.float16):
        super().__init__()
        assert bias is False, "bias is not supported"
        self.hidden_size, self.intermediate_size, self.dtype = hidden_size, intermediate_size, dtype
        self.up_proj = self.gate_proj = self.ug_proj = self.down_proj = None
        self.register_buffer("SU_ug", torch.ones(hidden_size, dtype=dtype))
        self.register_buffer("SU_dp", torch.ones(intermediate_size, dtype=dtype))
        hidden_had_T, self.hidden_K = _had_buffer(hidden_size)
        inter_had_T, self.inter_K = _had_buffer(intermediate_size)
        self.register_buffer("Wscale_ug", torch.ones(intermediate_size * 2, dtype=dtype), persistent=False)
        self.register_buffer("Wscale_dp", torch.ones(hidden_size, dtype=dtype), persistent=False)
        self.register_buffer("had_left_ug_T", hidden_had_T, persistent=False)
        self.register_buffer("had_left_dp_T", inter_had_T, persistent=False)
        self.scale = 64.0
        self.hidden_act = hidden_act
        self.act_fn = _ACTS[hidden_act]
        self.merge_ug = merge_ug

    # ---- fused pipeline (what forward runs)
    def _ug_raw(self, x16, prezero=None):
        """fp32 [n, 2I] = up | gate, already `* Wscale_ug * scale`."""
        inter = self.intermediate_size
        rot = (self.SU_ug, self.had_left_ug_T, self.hidden_K)
        if self.merge_ug:
            return _rotated_linears([self.ug_proj], x16, *rot, [self.Wscale_ug], self.scale, prezero=prezero)
        return _rotated_linears([self.up_proj, self.gate_proj], x16, *rot,
                                [self.Wscale_ug[:inter], self.Wscale_ug[inter:]], self.scale, prezero=prezero)

    def _dp_from_raw(self, ug, out=None):
        if self.hidden_act in ("silu", "swish"):
            xr = had.rotate(ug, hadK=self.had_left_dp_T, K=self.inter_K, su=self.SU_dp, post_scale=1.0 / self.scale,
                            in_mode=had.IN_SWIGLU_F32)
        else:
            up, gate = ug.half().split(self.intermediate_size, dim=-1)
            xr = had.rotate((self.act_fn(gate) * up).contiguous(), hadK=self.had_left_dp_T, K=self.inter_K, su=self.SU_dp,
                            post_scale=1.0 / self.scale)
        return _scaled_linears([self.down_proj], xr, [self.Wscale_dp], self.scale, out=out, out_zeroed=out is not None)

    def forward(self, input):
        n = len(self.SU_ug)
        x = input.reshape(-1, n).half()
        dp_out = _next_out(x, self.down_proj)  # zeroed by the up|gate launch: down_proj's split-K needs no memset
        y = self._dp_from_raw(self._ug_raw(x, prezero=dp_out), out=dp_out)
        return y.view(*input.shape[:-1], n).to(input.dtype)

    # ---- the reference's two-step interface (l.325-337), same semantics
    def compute_ug(self, x):
        up, gate = self._ug_raw(x.half()).half().split(self.intermediate_size, dim=-1)
        return self.act_fn(gate) * up

    def compute_dp(self, x):
        xr = had.rotate(x.half().contiguous(), hadK=self.had_left_dp_T, K=self.inter_K, su=self.SU_dp,
                        post_scale=1.0 / self.scale)
        return _scaled_linears([self.down_proj], xr, [self.Wscale_dp], self.scale).half()

    @staticmethod
    def gen_layer_from_info(config, info_up, info_gate, info_down, merge_ug=False, dummy=False, use_simt=False,
                            use_simt_u=None, use_simt_g=None, use_simt_d=None):
        mlp = IncoherentMLP(config.hidden_size, config.intermediate_size, config.hidden_act, merge_ug=merge_ug)
        if not dummy:
            mlp.SU_ug.data.copy_(info_up["SU"])
            mlp.SU_dp.data.copy_(info_down["SU"])
            mlp.Wscale_ug.data.copy_(torch.cat([info_up["Wscale"], info_gate["Wscale"]], dim=-1))
            mlp.Wscale_dp.data.copy_(info_down["Wscale"])
        use_simt_u = use_simt if use_simt_u is None else use_simt_u
        use_simt_g = use_simt if use_simt_g is None else use_simt_g
        use_simt_d = use_simt if use_simt_d is None else use_simt_d
        if merge_ug:
            cls = linear_class_for(info_up["quant_info"]["quantizer_str"], use_simt=use_simt_u)
            mlp.ug_proj = cls.gen_layer_from_info(cls.merge_infos(info_up["linear_info"], info_gate["linear_info"]))
        else:
            mlp.up_proj = make_linear(info_up, use_simt=use_simt_u)
            mlp.gate_proj = make_linear(info_gate, use_simt=use_simt_g)
        mlp.down_proj = make_linear(info_down, use_simt=use_simt_d)
        return mlp

    @staticmethod
    def gen_layer_from_quantizer_str_and_key(config, quant_dir, quantizer_str_up, quantizer_str_gate,
                                             quantizer_str_down, key_up, key_gate, key_down, merge_ug=False,
                                             dummy=False, use_simt=False, use_simt_u=None, use_simt_g=None,
                                             use_simt_d=None, model_key="3_8b"):
        infos = _load_infos(dummy, quant_dir, model_key,
                            [(quantizer_str_up, key_up, "mlp.up_proj"), (quantizer_str_gate, key_gate, "mlp.gate_proj"),
                             (quantizer_str_down, key_down, "mlp.down_proj")], config)
        return IncoherentMLP.gen_layer_from_info(config, *infos, merge_ug, dummy=dummy, use_simt=use_simt,
                                                 use_simt_u=use_simt_u, use_simt_g=use_simt_g, use_simt_d=use_simt_d)


def _load_infos(dummy, quant_dir, model_key, triples, config=None):
    model_key = mem_op.MODEL_KEYS.get(getattr(config, "_name_or_path", None), model_key)
    if not dummy:
        return [torch.load(f"{quant_dir}/{qstr}/{key}.pt") for qstr, key, _ in triples]
    return [mem_op.get_dummy_quant_results(model_key, layer_key, qstr) for qstr, _, layer_key in triples]


def _rotate_half(x):
    a, b = x[..., : x.shape[-1] // 2], x[..., x.shape[-1] // 2:]
    return torch.cat((-b, a), dim=-1)


class IncoherentSdpaAttention(nn.Module):
    """reference l.28-271.  `config` needs hidden_size, num_attention_heads, num_key_value_heads (and optionally
    head_dim, attention_dropout)."""

    def __init__(self, config, merge_qk=False, merge_kv=False, merge_qv=False, merge_qkv=False, layer_idx=None,
                 dtype=torch.float16):
        super().__init__()
        self.config = config
        self.attention_dropout = getattr(config, "attention_dropout", 0.0)
        self.hidden_size = config.hidden_size
        self.num_heads = config.num_attention_heads
        self.head_dim = getattr(config, "head_dim", None) or self.hidden_size // self.num_heads
        self.num_key_value_heads = config.num_key_value_heads
        self.num_key_value_groups = self.num_heads // self.num_key_value_heads
        self.kv_out = self.hidden_size * self.num_key_value_heads // self.num_heads
        self.q_proj = self.k_proj = self.v_proj = self.o_proj = None
        self.qk_proj = self.qkv_proj = self.kv_proj = self.qv_proj = None
        self.dtype, self.layer_idx = dtype, layer_idx
        self.register_buffer("SU_qkv", torch.ones(config.hidden_size, dtype=dtype))
        self.register_buffer("SU_o", torch.ones(config.hidden_size, dtype=dtype))
        hidden_had_T, self.hidden_K = _had_buffer(config.hidden_size)
        self.register_buffer("Wscale_qkv", torch.ones(config.hidden_size + 2 * self.kv_out, dtype=dtype), persistent=False)
        self.register_buffer("Wscale_o", torch.ones(config.hidden_size, dtype=dtype), persistent=False)
        self.register_buffer("had_left_qkv_T", hidden_had_T, persistent=False)
        self.register_buffer("had_left_o_T", hidden_had_T, persistent=False)
        self.scale = 64.0
        self.merge_qk, self.merge_kv, self.merge_qv, self.merge_qkv = merge_qk, merge_kv, merge_qv, merge_qkv
        assert sum([merge_qk, merge_kv, merge_qv, merge_qkv]) <= 1, \
            "Only one of merge_qk, merge_kv, merge_qv, merge_qkv can be True"

    def _qkv_layout(self):
        """(layers, their Wscale slices, column blocks as (name, width)) in the order of Wscale_qkv: the reference
        stores q|k|v, except q|v|k when q and v are merged (gen_layer_from_info l.205-211)."""
        H, kv = self.hidden_size, self.kv_out
        W = self.Wscale_qkv
        if self.merge_qkv:
            return [self.qkv_proj], [W], [("q", H), ("k", kv), ("v", kv)]
        if self.merge_qk:
            return [self.qk_proj, self.v_proj], [W[:H + kv], W[H + kv:]], [("q", H), ("k", kv), ("v", kv)]
        if self.merge_kv:
            return [self.q_proj, self.kv_proj], [W[:H], W[H:]], [("q", H), ("k", kv), ("v", kv)]
        if self.merge_qv:
            return [self.qv_proj, self.k_proj], [W[:H + kv], W[H + kv:]], [("q", H), ("v", kv), ("k", kv)]
        return ([self.q_proj, self.k_proj, self.v_proj], [W[:H], W[H:H + kv], W[H + kv:]],
                [("q", H), ("k", kv), ("v", kv)])

    def compute_qkv(self, input):
        n = len(self.SU_qkv)
        x = input.reshape(-1, n).half()
        layers, wscales, blocks = self._qkv_layout()
        # the q|k|v launch zeroes o_proj's output for the compute_o that follows (split-K without a memset)
        self._o_out = _next_out(x, self.o_proj)
        out = _rotated_linears(layers, x, self.SU_qkv, self.had_left_qkv_T, self.hidden_K, wscales, self.scale,
                               prezero=self._o_out).half()
        parts = dict(zip([b[0] for b in blocks], out.split([b[1] for b in blocks], dim=-1)))
        lead = input.shape[:-1]
        return (parts["q"].reshape(*lead, n), parts["k"].reshape(*lead, self.kv_out),
                parts["v"].reshape(*lead, self.kv_out))

    def compute_o(self, input):
        n = len(self.SU_o)
        x = input.reshape(-1, n).half()
        pre, self._o_out = getattr(self, "_o_out", None), None
        if pre is not None and pre.shape[0] != x.shape[0]:
            pre = None
        out = _rotated_linears([self.o_proj], x, self.SU_o, self.had_left_o_T, self.hidden_K, [self.Wscale_o], self.scale,
                               out=pre, out_zeroed=pre is not None)
        return out.half().view(*input.shape[:-1], n)

    def forward(self, hidden_states, attention_mask=None, position_ids=None, past_key_value=None,
                output_attentions=False, use_cache=False, cache_position=None, position_embeddings=None, **kwargs):
        bsz, q_len, _ = hidden_states.size()
        q, k, v = self.compute_qkv(hidden_states)
        q = q.view(bsz, q_len, self.num_heads, self.head_dim).transpose(1, 2)
        k = k.view(bsz, q_len, self.num_key_value_heads, self.head_dim).transpose(1, 2)
        v = v.view(bsz, q_len, self.num_key_value_heads, self.head_dim).transpose(1, 2)
        cos = sin = None
        if position_embeddings is not None:
            cos, sin = position_embeddings
            c, s = cos.unsqueeze(1), sin.unsqueeze(1)
            q, k = q * c + _rotate_half(q) * s, k * c + _rotate_half(k) * s
        if past_key_value is not None:
            k, v = past_key_value.update(k, v, self.layer_idx, {"sin": sin, "cos": cos, "cache_position": cache_position})
        mask = attention_mask[:, :, :, : k.shape[-2]] if attention_mask is not None else None
        # grouped-query attention without materialising the repeated keys / values (the reference's repeat_kv)
        attn = F.scaled_dot_product_attention(q, k.to(q.dtype), v.to(q.dtype), attn_mask=mask,
                                              dropout_p=self.attention_dropout if self.training else 0.0,
                                              is_causal=mask is None and q_len > 1,
                                              enable_gqa=self.num_key_value_groups > 1)
        attn = attn.transpose(1, 2).contiguous().view(bsz, q_len, -1)
        return self.compute_o(attn), None, past_key_value

    @staticmethod
    def gen_layer_from_info(config, layer_idx, info_q, info_k, info_v, info_o, merge_qk=False, merge_qv=False,
                            merge_kv=False, merge_qkv=False, dummy=False, use_simt=False, use_simt_q=None,
                            use_simt_k=None, use_simt_v=None, use_simt_o=None):
        attn = IncoherentSdpaAttention(config, merge_qk=merge_qk, merge_qv=merge_qv, merge_kv=merge_kv,
                                       merge_qkv=merge_qkv, layer_idx=layer_idx)
        if not dummy:
            attn.SU_qkv.data.copy_(info_q["SU"])
            attn.SU_o.data.copy_(info_o["SU"])
            order = [info_q, info_v, info_k] if merge_qv else [info_q, info_k, info_v]
            attn.Wscale_qkv.data.copy_(torch.cat([i["Wscale"] for i in order], dim=-1))
            attn.Wscale_o.data.copy_(info_o["Wscale"])
        simt = {"q": use_simt if use_simt_q is None else use_simt_q, "k": use_simt if use_simt_k is None else use_simt_k,
                "v": use_simt if use_simt_v is None else use_simt_v, "o": use_simt if use_simt_o is None else use_simt_o}
        infos = {"q": info_q, "k": info_k, "v": info_v}
        merged = "qk" if merge_qk else "kv" if merge_kv else "qv" if merge_qv else "qkv" if merge_qkv else ""
        if merged:
            cls = linear_class_for(infos[merged[0]]["quant_info"]["quantizer_str"], use_simt=simt[merged[0]])
            info = infos[merged[0]]["linear_info"]
            for name in merged[1:]:
                info = cls.merge_infos(info, infos[name]["linear_info"])
            setattr(attn, f"{merged}_proj", cls.gen_layer_from_info(info))
        for name in "qkv":
            if name not in merged:
                setattr(attn, f"{name}_proj", make_linear(infos[name], use_simt=simt[name]))
        attn.o_proj = make_linear(info_o, use_simt=simt["o"])
        return attn

    @staticmethod
    def gen_layer_from_quantizer_str_and_key(config, layer_idx, quant_dir, quantizer_str_q, quantizer_str_k,
                                             quantizer_str_v, quantizer_str_o, key_q, key_k, key_v, key_o,
                                             merge_qk=False, merge_qv=False, merge_kv=False, merge_qkv=False, dummy=False,
                                             use_simt=False, use_simt_q=None, use_simt_k=None, use_simt_v=None,
                                             use_simt_o=None, model_key="3_8b"):
        infos = _load_infos(dummy, quant_dir, model_key,
                            [(quantizer_str_q, key_q, "self_attn.q_proj"), (quantizer_str_k, key_k, "self_attn.k_proj"),
                             (quantizer_str_v, key_v, "self_attn.v_proj"), (quantizer_str_o, key_o, "self_attn.o_proj")],
                            config)
        return IncoherentSdpaAttention.gen_layer_from_info(
            config, layer_idx, *infos, merge_qk=merge_qk, merge_qv=merge_qv, merge_kv=merge_kv, merge_qkv=merge_qkv,
            dummy=dummy, use_simt=use_simt, use_simt_q=use_simt_q, use_simt_k=use_simt_k, use_simt_v=use_simt_v,
            use_simt_o=use_simt_o)


class IncoherentLinear(nn.Module):
    """Two-sided wrapper: y = (had_V((linear(had_U(x * SU) / scale)) * Wscale) * SV * scale) [+ bias]
    (reference l.397-558; `hadU` / `hadV` are the block sizes of the two rotations)."""

    def __init__(self, in_features, out_features, hadU, hadV, bias=False, dtype=torch.float16, use_linear=True):
        super().__init__()
        self.in_features, self.out_features, self.dtype = in_features, out_features, dtype
        self.linear = nn.Linear(in_features, out_features, bias=False, dtype=dtype) if use_linear else None
        if bias:
            self.register_buffer("bias", torch.ones(out_features))
        else:
            self.bias = None
        self.register_buffer("SU", torch.ones(in_features, dtype=dtype))
        self.register_buffer("SV", torch.ones(out_features, dtype=dtype))
        self.hadU, self.hadV = hadU, hadV
        had_left, self.K_left = had.get_hadK(hadU)
        had_right, self.K_right = had.get_hadK(hadV)
        self.register_buffer("Wscale", torch.ones(out_features, dtype=dtype), persistent=False)
        self.register_buffer("had_right", had_right.to(torch.float16) if had_right is not None else None,
                             persistent=False)
        self.register_buffer("had_left_T", had_left.T.contiguous().to(torch.float16) if had_left is not None else None,
                             persistent=False)
        self.scale = 32.0
        self.rot_info = "all"
        self.skip_l = self.skip_r = False

    def apply_rot_info(self):
        table = {"all": (False, False), "skip_l": (True, False), "skip_r": (False, True), "skip_lr": (True, True)}
        if self.rot_info not in table:
            raise ValueError(f"Invalid rot_info: {self.rot_info}")
        self.skip_l, self.skip_r = table[self.rot_info]

    def save_info(self, path, quant_info=None):
        info = {"in_features": self.in_features, "out_features": self.out_features, "hadU": self.hadU,
                "hadV": self.hadV, "dtype": self.dtype, "scale": self.scale, "Wscale": self.Wscale.detach().cpu(),
                "rot_info": self.rot_info, "linear_info": self.linear._info(),
                "bias": self.bias.detach().cpu() if self.bias is not None else None, "SU": self.SU.detach().cpu(),
                "SV": self.SV.detach().cpu(), "quant_info": quant_info}
        torch.save(info, path)

    def forward(self, input):
        n, m = len(self.SU), len(self.SV)
        x = input.reshape(-1, n).half()
        if not self.skip_l:
            x = had.rotate(x, hd=self.hadU, hadK=self.had_left_T, K=self.K_left, su=self.SU,
                           post_scale=1.0 / self.scale, round_mid=False)
        else:
            x = x / self.scale
        y = _scaled_linears([self.linear], x, [self.Wscale], 1.0)  # fp32 [rows, m]
        if not self.skip_r:
            y = had.rotate(y, hd=self.hadV, hadK=self.had_right, K=self.K_right, sv=(self.SV * self.scale),
                           in_mode=had.IN_F32, round_mid=False)
        else:
            y = y * self.scale
        y = y.view(*input.shape[:-1], m).to(input.dtype)
        return y + self.bias if self.bias is not None else y

    @staticmethod
    def gen_layer_from_info(info, merge_layers=False, dummy=False, use_simt=False):
        layer = IncoherentLinear(info["in_features"], info["out_features"], info.get("hadU", info["in_features"]),
                                 info.get("hadV", info["out_features"]), bias=info["bias"] is not None,
                                 dtype=info["dtype"], use_linear=False)
        if not dummy:
            if info["bias"] is not None:
                layer.bias.data.copy_(info["bias"])
            layer.SU.data.copy_(info["SU"])
            layer.SV.data.copy_(info["SV"])
            layer.Wscale.data.copy_(info["Wscale"])
        if info["quant_info"] is not None:
            layer.linear = make_linear(info, use_simt=use_simt)
        if info["quant_info"] is not None and "rot_info" in info["quant_info"]:
            layer.rot_info = info["quant_info"]["rot_info"]
        else:
            layer.rot_info = info.get("rot_info", "all")
        if merge_layers:
            layer.apply_rot_info()
        return layer

    @staticmethod
    def gen_layer_from_quantizer_str_and_key(config, quant_dir, quantizer_str, key, merge_layers=False, dummy=False,
                                             use_simt=False, model_key="3_8b"):
        layer_key = key.split("_", 1)[1] if dummy else None
        info = _load_infos(dummy, quant_dir, model_key, [(quantizer_str, key, layer_key)], config)[0]
        return IncoherentLinear.gen_layer_from_info(info, merge_layers=merge_layers, dummy=dummy, use_simt=use_simt)
