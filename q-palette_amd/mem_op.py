"""Quantizer-string grammar, Llama layer shapes and synthetic packed weights.

Mirrors the parts of the reference's lib/utils/mem_op.py that the hot path's callers need:
``LAYER_INFO`` (l.2-189), ``get_quant_info`` (l.271-307) and ``get_dummy_quant_results`` (l.198-269),
with one deliberate difference in the synthetic generator: packed buffers are filled with FULL-range
random bits (every 16-bit pattern is a valid trellis stream, every code is valid) instead of
``randint(0, 2**14)`` / ``randint(0, 2**30)``, so that the top bits of every word are exercised.
"""
import math

import torch

_LINEAR_KEYS = ["self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj",
                "mlp.gate_proj", "mlp.up_proj", "mlp.down_proj"]


def _llama(nlayers, hidden, kv, inter):
    shapes = {"self_attn.q_proj": (hidden, hidden), "self_attn.k_proj": (hidden, kv), "self_attn.v_proj": (hidden, kv),
              "self_attn.o_proj": (hidden, hidden), "mlp.gate_proj": (hidden, inter), "mlp.up_proj": (hidden, inter),
              "mlp.down_proj": (inter, hidden)}
    info = {"nlayers": nlayers}
    for key, (i, o) in shapes.items():
        info[key] = {"in_features": i, "out_features": o}
    return info


# (in_features, out_features) per linear; same table as the reference's LAYER_INFO
LAYER_INFO = {
    "2_7b": _llama(32, 4096, 4096, 11008),
    "2_13b": _llama(40, 5120, 5120, 13824),
    "2_70b": _llama(80, 8192, 1024, 28672),
    "3_8b": _llama(32, 4096, 1024, 14336),
    "3_3b": _llama(28, 3072, 1024, 8192),
    "3_1b": _llama(16, 2048, 512, 8192),
    "3_70b": _llama(80, 8192, 1024, 28672),
}
LINEAR_KEYS = list(_LINEAR_KEYS)
# HF model id -> key of LAYER_INFO (reference lib/config.py:1-5)
MODEL_KEYS = {"meta-llama/Llama-3.1-8B": "3_8b", "meta-llama/Llama-3.2-1B": "3_1b", "meta-llama/Llama-3.2-3B": "3_3b"}


def get_layer_info(model_key):
    return LAYER_INFO["3_8b" if model_key == "3_8b_0" else model_key]


def get_quant_info(quantizer_str):
    """tcq_{KV}_{hess}_{scale} | tcomb_{KV1}_{KV2}_{ratio}_{hess}_{scale} | comb_... | ldlq_{vec}_{bits}_{hess}_{scale}
    | default.  tlut_bits = 9 if KV <= 8 else KV + 1 (reference mem_op.py:274, 284)."""
    parts = quantizer_str.split("_")
    if quantizer_str.startswith("tcq"):
        kv = int(parts[1])
        return {"quantizer_str": quantizer_str, "quantizer": "tcq_ldlq", "KV": kv, "V": 2,
                "tlut_bits": 9 if kv <= 8 else kv + 1}
    if quantizer_str.startswith("tcomb") or quantizer_str.startswith("comb"):
        kv1, kv2, ratio = int(parts[1]), int(parts[2]), float(parts[3])
        return {"quantizer_str": quantizer_str,
                "quantizer": "combt_ldlq" if quantizer_str.startswith("tcomb") else "comb_ldlq",
                "KV": [kv1, kv2], "V": 2, "tlut_bits": 9 if kv2 <= 8 else kv2 + 1, "ratio": ratio}
    if quantizer_str.startswith("ldlq") or quantizer_str.startswith("sq") or quantizer_str.startswith("vq"):
        return {"quantizer_str": quantizer_str, "quantizer": "vq_ldlq", "vec_sz": int(parts[1]),
                "lut_bits": int(parts[2])}
    if quantizer_str == "default":
        return {"quantizer_str": quantizer_str}
    raise ValueError(f"Unknown quantizer: {quantizer_str}")


def bits_per_weight(quantizer_str):
    """tcq: KV/2; tcomb/comb: (KV1+KV2)/4; ldlq: bits/vec (reference solve_lat_const.py:5-39)."""
    qi = get_quant_info(quantizer_str)
    if "lut_bits" in qi:
        return qi["lut_bits"] / qi["vec_sz"]
    if isinstance(qi.get("KV"), list):
        return (qi["KV"][0] + qi["KV"][1]) / 4
    if "KV" in qi:
        return qi["KV"] / 2
    return 16.0


def _rand_i16(shape, gen, device):
    return torch.randint(-2 ** 15, 2 ** 15, shape, dtype=torch.int16, device=device, generator=gen)


def _rand_i32(shape, gen, device):
    return torch.randint(-2 ** 31, 2 ** 31, shape, dtype=torch.int32, device=device, generator=gen)


def dummy_linear_info(in_features, out_features, quantizer_str, seed=0, device="cpu", codebook_seed=None, part=None):
    """Random packed weights with the reference's shapes/dtypes for one linear (``linear_info`` dict).
    codebook_seed: draw the codebook from its own generator (same seed -> same codebook in every layer, as in
    real checkpoints where every layer carries a copy of one k-means codebook).  part: (first, second) sizes of a comb /
    tcomb layer's two halves when they are not equal (multiples of 32; the modules then run two single-stream ops,
    lib/linear/comb_linear.py:91-102, 234-245)."""
    qi = get_quant_info(quantizer_str)
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    cgen = gen
    if codebook_seed is not None:
        cgen = torch.Generator(device=device)
        cgen.manual_seed(codebook_seed)

    def randn(shape):
        return torch.randn(shape, generator=cgen, device=device, dtype=torch.float32).to(torch.float16)

    if quantizer_str.startswith("ldlq") or quantizer_str.startswith("sq") or quantizer_str.startswith("vq"):
        bits, vec = qi["lut_bits"], qi["vec_sz"]
        return {"in_features": in_features, "out_features": out_features, "lut_bits": bits, "dtype": torch.float16,
                "vec_sz": vec, "qweight": _rand_i32((out_features, bits * in_features // 32 // vec), gen, device),
                "lut": randn((2 ** bits, vec)), "bias": None}
    common = {"in_features": in_features, "out_features": out_features, "td_x": 16, "td_y": 16, "L": 16, "V": 2,
              "tlut_bits": qi.get("tlut_bits"), "dtype": torch.float16, "bias": None}

    def trellis(rows, cols, kv):
        return _rand_i16(((rows // 16) * (cols // 16), math.ceil(256 * kv / 16 / 2)), gen, device)

    if quantizer_str.startswith("tcq"):
        return dict(common, KV=qi["KV"], trellis=trellis(out_features, in_features, qi["KV"]),
                    tlut=randn((2 ** qi["tlut_bits"], 2)))
    if quantizer_str.startswith("tcomb"):
        assert qi["ratio"] == 0.5 or part is not None, "only support ratio = 0.5 for now"
        part = tuple(part) if part is not None else (in_features // 2, in_features // 2)
        assert sum(part) == in_features
        return dict(common, KV=qi["KV"], in_part=part,
                    trellis1=trellis(out_features, part[0], qi["KV"][0]),
                    trellis2=trellis(out_features, part[1], qi["KV"][1]), tlut=randn((2 ** qi["tlut_bits"], 2)))
    if quantizer_str.startswith("comb"):
        assert qi["ratio"] == 0.5 or part is not None, "only support ratio = 0.5 for now"
        part = tuple(part) if part is not None else (out_features // 2, out_features // 2)
        assert sum(part) == out_features
        return dict(common, KV=qi["KV"], out_part=part,
                    trellis1=trellis(part[0], in_features, qi["KV"][0]),
                    trellis2=trellis(part[1], in_features, qi["KV"][1]), tlut=randn((2 ** qi["tlut_bits"], 2)))
    raise ValueError(f"Unknown quantizer: {quantizer_str}")


def get_dummy_quant_results(model_key, layer_key, quantizer_str, seed=0, device="cpu"):
    """Same dict schema as the reference's get_dummy_quant_results (mem_op.py:198-269)."""
    li = get_layer_info(model_key)[layer_key]
    info = {"quant_info": get_quant_info(quantizer_str), "in_features": li["in_features"],
            "out_features": li["out_features"], "dtype": torch.float16, "bias": None}
    if quantizer_str == "default":
        info["linear_info"] = {"in_features": li["in_features"], "out_features": li["out_features"],
                               "dtype": torch.float16, "bias": None}
    else:
        info["linear_info"] = dummy_linear_info(li["in_features"], li["out_features"], quantizer_str, seed, device)
    return info


def packed_bytes(linear_info):
    """Bytes of packed weights + codebook of one linear (the algorithmic HBM read per token at n = 1)."""
    total = 0
    for key in ("trellis", "trellis1", "trellis2", "qweight", "tlut", "lut"):
        t = linear_info.get(key)
        if t is not None:
            total += t.numel() * t.element_size()
    return total
