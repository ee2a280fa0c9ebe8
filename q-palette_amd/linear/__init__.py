"""nn.Module mirror of the reference's quantized linears (lib/linear/{tcq,comb,vq}_linear.py)."""
from .comb_linear import CombLinearTCQ, CombtLinearTCQ
from .tcq_linear import QTIPLinearTCQ
from .vq_linear import VQLinearPackSIMT, VQLinearPackTensorCore


def linear_class_for(quantizer_str, use_simt=False):
    """Quantizer string -> module class; the substring order matters
    (reference: lib/linear/incoherent_linear.py:13-26)."""
    if "tcomb" in quantizer_str:
        return CombtLinearTCQ
    if "comb" in quantizer_str:
        return CombLinearTCQ
    if "tcq" in quantizer_str:
        return QTIPLinearTCQ
    if "sq" in quantizer_str or "vq" in quantizer_str or "ldlq" in quantizer_str:
        return VQLinearPackSIMT if use_simt else VQLinearPackTensorCore
    raise ValueError(f"no quantized linear for quantizer string {quantizer_str!r}")


def make_linear_from_info(quantizer_str, linear_info, use_simt=False):
    return linear_class_for(quantizer_str, use_simt).gen_layer_from_info(linear_info)


def share_codebooks(modules):
    """Make linears whose codebooks are bit-identical share ONE device tensor.

    Real Q-Palette checkpoints store a copy of the same k-means codebook in every layer (tcq_quant.py:131:
    ``tcq_linear.tlut.data.copy_(cb.tlut)``; assets/lut_cache/*.pt).  Sharing the storage lets a multi-job
    launch see equal pointers and keep the LDS codebook image across jobs instead of rebuilding it.
    Returns the number of distinct codebooks left."""
    import torch

    seen = []
    for mod in modules:
        name = "tlut" if hasattr(mod, "tlut") else ("lut" if hasattr(mod, "lut") else None)
        if name is None:
            continue
        t = getattr(mod, name)
        for ref in seen:
            if ref.shape == t.shape and ref.dtype == t.dtype and ref.device == t.device and torch.equal(ref, t):
                if isinstance(t, torch.nn.Parameter):
                    t.data = ref.data
                else:
                    setattr(mod, name, ref)
                break
        else:
            seen.append(t)
    return len(seen)


def multi_gemv(layers, x):
    """y_i = layers[i](x) for several quantized linears that share the input, batch <= 8.  Layers of one
    kind and codec (e.g. q|k|v or gate|up of one block under a single-scheme quantizer) go out as ONE
    kernel launch; anything else falls back to one launch per layer.  Returns fp32 [n, m_i] tensors."""
    from .. import ops

    x2 = x.reshape(-1, layers[0].in_features)
    n = x2.shape[0]
    first = layers[0]
    same_kind = all(type(l) is type(first) and l.in_features == first.in_features for l in layers)
    if n <= 8 and same_kind and len(layers) <= 8:
        if isinstance(first, QTIPLinearTCQ) and all((l.KV, l.tlut_bits) == (first.KV, first.tlut_bits) for l in layers):
            return ops.tcq_gemv_multi([(l.trellis, None, l.tlut, l.out_features) for l in layers], x2,
                                      first.tlut_bits, first.KV)
        if (isinstance(first, CombtLinearTCQ) and all(l.use_comb_kernel and (l.KV, l.tlut_bits) ==
                                                      (first.KV, first.tlut_bits) for l in layers)):
            return ops.tcq_gemv_multi([(l.trellis1, l.trellis2, l.tlut, l.out_features) for l in layers], x2,
                                      first.tlut_bits, first.KV[0], first.KV[1], split=2)
        if (isinstance(first, VQLinearPackTensorCore) and
                all((l.lut_bits, l.vec_sz) == (first.lut_bits, first.vec_sz) for l in layers)):
            return ops.lut_tc_gemv_multi([(l.qweight, l.lut, l.out_features) for l in layers], x2, first.lut_bits,
                                         first.vec_sz)
    return [l._gemv(x2, n) if n <= 8 else l(x2) for l in layers]


__all__ = ["multi_gemv", "share_codebooks","QTIPLinearTCQ", "CombLinearTCQ", "CombtLinearTCQ", "VQLinearPackTensorCore", "VQLinearPackSIMT",
           "linear_class_for", "make_linear_from_info"]
