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


__all__ = ["QTIPLinearTCQ", "CombLinearTCQ", "CombtLinearTCQ", "VQLinearPackTensorCore", "VQLinearPackSIMT",
           "linear_class_for", "make_linear_from_info"]
