"""Packers of the quantised weight formats (host side, CPU tensors) on the C-ABI's ``qpal_pack_*``.

Mirror of the reference's packing entry points, same argument meaning and output shapes/dtypes:
  pack_trellis(Qidxs, m, k, KV)             cb.pack_trellis + nibble permutation, lib/quantizer/tcq_quant.py:47-60
                                            (comb_quant.pack_trellis, lib/quantizer/comb_quant.py)
  pack_qweight(P, vec_sz, lut_bits)         lib/quantizer/quant_op.py:89-162 (tensor-core order)
  pack_qweight_sq_simt(P, lut_bits)         lib/quantizer/quant_op.py:80-87  (numba: pack_op.py:288-335)
  pack_qweight_vq_simt(P, lut_bits, vec_sz) lib/quantizer/quant_op.py:69-78
One pass of plain C++ per layer instead of GPU bit-tensor reshapes plus numba loops; nothing here touches a GPU.
"""
import torch

from . import _native


def _idx(t, name):
    if t.is_cuda:
        t = t.cpu()
    if t.dim() == 3:          # one-hot / score form (N, K // vec, 2 ** bits): the reference takes argmax
        t = t.argmax(dim=-1)
    if t.dim() != 2:
        raise _native.QpalError(f"{name} must be a 2-D index matrix")
    return t.to(torch.int32).contiguous()


def pack_trellis(Qidxs, m, k, KV):
    """Qidxs: (m, k // 2) trellis states of a tail-biting walk per 16x16 tile -> int16 [(m/16)(k/16), 8 KV]."""
    q = _idx(Qidxs, "Qidxs")
    if tuple(q.shape) != (m, k // 2):
        raise _native.QpalError(f"Qidxs must have shape ({m}, {k // 2})")
    out = torch.zeros((m // 16) * (k // 16), 8 * KV, dtype=torch.int16)
    _native.check(_native.lib().qpal_pack_tcq(out.data_ptr(), q.data_ptr(), m, k, KV), "qpal_pack_tcq")
    return out


def pack_qweight(P, vec_sz, lut_bits):
    """P: (N, K // vec_sz) codebook indices (or (N, K // vec_sz, 2 ** lut_bits) scores) -> int32 (N, lut_bits K / 32 / vec_sz)."""
    p = _idx(P, "P")
    n, k = p.shape[0], p.shape[1] * vec_sz
    out = torch.zeros(n, lut_bits * k // 32 // vec_sz, dtype=torch.int32)
    _native.check(_native.lib().qpal_pack_lut_tc(out.data_ptr(), p.data_ptr(), n, k, lut_bits, vec_sz), "qpal_pack_lut_tc")
    return out


def _pack_simt(P, lut_bits, vec_sz):
    p = _idx(P, "P")
    n, k = p.shape[0], p.shape[1] * vec_sz
    out = torch.zeros(n, lut_bits * k // 32 // vec_sz, dtype=torch.int32)
    _native.check(_native.lib().qpal_pack_lut_simt(out.data_ptr(), p.data_ptr(), n, k, lut_bits, vec_sz),
                  "qpal_pack_lut_simt")
    return out


def pack_qweight_sq_simt(P, lut_bits):
    return _pack_simt(P, lut_bits, 1)


def pack_qweight_vq_simt(P, lut_bits, vec_sz, code_n=None, codeT_sz=32):
    return _pack_simt(P, lut_bits, vec_sz)
