"""QTIPLinearTCQ — trellis-coded linear (reference: lib/linear/tcq_linear.py:5-122)."""
import math

import torch
import torch.nn as nn

from ._base import PackedLinearBase, merge_row_concat, op


class QTIPLinearTCQ(PackedLinearBase):
    max_fused_batch = 128  # batches 9..128: 1 / 2 / 4 / 8 groups of 16 batch rows per decoded step (csrc/tc_gemm16.h)
    max_chunked_batch = 256
    def __init__(self, in_features, out_features, td_x, td_y, L, KV, V, tlut_bits, bias=False, dtype=torch.float16):
        super().__init__()
        assert td_x == 16 and td_y == 16 and L == 16 and V == 2, "kernel format is 16x16 tiles, L=16, V=2"
        self.in_features, self.out_features = in_features, out_features
        self.td_x, self.td_y, self.L, self.KV, self.V = td_x, td_y, L, KV, V
        self.tlut_bits, self.dtype = tlut_bits, dtype
        ntiles = (out_features // td_x) * (in_features // td_y)
        self.register_buffer("trellis", torch.zeros(ntiles, math.ceil(td_x * td_y * KV / 16 / V), dtype=torch.int16))
        self.tlut = nn.Parameter(torch.zeros(2 ** tlut_bits, V, dtype=torch.float16), requires_grad=False)
        if bias:
            self.register_buffer("bias", torch.ones(out_features))
        else:
            self.bias = None
        self.register_ops()

    def op_names(self):
        m, k, S, KV = self.out_features, self.in_features, self.tlut_bits, self.KV
        return [f"decompress_gemm_tcq_{m}_{bs}_{k}_{S}_{KV}" for bs in range(1, self.max_fused_batch + 1)] + \
               [f"decompress_tcq_{S}_{KV}"]

    def _info(self):
        return {
            "in_features": self.in_features, "out_features": self.out_features, "td_x": self.td_x, "td_y": self.td_y,
            "L": self.L, "KV": self.KV, "V": self.V, "tlut_bits": self.tlut_bits, "dtype": self.dtype,
            "trellis": self.trellis.detach().cpu(), "tlut": self.tlut.detach().cpu().half(),
            "bias": self.bias.detach().cpu() if self.bias is not None else None,
        }

    def _gemv(self, x, bs):
        m, k = self.out_features, self.in_features
        return op(f"decompress_gemm_tcq_{m}_{bs}_{k}_{self.tlut_bits}_{self.KV}")(self.trellis, x, self.tlut)

    def get_weight(self):
        return op(f"decompress_tcq_{self.tlut_bits}_{self.KV}")(self.trellis, self.tlut, self.out_features,
                                                                 self.in_features)

    @staticmethod
    def gen_layer_from_info(info):
        layer = QTIPLinearTCQ(info["in_features"], info["out_features"], info["td_x"], info["td_y"], info["L"],
                              info["KV"], info["V"], info["tlut_bits"], info["bias"] is not None, info["dtype"])
        layer.trellis.data.copy_(info["trellis"])
        layer.tlut.data.copy_(info["tlut"])
        if info["bias"] is not None:
            layer.bias.data.copy_(info["bias"])
        return layer

    @staticmethod
    def merge_infos(info1, info2):
        return merge_row_concat(info1, info2,
                                ["in_features", "td_x", "td_y", "L", "KV", "V", "tlut_bits", "dtype"],
                                ["trellis"], "tlut")
