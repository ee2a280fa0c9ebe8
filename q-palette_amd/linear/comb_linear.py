"""CombLinearTCQ (rows split over two bit-widths) and CombtLinearTCQ (columns split) — the x.25/x.75
bit "half-and-half" trellis linears (reference: lib/linear/comb_linear.py:5-145, 148-320)."""
import math

import torch
import torch.nn as nn

from ._base import PackedLinearBase, merge_row_concat, op


def _trellis_buffer(rows, cols, KV, td_x=16, td_y=16, V=2):
    return torch.zeros((rows // td_x) * (cols // td_y), math.ceil(td_x * td_y * KV / 16 / V), dtype=torch.int16)


class _CombBase(PackedLinearBase):
    max_fused_batch = 128  # batches 9..128: 1 / 2 / 4 / 8 groups of 16 batch rows per decoded step (csrc/tc_gemm16.h)
    max_chunked_batch = 256
    part_key = None

    def _common_init(self, in_features, out_features, td_x, td_y, part, L, KV, V, tlut_bits, bias, dtype):
        assert len(part) == 2 and len(KV) == 2
        assert td_x == 16 and td_y == 16 and L == 16 and V == 2, "kernel format is 16x16 tiles, L=16, V=2"
        self.in_features, self.out_features = in_features, out_features
        self.td_x, self.td_y, self.L, self.V = td_x, td_y, L, V
        self.KV = tuple(KV)
        self.tlut_bits, self.dtype = tlut_bits, dtype
        self.tlut = nn.Parameter(torch.zeros(2 ** tlut_bits, V, dtype=torch.float16), requires_grad=False)
        if bias:
            self.register_buffer("bias", torch.ones(out_features))
        else:
            self.bias = None
        self.use_comb_kernel = part[0] == part[1]

    def _info(self):
        return {
            "in_features": self.in_features, "out_features": self.out_features, "td_x": self.td_x, "td_y": self.td_y,
            self.part_key: getattr(self, self.part_key), "L": self.L, "KV": self.KV, "V": self.V,
            "tlut_bits": self.tlut_bits, "dtype": self.dtype,
            "trellis1": self.trellis1.detach().cpu(), "trellis2": self.trellis2.detach().cpu(),
            "tlut": self.tlut.detach().cpu().half(),
            "bias": self.bias.detach().cpu() if self.bias is not None else None,
        }

    @classmethod
    def gen_layer_from_info(cls, info):
        layer = cls(info["in_features"], info["out_features"], info["td_x"], info["td_y"], info[cls.part_key],
                    info["L"], info["KV"], info["V"], info["tlut_bits"], info["bias"] is not None, info["dtype"])
        layer.trellis1.data.copy_(info["trellis1"])
        layer.trellis2.data.copy_(info["trellis2"])
        layer.tlut.data.copy_(info["tlut"])
        if info["bias"] is not None:
            layer.bias.data.copy_(info["bias"])
        return layer


class CombLinearTCQ(_CombBase):
    """Rows [0, out_part[0]) at KV[0] bits, the rest at KV[1] = KV[0] + 1."""
    part_key = "out_part"

    def __init__(self, in_features, out_features, td_x, td_y, out_part, L, KV, V, tlut_bits, bias=False,
                 dtype=torch.float16):
        super().__init__()
        assert out_part[0] + out_part[1] == out_features
        self._common_init(in_features, out_features, td_x, td_y, out_part, L, KV, V, tlut_bits, bias, dtype)
        self.out_part = tuple(out_part)
        self.register_buffer("trellis1", _trellis_buffer(out_part[0], in_features, KV[0]))
        self.register_buffer("trellis2", _trellis_buffer(out_part[1], in_features, KV[1]))
        self.register_ops()

    def op_names(self):
        k, S, (kv1, kv2), B = self.in_features, self.tlut_bits, self.KV, range(1, self.max_fused_batch + 1)
        if self.use_comb_kernel:
            return [f"decompress_gemm_tcq_comb_{self.out_features}_{bs}_{k}_{S}_{kv1}_{kv2}" for bs in B] + \
                   [f"decompress_tcq_comb_{S}_{kv1}_{kv2}"]
        return [f"decompress_gemm_tcq_{self.out_part[i]}_{bs}_{k}_{S}_{kv}" for i, kv in enumerate((kv1, kv2))
                if self.out_part[i] > 0 for bs in B] + [f"decompress_tcq_{S}_{kv1}", f"decompress_tcq_{S}_{kv2}"]

    def _gemv(self, x, bs):
        k, S, (kv1, kv2) = self.in_features, self.tlut_bits, self.KV
        if self.use_comb_kernel:
            name = f"decompress_gemm_tcq_comb_{self.out_features}_{bs}_{k}_{S}_{kv1}_{kv2}"
            return op(name)(self.trellis1, self.trellis2, x, self.tlut)
        ys = [op(f"decompress_gemm_tcq_{mp}_{bs}_{k}_{S}_{kv}")(t, x, self.tlut)
              for mp, kv, t in ((self.out_part[0], kv1, self.trellis1), (self.out_part[1], kv2, self.trellis2)) if mp > 0]
        return ys[0] if len(ys) == 1 else torch.cat(ys, dim=1)  # a row shard may hold rows of one half only

    def get_weight(self):
        k, S, (kv1, kv2) = self.in_features, self.tlut_bits, self.KV
        if self.use_comb_kernel:
            return op(f"decompress_tcq_comb_{S}_{kv1}_{kv2}")(self.trellis1, self.trellis2, self.tlut,
                                                               self.out_features, k)
        ws = [op(f"decompress_tcq_{S}_{kv}")(t, self.tlut, mp, k)
              for mp, kv, t in ((self.out_part[0], kv1, self.trellis1), (self.out_part[1], kv2, self.trellis2)) if mp > 0]
        return ws[0] if len(ws) == 1 else torch.cat(ws, dim=0)


class CombtLinearTCQ(_CombBase):
    """Columns [0, in_part[0]) at KV[0] bits, the rest at KV[1] = KV[0] + 1 (quantizer strings tcomb_*)."""
    part_key = "in_part"

    def __init__(self, in_features, out_features, td_x, td_y, in_part, L, KV, V, tlut_bits, bias=False,
                 dtype=torch.float16):
        super().__init__()
        assert in_part[0] + in_part[1] == in_features
        self._common_init(in_features, out_features, td_x, td_y, in_part, L, KV, V, tlut_bits, bias, dtype)
        self.in_part = tuple(in_part)
        self.register_buffer("trellis1", _trellis_buffer(out_features, in_part[0], KV[0]))
        self.register_buffer("trellis2", _trellis_buffer(out_features, in_part[1], KV[1]))
        self.register_ops()

    def op_names(self):
        m, k, S, (kv1, kv2), B = self.out_features, self.in_features, self.tlut_bits, self.KV, range(1, self.max_fused_batch + 1)
        if self.use_comb_kernel:
            return [f"decompress_gemm_tcq_combt_{m}_{bs}_{k}_{S}_{kv1}_{kv2}" for bs in B] + [f"decompress_tcq_combt_{S}_{kv1}_{kv2}"]
        return [f"decompress_gemm_tcq_{m}_{bs}_{self.in_part[i]}_{S}_{kv}" for i, kv in enumerate((kv1, kv2)) for bs in B] + \
               [f"decompress_tcq_{S}_{kv1}", f"decompress_tcq_{S}_{kv2}"]

    def _gemv(self, x, bs):
        m, k, S, (kv1, kv2) = self.out_features, self.in_features, self.tlut_bits, self.KV
        if self.use_comb_kernel:
            return op(f"decompress_gemm_tcq_combt_{m}_{bs}_{k}_{S}_{kv1}_{kv2}")(self.trellis1, self.trellis2, x,
                                                                                   self.tlut)
        k1, k2 = self.in_part
        y1 = op(f"decompress_gemm_tcq_{m}_{bs}_{k1}_{S}_{kv1}")(self.trellis1, x[:, :k1], self.tlut)
        y2 = op(f"decompress_gemm_tcq_{m}_{bs}_{k2}_{S}_{kv2}")(self.trellis2, x[:, k1:], self.tlut)
        return y1 + y2

    def get_weight(self):
        m, k, S, (kv1, kv2) = self.out_features, self.in_features, self.tlut_bits, self.KV
        if self.use_comb_kernel:
            return op(f"decompress_tcq_combt_{S}_{kv1}_{kv2}")(self.trellis1, self.trellis2, self.tlut, m, k)
        w1 = op(f"decompress_tcq_{S}_{kv1}")(self.trellis1, self.tlut, m, self.in_part[0])
        w2 = op(f"decompress_tcq_{S}_{kv2}")(self.trellis2, self.tlut, m, self.in_part[1])
        return torch.cat([w1, w2], dim=1)

    @staticmethod
    def merge_infos(info1, info2):
        assert tuple(info1["KV"]) == tuple(info2["KV"]) and tuple(info1["in_part"]) == tuple(info2["in_part"])
        a = dict(info1, KV=tuple(info1["KV"]), in_part=tuple(info1["in_part"]))
        b = dict(info2, KV=tuple(info2["KV"]), in_part=tuple(info2["in_part"]))
        return merge_row_concat(a, b, ["in_features", "td_x", "td_y", "L", "KV", "V", "tlut_bits", "dtype", "in_part"],
                                ["trellis1", "trellis2"], "tlut")
