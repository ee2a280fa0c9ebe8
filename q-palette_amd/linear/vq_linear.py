"""VQ / non-uniform-SQ linears on the tensor-core-order packing and on the SIMT packing
(reference: lib/linear/vq_linear.py:5-97, 99-208)."""
import os

import torch

from .. import ops
from ._base import PackedLinearBase, merge_row_concat, op


def _default_device(device):
    if device is not None:
        return device
    return "cuda" if torch.cuda.is_available() else "cpu"


class _VQBase(PackedLinearBase):
    def __init__(self, in_features, out_features, lut_bits, vec_sz, bias=False, dtype=torch.half, device=None):
        super().__init__()
        dev = _default_device(device)
        self.in_features, self.out_features = in_features, out_features
        self.lut_bits, self.vec_sz, self.dtype = lut_bits, vec_sz, dtype
        # like the reference, buffers start with arbitrary (here: random) contents until loaded
        self.register_buffer("qweight", torch.randint(0, 4, (out_features, lut_bits * in_features // 32 // vec_sz),
                                                      dtype=torch.int32, device=dev))
        self.register_buffer("lut", torch.randn((2 ** lut_bits, vec_sz), dtype=dtype, device=dev))
        if bias:
            self.register_buffer("bias", torch.randn((out_features,), dtype=dtype, device=dev))
        else:
            self.bias = None

    def _info(self):
        return {
            "in_features": self.in_features, "out_features": self.out_features, "lut_bits": self.lut_bits,
            "dtype": self.dtype, "vec_sz": self.vec_sz, "qweight": self.qweight.detach().cpu(),
            "lut": self.lut.detach().cpu().half(),
            "bias": self.bias.detach().cpu() if self.bias is not None else None,
        }

    @staticmethod
    def merge_infos(info1, info2):
        return merge_row_concat(info1, info2, ["in_features", "lut_bits", "vec_sz", "dtype"], ["qweight"], "lut")


class VQLinearPackTensorCore(_VQBase):
    """Codes stored in mma-tile order (quant_op.py:101-162).  forward() returns the input's dtype.  `_gemv` (internal) returns fp32 from
    the tensor-core-order kernel and fp16 from the SIMT-order twin that few-row layers keep (fp32 accumulation in both; which of the
    two runs depends only on the layer's shape and QPAL_SIMT_TWIN, see _simt_twin)."""
    max_fused_batch = 128

    def __init__(self, in_features, out_features, lut_bits, vec_sz=2, bias=False, dtype=torch.half, device=None):
        super().__init__(in_features, out_features, lut_bits, vec_sz, bias, dtype, device)
        self.vq_type = f"vq{vec_sz}" if vec_sz > 1 else ("sq_dup" if lut_bits <= 4 else "sq")
        # beside a 128 KiB codebook image the LDS holds the x tiles of 64 batch rows, not of 128 (csrc/lut_gemm.hip)
        idx = lut_bits if vec_sz == 2 else (2 * lut_bits if lut_bits <= 6 else lut_bits)
        if (4 << (idx + min(15 - idx, 5))) > 64 * 1024:
            self.max_fused_batch = 64
        self.max_chunked_batch = 2 * self.max_fused_batch
        self.register_ops()

    def op_names(self):
        m, k = self.out_features, self.in_features
        return [f"decompress_gemm_{m}_{bs}_{k}_{self.lut_bits}_{self.vq_type}" for bs in range(1, self.max_fused_batch + 1)] + \
               [f"decompress_{self.lut_bits}_{self.vq_type}"]

    # Few-row layers (k / v / kv projections: <= 64 supertile rows) occupy 32-64 of the 256 compute units in the tensor-core-order
    # kernel, one workgroup per supertile row, and the wide codebooks make every step heavy: on MI355X the SIMT-order kernel runs
    # the same layer in half the time (perf/latency/3_8b_latency_coeffs_mi355x.json: k_ldlq_1_8 13.5 vs 4.8 us; with the
    # tensor-core kernel these were the 12 table entries an RTX 4090 beat).  For such layers the module keeps a SIMT-order copy of
    # its codes, re-packed on the device at the first decode call (what VQLinearPackSIMT.gen_layer_from_info does at load time,
    # lib/linear/vq_linear.py:175-188) — a few MB per layer — and its batch <= 8 forward runs the SIMT kernel (fp16 out, fp32
    # accumulate).  `qweight` stays the reference's buffer: state dicts, get_weight() and the fused multi-job launches are unchanged.
    SIMT_TWIN_MAX_ROWS = 2048

    def _twin_eligible(self):
        return (self.out_features <= self.SIMT_TWIN_MAX_ROWS and self.qweight.is_cuda and os.environ.get("QPAL_SIMT_TWIN", "1") != "0"
                and not (self.vec_sz == 2 and self.lut_bits < 3))

    def _twin_key(self):
        q = self.qweight
        return (q.data_ptr(), q._version, q.device)

    def prepare(self):
        """(Re)build the SIMT-order twin of `qweight` now.  Called wherever this package writes the codes (gen_layer_from_info,
        .to() / .cuda(), load_state_dict); call it yourself after writing `qweight` through a path autograd's version counter
        does not see (`qweight.data.copy_(...)`).  No-op for layers that keep no twin (many rows, CPU, QPAL_SIMT_TWIN=0)."""
        object.__setattr__(self, "_simt_qweight", None)  # (not a buffer: never part of the state dict)
        object.__setattr__(self, "_simt_key", None)
        if not self._twin_eligible():
            return None
        tw = ops.tc_to_simt(self.qweight, self.out_features, self.in_features, self.lut_bits, self.vec_sz)
        object.__setattr__(self, "_simt_qweight", tw)
        object.__setattr__(self, "_simt_key", self._twin_key())
        ops.register_names(["sq_pack_gemm_simt"] if self.vec_sz == 1 else
                           [f"vq_pack_gemm_simt_{bs}_{self.vec_sz}_{self.lut_bits}" for bs in range(1, 9)])
        return tw

    def _simt_twin(self):
        """The twin iff it matches the codes `qweight` holds NOW: the cache is keyed on (storage address, autograd version, device), so
        a load_state_dict() / in-place update / move after the first call re-packs instead of decoding stale codes.  While a stream
        capture or a trace is running a missing / stale twin is not rebuilt (the re-pack is a launch and an allocation): that call
        takes the tensor-core-order kernel, which reads `qweight` itself."""
        if not self._twin_eligible():
            return None
        tw = getattr(self, "_simt_qweight", None)
        if tw is not None and getattr(self, "_simt_key", None) == self._twin_key():
            return tw
        if torch.compiler.is_compiling() or torch.cuda.is_current_stream_capturing():
            return None
        return self.prepare()

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        object.__setattr__(self, "_simt_qweight", None)  # the buffers may have moved: re-pack at the next eager call
        object.__setattr__(self, "_simt_key", None)
        return out

    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)
        object.__setattr__(self, "_simt_qweight", None)
        object.__setattr__(self, "_simt_key", None)

    def _gemv(self, x, bs):
        m, k = self.out_features, self.in_features
        if bs <= 8:
            tw = self._simt_twin()
            if tw is not None:
                x3 = x.reshape(bs, 1, k)
                if x3.dtype != torch.float16:
                    x3 = x3.half()
                y = (op("sq_pack_gemm_simt")(x3, tw, self.lut, self.lut_bits) if self.vec_sz == 1
                     else op(f"vq_pack_gemm_simt_{bs}_{self.vec_sz}_{self.lut_bits}")(x3, tw, self.lut))
                return y.reshape(bs, m)
        return op(f"decompress_gemm_{m}_{bs}_{k}_{self.lut_bits}_{self.vq_type}")(self.qweight, x, self.lut)

    def get_weight(self):
        return op(f"decompress_{self.lut_bits}_{self.vq_type}")(self.qweight, self.lut, self.out_features,
                                                                 self.in_features)

    @staticmethod
    def gen_layer_from_info(info):
        layer = VQLinearPackTensorCore(info["in_features"], info["out_features"], info["lut_bits"], info["vec_sz"],
                                       info["bias"] is not None, info["dtype"], device=info["qweight"].device)
        layer.qweight.data.copy_(info["qweight"])
        layer.lut.data.copy_(info["lut"])
        if info["bias"] is not None:
            layer.bias.data.copy_(info["bias"])
        if layer.qweight.is_cuda:
            layer.prepare()
        return layer


class VQLinearPackSIMT(_VQBase):
    """Codes stored row-major in 32-lane blocks (pack_op.py:288-335); fp16 GEMV output."""

    def __init__(self, in_features, out_features, lut_bits, vec_sz=1, bias=False, dtype=torch.half, device=None):
        super().__init__(in_features, out_features, lut_bits, vec_sz, bias, dtype, device)
        self.register_ops()

    def op_names(self):
        if self.vec_sz == 1:
            return ["sq_pack_gemm_simt", "sq_pack_dequant_simt"]
        return [f"vq_pack_gemm_simt_{bs}_{self.vec_sz}_{self.lut_bits}" for bs in range(1, self.max_fused_batch + 1)] + \
               [f"vq_pack_dequant_simt_{self.vec_sz}_{self.lut_bits}"]

    def _gemv(self, x, bs):
        x3 = x.reshape(bs, 1, self.in_features)
        if self.vec_sz == 1:
            y = op("sq_pack_gemm_simt")(x3, self.qweight, self.lut, self.lut_bits)
        else:
            y = op(f"vq_pack_gemm_simt_{bs}_{self.vec_sz}_{self.lut_bits}")(x3, self.qweight, self.lut)
        return y.reshape(bs, self.out_features)

    def get_weight(self):
        m, k = self.out_features, self.in_features
        if self.vec_sz == 1:
            return op("sq_pack_dequant_simt")(self.qweight, self.lut, self.lut_bits, m, k)
        return op(f"vq_pack_dequant_simt_{self.vec_sz}_{self.lut_bits}")(self.qweight, self.lut, m, k)

    @staticmethod
    def gen_layer_from_info(info, device=None):
        """Layer files always hold the tensor-core packing for vec_sz <= 2; it is re-packed to the SIMT
        layout on the device at load time (reference: vq_linear.py:175-188 -> quant_op.py:246-257)."""
        dev = _default_device(device if device is not None else
                              (info["qweight"].device if info["qweight"].is_cuda else None))
        layer = VQLinearPackSIMT(info["in_features"], info["out_features"], info["lut_bits"], info["vec_sz"],
                                 info["bias"] is not None, info["dtype"], device=dev)
        if info["vec_sz"] <= 2:
            if not torch.device(dev).type == "cuda":
                raise RuntimeError("VQLinearPackSIMT needs the GPU to re-pack tensor-core-format codes (no CPU path)")
            converted = ops.tc_to_simt(info["qweight"].to(dev), info["out_features"], info["in_features"],
                                       info["lut_bits"], info["vec_sz"])
            layer.qweight.data.copy_(converted)
        else:
            layer.qweight.data.copy_(info["qweight"])
        layer.lut.data.copy_(info["lut"])
        if info["bias"] is not None:
            layer.bias.data.copy_(info["bias"])
        return layer
