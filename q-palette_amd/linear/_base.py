"""Shared plumbing of the quantized linears: batch dispatch (fused decode + GEMV / skinny GEMM up to ``max_fused_batch``,
decode-to-fp16 + fp16 GEMM above) and the row-concatenating ``merge_infos`` used for QKV / up+gate layer fusion."""
import torch
import torch.nn as nn

from .. import ops

GEMV_MAX_BATCH = 8  # reference: `if bs <= 8` in every forward (e.g. lib/linear/tcq_linear.py:68); SIMT packings keep it


class PackedLinearBase(nn.Module):
    in_features: int
    out_features: int
    max_fused_batch = GEMV_MAX_BATCH  # tensor-core-order families raise this to 128 (8 groups of 16 batch rows; 64 under a 128 KiB image)
    # up to this batch the forward runs ceil(bs / max_fused_batch) passes of the fused kernel instead of decode-to-HBM + GEMM:
    # measured on Llama-8B shapes the passes win up to ~2 x 64 rows; 0: no chunked passes (SIMT packings)
    max_chunked_batch = 0

    def _gemv(self, x, bs):  # -> [bs, m] (fp32 or fp16)
        raise NotImplementedError

    def get_weight(self):  # -> fp16 [m, k]
        raise NotImplementedError

    def op_names(self):
        """Every ``torch.ops.ours_lib`` name this module's forward can request."""
        raise NotImplementedError

    def register_ops(self):
        """Called at the end of every constructor: the operators exist before the first forward / trace / capture
        (the reference registers its ~12k names eagerly at import, lib/linear/__init__.py:42-420)."""
        ops.register_names(self.op_names())

    def forward(self, inp, **kwargs):
        x = inp.reshape(-1, self.in_features)
        bs = x.shape[0]
        if bs <= self.max_fused_batch:
            y = self._gemv(x, bs)
        elif bs <= self.max_chunked_batch:
            from . import multi_gemv  # passes of the fused launch, each writing its rows of one [bs, m] output
            y = multi_gemv([self], x)[0]
        else:
            with torch.no_grad():
                dq = self.get_weight()
            y = x.to(dq.dtype) @ dq.T
        # note: like the reference, `bias` is carried but never added
        return y.reshape(*inp.shape[:-1], self.out_features).to(inp.dtype)


def _check_mergeable(a, b, keys):
    for key in keys:
        if a[key] != b[key]:
            raise AssertionError(f"cannot merge layers: {key} differs ({a[key]} vs {b[key]})")
    if a["bias"] is not None or b["bias"] is not None:
        raise AssertionError("cannot merge layers with bias")


def merge_row_concat(a, b, same_keys, cat_keys, table_key):
    """Fuse two layers that share the input (q|k|v, up|gate): packed rows are independent, so the
    fused layer is the row-wise concatenation of the packed buffers."""
    _check_mergeable(a, b, same_keys)
    if not torch.allclose(a[table_key].float(), b[table_key].float(), atol=1e-4):
        print(f"warning: {table_key} is not close. it is unexpected behavior if you do not use dummy quantizers.")
    out = {key: a[key] for key in same_keys}
    out["out_features"] = a["out_features"] + b["out_features"]
    out["bias"] = None
    for key in cat_keys:
        out[key] = torch.cat([a[key], b[key]], dim=0)
    out[table_key] = a[table_key]
    return out


def op(name):
    return ops.get_op(name)
