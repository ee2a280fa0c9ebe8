"""nn.Module mirror of the reference's quantized linears (lib/linear/{tcq,comb,vq}_linear.py)."""
import os

import torch

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


def _codec_key(layer):
    """Layers with equal keys run the same kernel instantiation and can share one multi-job launch."""
    if isinstance(layer, QTIPLinearTCQ):
        return ("tcq", layer.in_features, layer.tlut_bits, layer.KV)
    if isinstance(layer, CombtLinearTCQ) and layer.use_comb_kernel:
        return ("tcombt", layer.in_features, layer.tlut_bits, tuple(layer.KV))
    if isinstance(layer, VQLinearPackTensorCore):
        return ("lut_tc", layer.in_features, layer.lut_bits, layer.vec_sz)
    return ("single", id(layer))


def _launch_key(layer, mixed_kv):
    """Key of the launch a layer can join.  mixed_kv: TCQ layers of one codebook size — single-stream and column-split
    (tcomb) ones — share a launch whatever their KV (any-KV kernel: KV 2..8 at S = 9, 8..10 at S = 10, 9..10 at S = 11;
    fused batch <= 8)."""
    key = _codec_key(layer)
    if mixed_kv and key[0] in ("tcq", "tcombt"):
        s = layer.tlut_bits
        kvs = [layer.KV] if key[0] == "tcq" else list(layer.KV)
        if all((s == 9 and kv <= 8) or (s == 10 and kv >= 8) or (s == 11 and kv >= 9) for kv in kvs):
            return ("tcq", key[1], s, "any")
    return key


def launch_groups(layers, mixed_kv=False):
    """Partition `layers` (projections of one input) into multi-job launches: lists of indices, at most 8 each.
    mixed_kv: TCQ layers of one codebook size share a launch whatever their KV (any-KV kernel)."""
    groups = {}
    for i, layer in enumerate(layers):
        groups.setdefault(_launch_key(layer, mixed_kv), []).append(i)
    out = []
    for key, idxs in groups.items():
        if key[0] == "single":
            out.append(idxs)
        else:
            out += [idxs[a:a + 8] for a in range(0, len(idxs), 8)]
    return out


_PACKED_KEYS = {QTIPLinearTCQ: ("trellis",), CombtLinearTCQ: ("trellis1", "trellis2"), VQLinearPackTensorCore: ("qweight",)}


def interleave_up_gate(up, gate):
    """One layer whose supertile rows (32 output rows) alternate up, gate, up, gate, ...: what the GEMV's SwiGLU epilogue needs
    (multi_gemv(..., act_out=...): both halves of `act_fn(gate) * up` then live in the same workgroup).  up / gate: two
    tensor-core-order layers of one codec and shape (or a merged up|gate layer as `up`, gate=None: rows [up; gate]).  A
    supertile row is contiguous in every packed buffer, so this is a row shuffle of the packed data.  Returns the new layer;
    use interleave_rows() on per-row vectors (Wscale)."""
    if gate is None:
        info = up._info()
        m = info["out_features"]
        keys = _PACKED_KEYS[type(up)]
        halves = {key: info[key].reshape(m // 32, -1) for key in keys}
        for key in keys:
            t = halves[key]
            info[key] = torch.stack([t[: m // 64], t[m // 64:]], dim=1).reshape(info[key].shape).contiguous()
        return type(up).gen_layer_from_info(info).to(next(up.buffers()).device)
    assert type(up) is type(gate) and _codec_key(up) == _codec_key(gate) and up.out_features == gate.out_features
    iu, ig = up._info(), gate._info()
    m = iu["out_features"]
    info = dict(iu)
    info["out_features"] = 2 * m
    for key in _PACKED_KEYS[type(up)]:
        a, b = iu[key].reshape(m // 32, -1), ig[key].reshape(m // 32, -1)
        info[key] = torch.stack([a, b], dim=1).reshape(2 * iu[key].shape[0], *iu[key].shape[1:]).contiguous()
    return type(up).gen_layer_from_info(info).to(next(up.buffers()).device)


def interleave_rows(up_vec, gate_vec):
    """Per-output-row vectors (Wscale) of an up / gate pair in the row order of interleave_up_gate."""
    m = up_vec.numel()
    return torch.stack([up_vec.reshape(m // 32, 32), gate_vec.reshape(m // 32, 32)], dim=1).reshape(-1).contiguous()


def rotation_fusable(layers, n):
    """True if multi_gemv(..., x_rot=...) can apply the left rotation inside the GEMV launches of `layers`."""
    from .. import ops
    return (all(_codec_key(l)[0] != "single" for l in layers)
            and ops.can_fuse_rotation(n, layers[0].in_features))


def multi_gemv(layers, x, outs=None, outs_zeroed=False, prezero=None, wscales=None, oscale=1.0, x_rot=None, x_rms=None,
               accumulate=False, act_out=None, act_su=None):
    """y_i = layers[i](x) for several quantized linears that share the input, batch <= 8.  Layers of one
    kind and codec (e.g. q|k|v or gate|up of one block) go out as ONE kernel launch per codec
    (C-ABI qpal_*_gemv_multi); anything else is one launch per layer.  Returns fp32/fp16 [n, m_i] tensors in
    the order of `layers`.

    outs: fp32 [n, m_i] tensors to write into (outs_zeroed: they hold zeros, so a split-K layer needs no
    memset of its own); prezero: a tensor the first multi-job launch also zeroes — the way a decode block
    prepares down_proj's output during the gate|up launch.
    wscales (fp16 [m_i] vectors) / oscale: y_i = layers[i](x) * wscales[i] * oscale, fused into the GEMV epilogue for
    the tensor-core-order families (the `* Wscale * scale` of the incoherent wrappers).
    x_rot = (su, post_scale): x is the UN-rotated input and every launch stages fp16(fp16(H (x*su)/sqrt(k)) * post)
    itself (only where rotation_fusable(layers, n)); x may then be the fp32 residual stream, and x_rms = (eps, weight or
    None) applies the RMSNorm in front of the rotation (decoder-block fusion).  accumulate: outs[i] += y_i (the residual add:
    outs[i] holds the residual stream; the launch may split K on its own: the atomics add onto the live output, outs_zeroed is ignored).
    act_out (fp16 [1, m / 2]; one layer built by interleave_up_gate, batch 1, x_rot): the launch's epilogue writes
    silu(gate) * up there and no fp32 output (returns [None]); act_su (fp16 [m / 2] of +-1): multiplied into it — the sign flip in
    front of the NEXT projection's rotation, which then reads one vector instead of two.
    x_rot = (su, post, hadK, K) with K = 28 on a k = 14336 layer: the (hadK (x) H_512) rotation of a down_proj input inside the
    launch's x staging (ops.can_fuse_rotation(n, k, K))."""
    from .. import ops

    x2 = x.reshape(-1, layers[0].in_features)
    n = x2.shape[0]
    if act_out is not None:
        if len(layers) != 1 or n != 1 or x_rot is None or accumulate or outs is not None:
            raise RuntimeError("multi_gemv: act_out needs ONE interleaved up|gate layer, batch 1 and x_rot")
        layer = layers[0]
        common = dict(prezero=prezero, wscales=wscales, oscale=oscale, x_rot=x_rot, x_rms=x_rms, act_outs=[act_out], act_su=act_su)
        if isinstance(layer, QTIPLinearTCQ):
            ops.tcq_gemv_multi([(layer.trellis, None, layer.tlut, layer.out_features)], x2, layer.tlut_bits, layer.KV, **common)
        elif isinstance(layer, CombtLinearTCQ) and layer.use_comb_kernel:
            ops.tcq_gemv_multi([(layer.trellis1, layer.trellis2, layer.tlut, layer.out_features)], x2, layer.tlut_bits,
                               layer.KV[0], layer.KV[1], split=2, **common)
        elif isinstance(layer, VQLinearPackTensorCore):
            ops.lut_tc_gemv_multi([(layer.qweight, layer.lut, layer.out_features)], x2, layer.lut_bits, layer.vec_sz, **common)
        else:
            raise RuntimeError("multi_gemv: act_out needs a tensor-core-order packed layer")
        return [None]
    fused = min(l.max_fused_batch for l in layers)
    if n > fused:
        if outs is not None or prezero is not None or wscales is not None or oscale != 1.0 or x_rot is not None or accumulate:
            raise RuntimeError("multi_gemv: outs / prezero / wscales / oscale / x_rot need a fused batch (n <= 128)")
        if n <= min(l.max_chunked_batch for l in layers):
            # passes of the fused launches over slices of the batch (each pass keeps the multi-job grouping): faster than
            # decode-to-HBM + GEMM up to ~2 passes
            # — every pass writes its rows of ONE [n, m] output per layer (no concatenation kernels)
            if any(_codec_key(l)[0] == "single" for l in layers):
                # row-split (comb_*) layers and column-split ones with unequal parts have no multi-job form: passes of their
                # own fused launches (NOT l(x2): the module's forward sends this batch range back here)
                return [torch.cat([l._gemv(x2[i:i + fused], min(fused, n - i)).float() for i in range(0, n, fused)], dim=0)
                        for l in layers]
            full = [torch.empty((n, l.out_features), dtype=torch.float32, device=x2.device) for l in layers]
            for i in range(0, n, fused):
                multi_gemv(layers, x2[i:i + fused], outs=[f[i:i + fused] for f in full])
            return full
        return [l(x2) for l in layers]  # decode-to-fp16 + GEMM path of the modules
    results = [None] * len(layers)
    mixed_kv = x_rot is None and n <= 8 and not accumulate  # the any-KV kernels: no rotation, batch <= 8
    for idxs in launch_groups(layers, mixed_kv):
        first = layers[idxs[0]]
        kind = _codec_key(first)[0]
        grp = [layers[i] for i in idxs]
        o = [outs[i] for i in idxs] if outs is not None else None
        ws = [wscales[i] for i in idxs] if wscales is not None else None
        extra = dict(outs=o, outs_zeroed=outs_zeroed, prezero=prezero, wscales=ws, oscale=oscale, x_rot=x_rot, x_rms=x_rms,
                     accumulate=accumulate)
        if kind == "tcq" or (kind == "tcombt" and any(_codec_key(l) != _codec_key(first) for l in grp)):
            # one codec: its own kernel; different KV / single- and two-stream layers mixed: the any-KV kernel (per-job KV)
            jobs = [(l.trellis, None, l.tlut, l.out_features, l.KV) if isinstance(l, QTIPLinearTCQ)
                    else (l.trellis1, l.trellis2, l.tlut, l.out_features, l.KV[0], l.KV[1]) for l in grp]
            kv1 = first.KV if isinstance(first, QTIPLinearTCQ) else first.KV[0]
            ys = ops.tcq_gemv_multi(jobs, x2, first.tlut_bits, kv1, **extra)
            prezero = None
        elif kind == "tcombt":
            ys = ops.tcq_gemv_multi([(l.trellis1, l.trellis2, l.tlut, l.out_features) for l in grp], x2,
                                    first.tlut_bits, first.KV[0], first.KV[1], split=2, **extra)
            prezero = None
        elif kind == "lut_tc":
            ys = ops.lut_tc_gemv_multi([(l.qweight, l.lut, l.out_features) for l in grp], x2, first.lut_bits,
                                       first.vec_sz, **extra)
            prezero = None
        else:
            if x_rot is not None or accumulate:
                raise RuntimeError("x_rot / accumulate need tensor-core-order packed layers (see rotation_fusable)")
            ys = [first._gemv(x2, n) if n <= first.max_fused_batch else first(x2)]
            if ws is not None and ws[0] is not None:
                ys = [ys[0].float() * ws[0].float() * oscale]
            elif oscale != 1.0:
                ys = [ys[0].float() * oscale]
            if o is not None:
                o[0].copy_(ys[0])
                ys = o
        for i, y in zip(idxs, ys):
            results[i] = y
    if prezero is not None:  # no multi-job launch took it along
        prezero.zero_()
    return results


from .incoherent_linear import IncoherentLinear, IncoherentMLP, IncoherentSdpaAttention, make_linear  # noqa: E402

__all__ = ["rotation_fusable", "IncoherentLinear", "IncoherentMLP", "IncoherentSdpaAttention", "make_linear", "multi_gemv", "launch_groups", "share_codebooks","QTIPLinearTCQ", "CombLinearTCQ", "CombtLinearTCQ", "VQLinearPackTensorCore", "VQLinearPackSIMT",
           "linear_class_for", "make_linear_from_info"]
