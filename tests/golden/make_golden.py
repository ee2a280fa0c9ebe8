#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference's own Python.

Runs only in the builder container (needs /root/reference; CPU only).  Nothing of the reference is
copied: the reference functions are called as they are (``torch.Tensor.cuda`` is patched to the
identity so their hard ``.cuda()`` calls stay on the CPU; absent third-party imports are stubbed by
empty modules).  What is written to disk is data only: packed inputs and the reference's outputs.

    cd /root/repo && TORCHDYNAMO_DISABLE=1 python tests/golden/make_golden.py

Reference entry points used (paths relative to /root/reference):
  lib/codebook/bitshift.py      bitshift_codebook(...).recons / .pack_trellis / quantlut_sym
  lib/quantizer/comb_quant.py   pack_trellis (pack + nibble permutation, == tcq_quant.py:47-60)
  lib/utils/kernel_decompress.py decode_compressed (even KV only)
  lib/algo/ldlq.py              _INV_PERMUTE
  lib/quantizer/quant_op.py     pack_qweight, dequantize_mat_sq_inds(_vec2), pack_qweight_vq_simt,
                                pack_qweight_sq_simt, convert_tensor_core_to_simt
  lib/utils/matmul_had.py       get_hadK, matmul_hadU, matmul_hadUt (the pure-torch transform; the *_cuda variants
                                need the absent third-party fast_hadamard_transform)

    python tests/golden/make_golden.py hadamard      # only (re)generate hadamard.npz
"""
import os
import sys
import types

os.environ.setdefault("TORCHDYNAMO_DISABLE", "1")
REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))

import numpy as np
import torch

# ---- stubs for imports the image lacks; none of them is on the code paths exercised here
for name in ["glog", "fast_hadamard_transform", "flash1dkmeans", "vq_tensor_kernels", "tcq_kernels",
             "sq_pack_gemm", "vq_pack_gemm"]:
    sys.modules[name] = types.ModuleType(name)
sys.modules["glog"].info = print
_nb = types.ModuleType("numba")
_nb.njit = lambda *a, **k: (a[0] if (a and callable(a[0])) else (lambda f: f))
sys.modules["numba"] = _nb
# lib.linear registers ~12k torch ops at import and needs the CUDA extensions: give the quantizer
# modules an empty stand-in for the names they import from it (never used here).
_ll = types.ModuleType("lib.linear")
for n in ["QTIPLinearTCQ", "IncoherentLinear", "CombLinearTCQ", "CombtLinearTCQ", "VQLinearPackSIMT",
          "VQLinearPackTensorCore", "QuantizedLinear"]:
    setattr(_ll, n, None)
torch.Tensor.cuda = lambda self, *a, **k: self  # keep the reference's .cuda() calls on the CPU

os.chdir(REF)
sys.path.insert(0, REF)
import lib  # noqa: E402
sys.modules["lib.linear"] = _ll
lib.linear = _ll

from lib.codebook.bitshift import bitshift_codebook, quantlut_sym  # noqa: E402
from lib.utils.kernel_decompress import decode_compressed  # noqa: E402
from lib.algo.ldlq import _INV_PERMUTE  # noqa: E402
from lib.quantizer import quant_op  # noqa: E402
from lib.quantizer.comb_quant import pack_trellis as ref_pack_trellis  # noqa: E402


def tail_biting_states(rng, ntiles, KV):
    """Random valid tail-biting state sequences: 128 windows of a random circular 128*KV-bit stream."""
    nb = 128 * KV
    bits = rng.integers(0, 2, size=(ntiles, nb), dtype=np.int64)
    ext = np.concatenate([bits, bits[:, :16]], axis=1)
    w = (1 << np.arange(15, -1, -1)).astype(np.int64)
    states = np.stack([(ext[:, t * KV:t * KV + 16] * w).sum(axis=1) for t in range(128)], axis=1)
    return states  # [ntiles, 128]


def gen_tcq(rng, m, k, S, KV):
    cb0 = bitshift_codebook(L=16, KV=KV, V=2, tlut_bits=S, decode_mode="quantlut_sym")
    tlut16 = cb0.tlut.half()                      # what the module stores (tcq_linear.py:37-40)
    cb = bitshift_codebook(L=16, KV=KV, V=2, tlut_bits=S, decode_mode="quantlut_sym", tlut=tlut16.float())
    ntr, ntc = m // 16, k // 16
    states = tail_biting_states(rng, ntr * ntc, KV)            # [tiles, 128], tile = tr*ntc + tc
    st = torch.from_numpy(states).to(torch.int32)
    # Qidxs layout (m, k/2): Qidxs[16b + t//8, 8i + t%8] = state t of tile (b, i)   (ldlq.py:107-110)
    Qidxs = st.reshape(ntr, ntc, 16, 8).transpose(1, 2).reshape(m, k // 2).contiguous()
    packed = ref_pack_trellis(Qidxs, 16, 16, cb, m, k, KV, 2)   # int16 [(m/16)(k/16), 8*KV]
    # expected weights: reference recons + inverse mma permutation (ldlq.py:100-104, bitshift.py:287-291)
    rec = cb.recons(st)                                         # [2, tiles, 128] fp32
    seq = rec.permute(1, 2, 0).reshape(ntr * ntc, 256)          # position 2t+e
    tiles = seq[:, _INV_PERMUTE].reshape(ntr, ntc, 16, 16)
    W = tiles.permute(0, 2, 1, 3).reshape(m, k)
    W16 = W.half()
    assert torch.equal(W16.float(), W), "recons values must be exactly representable in fp16"
    if KV % 2 == 0:
        # the reference's own full decoder (QTIP convention: R = KV/2 bits per weight, V = 1) must agree
        lut_pairs = quantlut_sym(tlut16.float(), 16, S)                            # [65536, 2]
        dec = decode_compressed(16, S, KV // 2, 1, m, k, packed.reshape(-1).view(torch.uint16), lut_pairs)
        assert torch.equal(dec, W), f"reference decode_compressed disagrees for KV={KV}"
    return dict(trellis=packed.numpy().astype(np.int16), tlut=tlut16.numpy(), W=W16.numpy(),
                states=states.astype(np.uint16))


def check_even_kv_with_reference_decoder(g, m, k, S, KV):
    """Even KV: decode the packed bytes with the reference's decode_indices and compare the states."""
    from lib.utils.kernel_decompress import decode_indices
    packed = torch.from_numpy(g["trellis"].copy())
    # QTIP convention: R bits per weight, V = log2(vector) : R = KV/2, V = 1 -> stride R<<V = KV
    idx = decode_indices(16, KV // 2, 1, m, k, packed.reshape(-1).view(torch.uint16))
    idx = idx.reshape(-1, 128).numpy().astype(np.uint16)
    assert np.array_equal(idx, g["states"]), f"reference decode_indices disagrees for KV={KV}"


def gen_lut_tc(rng, m, k, bits, vec):
    Q = torch.from_numpy(rng.integers(0, 1 << bits, size=(m, k // vec), dtype=np.int64))
    packed = quant_op.pack_qweight(Q, vec, bits)                 # int32/uint32 [m, bits*k/32/vec]
    packed = packed.view(torch.int32).reshape(m, -1).contiguous()
    if vec == 1:
        back = quant_op.dequantize_mat_sq_inds(packed, m, k, bits)
    else:
        back = quant_op.dequantize_mat_sq_inds_vec2(packed, m, k, bits)
    assert torch.equal(back.to(torch.int64), Q)
    lut = torch.randn((1 << bits, vec), generator=torch.Generator().manual_seed(bits * 10 + vec)).half()
    if vec == 1:
        W = quant_op.dequantize_mat_sq(packed, lut.reshape(-1), m, k, bits)
    else:
        W = lut[Q].reshape(m, k)
    return dict(qweight=packed.numpy(), idx=Q.numpy().astype(np.int32), lut=lut.numpy(),
                W=W.half().numpy())


def gen_simt(rng, m, k, bits, vec):
    Q = torch.from_numpy(rng.integers(0, 1 << bits, size=(m, k // vec), dtype=np.int64))
    if vec == 1:
        packed = quant_op.pack_qweight_sq_simt(Q, bits)
    else:
        packed = quant_op.pack_qweight_vq_simt(Q, bits, vec, bits)
    packed = packed.contiguous().numpy().view(np.uint32).reshape(m, -1)
    return dict(qweight=packed, idx=Q.numpy().astype(np.int32))


def gen_tc_to_simt(rng, m, k, bits, vec):
    Q = torch.from_numpy(rng.integers(0, 1 << bits, size=(m, k // vec), dtype=np.int64))
    tc = quant_op.pack_qweight(Q, vec, bits).view(torch.int32).reshape(m, -1).contiguous()
    simt = quant_op.convert_tensor_core_to_simt(tc, m, k, vec, bits, code_n=bits)
    return dict(tc=tc.numpy(), simt=simt.contiguous().numpy().view(np.uint32).reshape(m, -1),
                idx=Q.numpy().astype(np.int32))


def gen_hadamard():
    """get_hadK sign matrices and matmul_hadU / matmul_hadUt outputs (float64) for the block sizes the build
    constructs itself at small n, plus the two Llama-3.1-8B sizes and Llama-2-7B's 11008 = 172 * 64."""
    from lib.utils.matmul_had import get_hadK, matmul_hadU, matmul_hadUt
    rng = np.random.default_rng(20251011)
    out = {}
    for K in (12, 20, 28, 36, 60, 108, 140, 52, 116, 124, 156, 172):
        hadK, kk = get_hadK(K * 16)
        assert kk == K
        h = hadK.numpy()
        assert np.all(np.abs(h) == 1)
        out[f"hadK_{K}"] = np.packbits(h > 0, axis=1)          # sign bits, row-major
    sizes = [64, 128, 12 * 16, 20 * 32, 28 * 16, 36 * 16, 60 * 16, 4096, 14336, 52 * 16, 172 * 64]
    for n in sizes:
        x = rng.standard_normal((2, n)).astype(np.float32)
        xt = torch.from_numpy(x).to(torch.float64)
        out[f"x_{n}"] = x
        out[f"hadU_{n}"] = matmul_hadU(xt).numpy()
        out[f"hadUt_{n}"] = matmul_hadUt(xt).numpy()
        print(f"hadamard n={n} K={get_hadK(n)[1]} ok", flush=True)
    np.savez_compressed(os.path.join(OUT, "hadamard.npz"), sizes=np.array(sizes), **out)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "hadamard":
        gen_hadamard()
        return
    gen_hadamard()
    rng = np.random.default_rng(20251010)
    out = {}
    m, k = 64, 160
    combos = [(9, kv) for kv in range(2, 11)] + [(10, 8), (10, 9), (10, 10), (11, 9), (11, 10)]
    for S, KV in combos:
        g = gen_tcq(rng, m, k, S, KV)
        if KV % 2 == 0:
            check_even_kv_with_reference_decoder(g, m, k, S, KV)
        for key, v in g.items():
            out[f"tcq_S{S}_KV{KV}_{key}"] = v
        print(f"tcq S={S} KV={KV} ok", flush=True)
    np.savez_compressed(os.path.join(OUT, "tcq.npz"), m=m, k=k, **out)

    out = {}
    for vec, blist in [(1, range(2, 9)), (2, range(2, 13))]:
        for bits in blist:
            g = gen_lut_tc(rng, 64, 192, bits, vec)
            for key, v in g.items():
                out[f"tc_v{vec}_b{bits}_{key}"] = v
            print(f"lut_tc vec={vec} bits={bits} ok", flush=True)
    np.savez_compressed(os.path.join(OUT, "lut_tc.npz"), m=64, k=192, **out)

    out = {}
    for vec, blist, kk in [(1, range(2, 9), 2560), (2, range(3, 13), 2560), (4, range(6, 13), 5120)]:
        for bits in blist:
            g = gen_simt(rng, 4, kk, bits, vec)
            for key, v in g.items():
                out[f"simt_v{vec}_b{bits}_{key}"] = v
            out[f"simt_v{vec}_k"] = kk
            print(f"simt vec={vec} bits={bits} ok", flush=True)
    for vec, bits in [(1, 3), (1, 4), (1, 8), (2, 5), (2, 8), (2, 12)]:
        g = gen_tc_to_simt(rng, 32, 2560, bits, vec)
        for key, v in g.items():
            out[f"conv_v{vec}_b{bits}_{key}"] = v
        print(f"tc->simt vec={vec} bits={bits} ok", flush=True)
    np.savez_compressed(os.path.join(OUT, "simt.npz"), m=4, conv_m=32, conv_k=2560, **out)


if __name__ == "__main__":
    main()
