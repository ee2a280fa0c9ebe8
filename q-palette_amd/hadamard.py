"""Hadamard rotations of the incoherence wrapper, on the C-ABI's ``qpal_hadamard``.

Host-side mirror of the reference's lib/utils/matmul_had.py for the functions on the inference path:
``get_hadK`` (l.10-64), ``matmul_hadU_cuda`` / ``matmul_hadUt_cuda`` (l.137-151), ``matmul_hadU_head_cuda`` /
``matmul_hadUt_head_cuda`` (l.95-113) and the ``hadamard::hadamard`` torch op (l.124-134; there it forwards
to the third-party fast_hadamard_transform).  Differences by design:

* the reference ships its K x K Hadamard factors as 95 k lines of literals; here the ones with a classical
  are GENERATED — Paley I for K = 12, 20, 60, 108, 140, Paley II for K = 28, 36, and Williamson arrays of four
  symmetric circulants for K = 52, 116, 124, 156, 172 (the reference's literals turn out to be exactly that: the four
  first rows, K bits per matrix, are the only constants kept) — and checked entry for entry against ``get_hadK``
  outputs in tests/golden/hadamard.npz; ``register_hadK`` accepts any other factor from the caller;
* sign flip, both Hadamard factors, the scales and (optionally) SwiGLU run in ONE launch (``rotate``).
"""
import math

import torch

from . import _native

IN_F16, IN_F32, IN_SWIGLU_F32 = 0, 1, 2

_PALEY = {12: (1, 11), 20: (1, 19), 28: (2, 13), 36: (2, 17), 60: (1, 59), 108: (1, 107), 140: (1, 139)}
# Williamson arrays [[A, B, C, D], [-B, A, -D, C], [-C, D, A, -B], [-D, -C, B, A]]: first rows of the circulants A, B, C, D
# (bit i from the top = element i; 1 = +1), found by analysing the reference's get_had52 .. get_had172 outputs
_WILLIAMSON = {
    52: (0x14f2, 0x11f8, 0x1a65, 0x1090),
    116: (0x192ef749, 0x1ed1f8b7, 0x1465fa62, 0x1c650a63),
    124: (0x4e90c25c, 0x4e90c25c, 0x41d4cae0, 0x7e2b351f),
    156: (0x7282619053, 0x7898528647, 0x73452128b3, 0x46a0edc158),
    172: (0x467aecdd798, 0x6fc29b650fd, 0x7595e85ea6b, 0x63d260192f1),
}
# precedence of the reference's get_hadK (a size divisible by several K takes the first)
_ORDER = (172, 156, 140, 124, 116, 108, 60, 52, 36, 28, 20, 12)
_registered = {}
_cache = {}


def is_pow2(n):
    return n > 0 and (n & (n - 1)) == 0


def _jacobsthal(q):
    residues = {(i * i) % q for i in range(1, q)}
    chi = [0] + [1 if a in residues else -1 for a in range(1, q)]
    idx = torch.arange(q)
    return torch.tensor(chi, dtype=torch.float32)[(idx[:, None] - idx[None, :]) % q]


def _paley(kind, q):
    n = q + 1
    core = torch.zeros(n, n)
    core[1:, 1:] = _jacobsthal(q)
    eye = torch.eye(n)
    if kind == 1:            # q = 3 (mod 4): H = I + S, S skew with first row -1, first column +1
        core[0, 1:] = -1.0
        core[1:, 0] = 1.0
        return core + eye
    core[0, 1:] = 1.0        # q = 1 (mod 4): symmetric conference matrix C, H = [[C+I, C-I], [C-I, -C-I]]
    core[1:, 0] = 1.0
    return torch.cat([torch.cat([core + eye, core - eye], 1), torch.cat([core - eye, -core - eye], 1)], 0)


def _williamson(K, seeds):
    q = K // 4
    idx = torch.arange(q)
    blocks = []
    for seed in seeds:
        row = torch.tensor([1.0 if (seed >> (q - 1 - i)) & 1 else -1.0 for i in range(q)])
        blocks.append(row[(idx[None, :] - idx[:, None]) % q])  # circulant: row r = first row rotated right by r
    a, b, c, d = blocks
    return torch.cat([torch.cat([a, b, c, d], 1), torch.cat([-b, a, -d, c], 1), torch.cat([-c, d, a, -b], 1),
                      torch.cat([-d, -c, b, a], 1)], 0)


def register_hadK(K, matrix):
    """Supply a K x K +-1 Hadamard factor the build does not construct itself."""
    m = torch.as_tensor(matrix, dtype=torch.float32)
    if m.shape != (K, K) or not bool((m.abs() == 1).all()) or not torch.equal(m @ m.T, K * torch.eye(K)):
        raise ValueError(f"not a {K}x{K} Hadamard matrix")
    _registered[K] = m.clone()
    _cache.pop(K, None)


def hadK_matrix(K):
    if K not in _cache:
        if K in _registered:
            _cache[K] = _registered[K]
        elif K in _PALEY:
            _cache[K] = _paley(*_PALEY[K])
        elif K in _WILLIAMSON:
            _cache[K] = _williamson(K, _WILLIAMSON[K])
        else:
            raise NotImplementedError(f"no built-in {K}x{K} Hadamard factor: pass one to "
                                      f"qpalette_amd.hadamard.register_hadK({K}, matrix)")
    return _cache[K]


def get_hadK(n, transpose=False):
    """(hadK fp32 [K, K] or None, K): the factorisation n = K * 2^p the reference uses (matmul_had.py:10-64)."""
    for K in _ORDER:
        if n % K == 0:
            if not is_pow2(n // K):
                raise AssertionError(f"{n} / {K} is not a power of two")
            h = hadK_matrix(K)
            return (h.T.contiguous() if transpose else h.clone()), K
    if not is_pow2(n):
        raise AssertionError(f"{n} has no supported Hadamard factorisation")
    return None, 1


def rotate(x, hd=None, hadK=None, K=1, su=None, sv=None, post_scale=1.0, in_mode=IN_F16, round_mid=True, out=None, rms=None):
    """One launch of ``qpal_hadamard`` (see include/qpal.h).  x: [..., n] fp16 | fp32, or [..., 2n] fp32 for
    in_mode = IN_SWIGLU_F32 (up | gate).  Returns fp16 [..., n].
    rms = (eps, weight fp16 [n] or None): RMSNorm of the fp32 row in front of the rotation, same launch (``qpal_hadamard_rms``;
    in_mode IN_F32, hd = n, no sv) — the decoder-block fusion for widths the GEMV staging cannot rotate itself."""
    if not x.is_cuda:
        raise _native.QpalError("qpalette_amd has no CPU implementation: rotate() needs device tensors")
    want = torch.float16 if in_mode == IN_F16 else torch.float32
    if x.dtype != want:
        raise _native.QpalError(f"rotate: in_mode {in_mode} needs {want}, got {x.dtype}")
    x = x.contiguous()
    n = x.shape[-1] // (2 if in_mode == IN_SWIGLU_F32 else 1)
    hd = n if hd is None else hd
    rows = x.numel() // x.shape[-1]
    if out is None:
        out = torch.empty(*x.shape[:-1], n, dtype=torch.float16, device=x.device)

    def vec(t, name):
        if t is None:
            return None
        if t.dtype != torch.float16 or t.numel() != n or not t.is_cuda or not t.is_contiguous() or t.device != x.device:
            raise _native.QpalError(f"rotate: {name} must be a contiguous fp16 vector of {n} elements on {x.device}")
        return t.data_ptr()

    hk = None
    if K > 1:
        if (hadK is None or hadK.dtype != torch.float16 or tuple(hadK.shape) != (K, K) or not hadK.is_cuda
                or hadK.device != x.device):
            raise _native.QpalError(f"rotate: hadK must be an fp16 matrix [{K}, {K}] on {x.device}")
        hadK = hadK.contiguous()
        hk = hadK.data_ptr()
    if not x.is_cuda or out.device != x.device:
        raise _native.QpalError("rotate: x and out must live on the same GPU")
    # launch on x's device and ITS current stream (a module on cuda:N of a device_map'ed model is called while another
    # device is current: the GEMVs around the rotation already run on cuda:N's stream, ops.py)
    if rms is not None:
        eps, w = rms
        if in_mode != IN_F32 or hd != n or sv is not None or not round_mid:
            raise _native.QpalError("rotate: rms needs in_mode IN_F32, hd = n, no sv")
        with torch.cuda.device_of(x):
            _native.check(_native.lib().qpal_hadamard_rms(out.data_ptr(), x.data_ptr(), vec(w, "rms weight"), float(eps), vec(su, "su"),
                                                          hk, rows, n, K, float(post_scale),
                                                          torch.cuda.current_stream(x.device).cuda_stream), "qpal_hadamard_rms")
        return out
    with torch.cuda.device_of(x):
        _native.check(_native.lib().qpal_hadamard(out.data_ptr(), x.data_ptr(), vec(su, "su"), vec(sv, "sv"), hk, rows, n, hd,
                                                  K, in_mode, 1 if round_mid else 0, float(post_scale),
                                                  torch.cuda.current_stream(x.device).cuda_stream), "qpal_hadamard")
    return out


def _as_f16_matrix(hadK, K, device):
    if K == 1:
        return None
    return hadK.to(device=device, dtype=torch.float16)


def matmul_hadU_cuda(X, hadK, K, part=1, transpose=False):
    """reference matmul_had.py:137-148 (fp16 pipeline, whole last dimension)."""
    assert part == 1, "only part = 1 is supported"
    if hadK is not None and transpose:
        hadK = hadK.T.contiguous()
    y = rotate(X.to(torch.float16), hadK=_as_f16_matrix(hadK, K, X.device), K=K, round_mid=True)
    return y.to(X.dtype).reshape(X.shape)


def matmul_hadUt_cuda(X, hadK, K):
    return matmul_hadU_cuda(X, hadK, K, transpose=True)


def matmul_hadU_head_cuda(X, hadK, K, head_dim, transpose=False):
    """reference matmul_had.py:95-110 (float pipeline, blocks of head_dim)."""
    if hadK is not None and transpose:
        hadK = hadK.T.contiguous()
    y = rotate(X.to(torch.float16), hd=head_dim, hadK=_as_f16_matrix(hadK, K, X.device), K=K, round_mid=False)
    return y.to(X.dtype).reshape(X.shape)


def matmul_hadUt_head_cuda(X, hadK, K, head_dim):
    return matmul_hadU_head_cuda(X, hadK, K, head_dim, transpose=True)


_op_defined = False


def ensure_hadamard_op():
    """``torch.ops.hadamard.hadamard(x, scale)`` (reference matmul_had.py:124-134): power-of-two transform of the
    last dimension times ``scale``, output in x's dtype."""
    global _op_defined
    if _op_defined:
        return
    _op_defined = True
    torch.library.define("hadamard::hadamard", "(Tensor x, float scale) -> Tensor")

    @torch.library.register_fake("hadamard::hadamard")
    def _(x, scale):
        return torch.empty_like(x)

    @torch.library.impl("hadamard::hadamard", "CUDA")
    def _(x, scale):
        if x.dtype != torch.float16:
            raise _native.QpalError("hadamard::hadamard is implemented for fp16 tensors (the inference path's dtype)")
        return rotate(x, post_scale=scale * math.sqrt(x.shape[-1]))
