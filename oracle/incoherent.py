"""CPU ORACLE (numpy, float64) of the incoherence steps either side of the packed GEMV.

TEST INFRASTRUCTURE ONLY (same rule as oracle/oracle.py): imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, never by the product package.

Restates, with the reference's fp16 rounding points made explicit:
  * matmul_hadU / matmul_hadUt            lib/utils/matmul_had.py:67-91        (pinned: tests/golden/hadamard.npz)
  * matmul_hadU_cuda                      lib/utils/matmul_had.py:137-148
  * matmul_hadU_head_cuda                 lib/utils/matmul_had.py:95-110
  * IncoherentMLP.compute_ug / compute_dp lib/linear/incoherent_linear.py:325-342
  * IncoherentSdpaAttention.compute_qkv/o lib/linear/incoherent_linear.py:76-109
  * IncoherentLinear.forward              lib/linear/incoherent_linear.py:486-507
The butterflies of matmul_hadU_cuda live in the third-party package fast_hadamard_transform (Dao-AILab,
un-vendored and unpinned in the reference: kernels/README.md:19-23); its published contract is
``hadamard_transform(x, scale) = (x @ H_n) * scale`` with H_n the Sylvester matrix, computed in fp32 and
returned in x's dtype.  That contract is what `wht` + `f16` below restate; the index convention of the
hadK (x) H_P composition is pinned against the reference's pure-torch matmul_hadU golden outputs.
Parity status of the fp16 rounding points: restated from the source text, not executable here (CUDA only).
"""
import numpy as np


def f16(a):
    """Round to fp16 and come back (the reference keeps activations in torch.float16)."""
    return np.asarray(a, dtype=np.float64).astype(np.float16).astype(np.float64)


def wht(a):
    """Unnormalised Walsh-Hadamard transform (Sylvester order) over the last axis (a power of two)."""
    a = np.array(a, dtype=np.float64)
    n = a.shape[-1]
    assert n & (n - 1) == 0
    lead = a.shape[:-1]
    h = 1
    while h < n:
        a = a.reshape(*lead, n // (2 * h), 2, h)
        a = np.stack([a[..., 0, :] + a[..., 1, :], a[..., 0, :] - a[..., 1, :]], axis=-2)
        h *= 2
    return a.reshape(*lead, n)


def unpack_hadk(bits, K):
    """tests/golden/hadamard.npz stores get_hadK sign matrices as packbits rows."""
    return np.unpackbits(bits, axis=1)[:, :K].astype(np.float64) * 2 - 1


def had_blocks(x, hd, hadk=None, round_mid=False):
    """(hadK (x) H_P) / sqrt(hd) over every block of hd consecutive elements of the last axis, block viewed as
    [K][P].  round_mid: fp16 between the two factors (matmul_hadU_cuda); hadk is applied as given."""
    x = np.asarray(x, dtype=np.float64)
    n = x.shape[-1]
    K = 1 if hadk is None else hadk.shape[0]
    P = hd // K
    t = wht(x.reshape(*x.shape[:-1], n // hd, K, P)) * (float(hd) ** -0.5)
    if K > 1:
        if round_mid:
            t = f16(t)
        t = np.einsum("ji,...ic->...jc", np.asarray(hadk, dtype=np.float64), t)
    return t.reshape(x.shape)


def matmul_hadU_cuda(x16, hadk=None, scale_div=None):
    """fp16 in, fp16 out; optional trailing `/ scale` (an fp16 op in the reference)."""
    y = f16(had_blocks(f16(x16), x16.shape[-1], hadk, round_mid=True))
    return f16(y / scale_div) if scale_div else y


def matmul_hadU_head_cuda(x16, hd, hadk=None):
    """float path, one rounding at the end (`.to(X.dtype)`)."""
    return f16(had_blocks(f16(x16), hd, hadk, round_mid=False))


def silu(a):
    return a / (1.0 + np.exp(-a))


def linear_post(y_acc, wscale, scale):
    """`linear(x) * Wscale * scale` with the reference's roundings: fp32 accumulators -> fp16, then two fp16
    multiplies."""
    return f16(f16(f16(y_acc) * f16(wscale)) * scale)


def left_input(x, su, hadk, scale):
    """x.half() * SU -> matmul_hadU_cuda -> / scale   (incoherent_linear.py:81, 106, 326, 336)."""
    return matmul_hadU_cuda(f16(f16(x) * f16(su)), hadk, scale_div=scale)


def swiglu(up16, gate16):
    """act_fn(x_gate) * x_up in fp16 (incoherent_linear.py:333)."""
    return f16(f16(silu(f16(gate16))) * f16(up16))
