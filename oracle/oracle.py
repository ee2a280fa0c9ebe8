"""ctypes front-end of the CPU ORACLE (oracle/qpal_oracle.c).

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module; the product package (q-palette_amd/) never does and fails loudly instead when
its HIP library is missing.  Parity status: pinned by tests/golden/*.npz (generated from the
reference's own Python by tests/golden/make_golden.py).

All arrays are numpy; fp16 tensors are passed as np.float16 (viewed as uint16 bit patterns).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libqpal_oracle.so")
_lib = None


def build(force=False):
    """Compile oracle/qpal_oracle.c with gcc (seconds)."""
    src = os.path.join(_HERE, "qpal_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "_build/libqpal_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.qo_num_threads.restype = ctypes.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _u16(a):
    a = np.ascontiguousarray(a)
    if a.dtype == np.float16 or a.dtype == np.int16:
        a = a.view(np.uint16)
    assert a.dtype == np.uint16, a.dtype
    return a


def _u32(a):
    a = np.ascontiguousarray(a)
    if a.dtype == np.int32:
        a = a.view(np.uint32)
    assert a.dtype == np.uint32, a.dtype
    return a


def num_threads():
    return int(lib().qo_num_threads())


def set_num_threads(n):
    lib().qo_set_num_threads(ctypes.c_int(n))


# ---------------------------------------------------------------- TCQ
def tcq_states(trellis, m, k, KV):
    """int16 trellis [(m/16)(k/16), 8*KV] -> uint16 states [m/16, k/16, 128] (t = 4*lane + j)."""
    t = _u16(trellis)
    assert t.size * 32 == m * k * KV, (t.size, m, k, KV)
    out = np.empty((m // 16, k // 16, 128), dtype=np.uint16)
    lib().qo_tcq_states(_p(t), m, k, KV, _p(out))
    return out


def tcq_dequant(c1, tlut, m, k, S, KV1, c2=None, KV2=0, split=0):
    """fp16 W [m, k].  split: 0 single, 1 comb (row halves), 2 combt (column halves)."""
    c1 = _u16(c1)
    c2a = _u16(c2) if c2 is not None else c1
    tl = _u16(tlut)
    assert tl.size == 2 << S
    W = np.empty((m, k), dtype=np.uint16)
    rc = lib().qo_tcq_dequant(_p(c1), _p(c2a), _p(tl), m, k, S, KV1, KV2, split, _p(W))
    assert rc == 0
    return W.view(np.float16)


# ---------------------------------------------------------------- LUT formats
def lut_tc_indices(qweight, m, k, bits, vec):
    q = _u32(qweight)
    assert q.size * 32 * vec == m * k * bits
    idx = np.empty((m, k // vec), dtype=np.int32)
    lib().qo_lut_tc_indices(_p(q), m, k, bits, vec, _p(idx))
    return idx


def lut_tc_dequant(qweight, lut, m, k, bits, vec):
    q = _u32(qweight)
    l = _u16(lut)
    assert l.size == vec << bits
    W = np.empty((m, k), dtype=np.uint16)
    lib().qo_lut_tc_dequant(_p(q), _p(l), m, k, bits, vec, _p(W))
    return W.view(np.float16)


def simt_indices(qweight, m, k, bits, vec):
    q = _u32(qweight)
    assert q.size * 32 * vec == m * k * bits
    idx = np.empty((m, k // vec), dtype=np.int32)
    lib().qo_simt_indices(_p(q), m, k, bits, vec, _p(idx))
    return idx


def simt_dequant(qweight, lut, m, k, bits, vec):
    q = _u32(qweight)
    l = _u16(lut)
    assert l.size == vec << bits
    W = np.empty((m, k), dtype=np.uint16)
    lib().qo_simt_dequant(_p(q), _p(l), m, k, bits, vec, _p(W))
    return W.view(np.float16)


# ---------------------------------------------------------------- GEMV
def gemv(W, x):
    """W fp16 [m,k], x fp16 [n,k] -> (out float64 [n,m], abs-sum float64 [n,m])."""
    W = _u16(W)
    x = _u16(x)
    m, k = W.shape
    n = x.shape[0]
    assert x.shape[1] == k
    out = np.empty((n, m), dtype=np.float64)
    aout = np.empty((n, m), dtype=np.float64)
    lib().qo_gemv_f16(_p(W), _p(x), m, n, k, _p(out), _p(aout))
    return out, aout


# ---------------------------------------------------------------- CPU baseline (bench.py only)
def cpu_tcq_linear(c1, c2, tlut, x, m, n, k, S, KV1, KV2, split, scratch=None):
    c1 = _u16(c1)
    c2a = _u16(c2) if c2 is not None else c1
    tl = _u16(tlut)
    xx = _u16(x)
    if scratch is None:
        scratch = np.empty((m, k), dtype=np.uint16)
    out = np.empty((n, m), dtype=np.float32)
    rc = lib().qo_cpu_tcq_linear(_p(c1), _p(c2a), _p(tl), _p(xx), m, n, k, S, KV1, KV2, split,
                                 _p(scratch), _p(out))
    assert rc == 0
    return out


def cpu_lut_tc_linear(qweight, lut, x, m, n, k, bits, vec, scratch=None):
    q = _u32(qweight)
    l = _u16(lut)
    xx = _u16(x)
    if scratch is None:
        scratch = np.empty((m, k), dtype=np.uint16)
    out = np.empty((n, m), dtype=np.float32)
    lib().qo_cpu_lut_tc_linear(_p(q), _p(l), _p(xx), m, n, k, bits, vec, _p(scratch), _p(out))
    return out


def cpu_tcq_linear_fused(c1, c2, tlut, x, m, n, k, S, KV1, KV2, split):
    """Fused CPU variant: decode tile by tile and multiply at once, W is never written (bench.py cpu_baseline, tests)."""
    c1 = _u16(c1)
    c2a = _u16(c2) if c2 is not None else c1
    out = np.empty((n, m), dtype=np.float32)
    rc = lib().qo_cpu_tcq_linear_fused(_p(c1), _p(c2a), _p(_u16(tlut)), _p(_u16(x)), m, n, k, S, KV1, KV2, split, _p(out))
    assert rc == 0
    return out


def cpu_lut_tc_linear_fused(qweight, lut, x, m, n, k, bits, vec):
    out = np.empty((n, m), dtype=np.float32)
    rc = lib().qo_cpu_lut_tc_linear_fused(_p(_u32(qweight)), _p(_u16(lut)), _p(_u16(x)), m, n, k, bits, vec, _p(out))
    assert rc == 0
    return out

