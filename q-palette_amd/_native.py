"""ctypes binding of libqpal_hip.so (C-ABI: include/qpal.h).

The HIP library is the product: there is no fallback.  ``lib()`` raises if it has not been built
(``python -c 'import __graft_entry__ as g; g.build()'`` or ``make -C q-palette_amd/csrc``)."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("QPAL_LIB") or os.path.join(_HERE, "libqpal_hip.so")  # QPAL_LIB: perf experiments only
_lib = None

QPAL_SPLIT_NONE, QPAL_SPLIT_ROWS, QPAL_SPLIT_COLS = 0, 1, 2

_P, _I, _F = ctypes.c_void_p, ctypes.c_int, ctypes.c_float


class TcqJob(ctypes.Structure):
    """qpal_tcq_job (include/qpal.h)"""
    _fields_ = [("out", _P), ("c1", _P), ("c2", _P), ("x", _P), ("tlut", _P), ("m", _I), ("k", _I),
                ("out_zeroed", _I), ("wscale", _P), ("oscale", _F), ("ldo", ctypes.c_long),
                ("x_had", _I), ("x_post", _F), ("x_su", _P), ("kv", _I),
                ("x_f32", _P),
                ("x_rms_eps", _F), ("x_rms_w", _P), ("accumulate", _I), ("kv2", _I), ("act_out", _P), ("x_hadk", _P), ("x_K", _I), ("act_su", _P)]


class LutJob(ctypes.Structure):
    """qpal_lut_job (include/qpal.h)"""
    _fields_ = [("out", _P), ("qweight", _P), ("x", _P), ("lut", _P), ("m", _I), ("k", _I), ("out_zeroed", _I),
                ("wscale", _P), ("oscale", _F), ("ldo", ctypes.c_long), ("x_had", _I), ("x_post", _F), ("x_su", _P),
                ("x_f32", _P),
                ("x_rms_eps", _F), ("x_rms_w", _P), ("accumulate", _I), ("act_out", _P), ("x_hadk", _P), ("x_K", _I), ("act_su", _P)]


PEER_WS_BYTES_PER_SLOT = 256
IPC_HANDLE_BYTES = 64


_SIGNATURES = {
    "qpal_tcq_gemv_multi": [ctypes.POINTER(TcqJob), _I, _I, _I, _I, _I, _I, _P, ctypes.c_long, _P],
    "qpal_lut_tc_gemv_multi": [ctypes.POINTER(LutJob), _I, _I, _I, _I, _P, ctypes.c_long, _P],
    "qpal_tcq_gemv": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "qpal_tcq_dequant": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "qpal_lut_tc_gemv": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "qpal_lut_tc_dequant": [_P, _P, _P, _I, _I, _I, _I, _P],
    "qpal_lut_simt_gemv": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "qpal_lut_simt_dequant": [_P, _P, _P, _I, _I, _I, _I, _P],
    "qpal_tc_to_simt": [_P, _P, _I, _I, _I, _I, _P],
    "qpal_plan_gemv": [ctypes.POINTER(_I), ctypes.POINTER(_I), ctypes.POINTER(_I), ctypes.POINTER(_I), _I, _I, _I, ctypes.POINTER(_I), _I],
    "qpal_can_fuse_rotation": [_I, _I],
    "qpal_can_fuse_rotation_k": [_I, _I, _I],
    "qpal_pack_tcq": [_P, _P, _I, _I, _I],
    "qpal_pack_tcq_states": [_P, _P, _I, _I, _I],
    "qpal_pack_lut_tc": [_P, _P, _I, _I, _I, _I],
    "qpal_pack_lut_simt": [_P, _P, _I, _I, _I, _I],
    "qpal_hadamard": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _P],
    "qpal_hadamard_rms": [_P, _P, _P, _F, _P, _P, _I, _I, _I, _F, _P],
    "qpal_rope_kv": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, ctypes.c_long, _P],
    "qpal_attn_decode": [_P, _P, _P, _P, _P, _I, _I, _I, ctypes.c_long, _F, _P],
    "qpal_attn_rope_decode": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, ctypes.c_long, _F, _P, ctypes.c_long, _P],
    "qpal_attn_ws_bytes": [_I, _I, _I, ctypes.c_long],
    "qpal_lm_head_argmax": [_P, _P, _F, _P, _P, _P, _P, ctypes.c_long, _I, _I, _P],
    "qpal_lm_head_ws_bytes": [_I],
    "qpal_peer_gather": [_P, ctypes.c_long, _I, ctypes.POINTER(_P), ctypes.POINTER(_P), _I, _I, _P],
    "qpal_peer_alloc": [ctypes.POINTER(_P), ctypes.c_long, _I],
    "qpal_peer_free": [_P],
    "qpal_ipc_export": [_P, ctypes.c_char_p],
    "qpal_ipc_open": [ctypes.c_char_p, ctypes.POINTER(_P)],
    "qpal_ipc_close": [_P],
    "qpal_calib_stream_read": [ctypes.POINTER(_P), ctypes.POINTER(ctypes.c_long), _I, _P, _I, _P],
    "qpal_calib_decode_rate": [_P, _P, _I, _I, _I, _I, _P],
}


class QpalError(RuntimeError):
    pass


def build(verbose=False):
    """Compile every HIP translation unit for gfx950 and link libqpal_hip.so (cross-compiles on CPU)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j", str(min(8, os.cpu_count() or 1))]
    subprocess.check_call(cmd, stdout=None if verbose else subprocess.DEVNULL)
    return SO_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise QpalError(
                f"{SO_PATH} is missing: the HIP extension is the only implementation of this path "
                "(no CPU fallback). Build it with __graft_entry__.build() or `make -C q-palette_amd/csrc`.")
        l = ctypes.CDLL(SO_PATH)
        for name, args in _SIGNATURES.items():
            if os.environ.get("QPAL_LIB") and not hasattr(l, name):
                continue  # (perf experiments: an older build of the library beside the current host code)
            fn = getattr(l, name)
            fn.argtypes = args
            fn.restype = _I
        l.qpal_error_string.argtypes = [_I]
        l.qpal_error_string.restype = ctypes.c_char_p
        l.qpal_version.restype = _I
        l.qpal_attn_ws_bytes.restype = ctypes.c_long
        l.qpal_lm_head_ws_bytes.restype = ctypes.c_long
        _lib = l
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().qpal_error_string(rc).decode()
        raise QpalError(f"{what} failed: {msg} (code {rc})")


def exported_symbols():
    return list(_SIGNATURES) + ["qpal_error_string", "qpal_version"]
