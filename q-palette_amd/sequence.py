"""Launch sequences: the library's own replayable list of dependent launches (include/qpal.h qpal_seq_*).

The reference replays a decoded token from a CUDA graph (eval/measure_latency.py:130-135, 236-254).  On MI355X a HIP graph keeps
every kernel boundary of the token: ~130 dependent launches whose boundaries (the command processor's barrier, the new workgroups'
start-up and prologue) cost 2-3 us each.  A sequence records the same calls once and re-issues them per token from C++ — the fused
GEMV launches without the barrier bit of their AQL packet, their dependency kept by in-kernel arrival counters, so that a launch's
prologue runs while the launch before it drains (csrc/seq.hip).  Results are bit-identical to the stream-ordered calls.

    seq = LaunchSequence()
    with seq.record():
        outs = token()          # any calls into this package: recorded, not launched
    seq.launch()                # per token, on the current stream
"""
import contextlib
import ctypes

from . import _native


class LaunchSequence:
    def __init__(self):
        self._h = ctypes.c_void_p()
        _native.check(_native.lib().qpal_seq_create(ctypes.byref(self._h)), "qpal_seq_create")
        self.recorded = None  # what the recorded callable returned: keeps the buffers the launches name alive

    @contextlib.contextmanager
    def record(self):
        lib = _native.lib()
        _native.check(lib.qpal_seq_begin(self._h), "qpal_seq_begin")
        try:
            yield self
        finally:
            _native.check(lib.qpal_seq_end(self._h), "qpal_seq_end")

    def capture(self, fn):
        """record fn() and keep its result (the output tensors) alive with the sequence"""
        with self.record():
            self.recorded = fn()
        return self.recorded

    def launch(self, stream=None):
        import torch

        s = stream if stream is not None else torch.cuda.current_stream()
        _native.check(_native.lib().qpal_seq_launch(self._h, s.cuda_stream), "qpal_seq_launch")

    def info(self, read_error=False):
        """-> {"launches", "overlapped"[, "error"]}; read_error synchronises nothing itself: synchronise the stream first"""
        n, w, e = ctypes.c_int(), ctypes.c_int(), ctypes.c_uint()
        _native.check(_native.lib().qpal_seq_info(self._h, ctypes.byref(n), ctypes.byref(w), ctypes.byref(e) if read_error else None),
                      "qpal_seq_info")
        d = {"launches": n.value, "overlapped": w.value}
        if read_error:
            d["error"] = e.value
        return d

    def __del__(self):
        try:
            if self._h:
                _native.lib().qpal_seq_destroy(self._h)
                self._h = None
        except Exception:
            pass
