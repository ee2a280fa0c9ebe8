"""Persistent chain launches (C-ABI qpal_*_chain_build / qpal_chain_launch, csrc/tc_chain.h).

A chain is a sequence of DEPENDENT multi-job GEMV phases of one codec and one batch — q|k|v -> o -> gate|up -> down -> the
next block's q|k|v ... — executed by ONE kernel launch: stream order between the phases is kept by an in-kernel arrival
counter, and everything that does not depend on the activations (codebook image, the first weight steps and their decode)
runs ahead of it.  The reference launches one cold kernel per linear (lib/linear/tcq_linear.py:64-85 ->
kernels/tcq-kernels/src/inference.cu:1826-1860); at batch 1 on MI355X the kernel boundaries and cold prologues of such a
sequence cost more than the GEMVs themselves.

    plan = [Phase(layers=[q, k, v], x=x_attn, prezero=o_out), Phase(layers=[o], x=x_o, outs=[o_out], outs_zeroed=True), ...]
    for chain in build_chains(plan, n=1): chain.launch()        # graph-capturable; replay needs no reset
"""
import ctypes
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import torch

from . import _native as nat
from .linear import CombtLinearTCQ, QTIPLinearTCQ, VQLinearPackTensorCore

MAX_CHAIN_BATCH = 8


@dataclass
class Phase:
    """One multi-job GEMV of a chain: `layers` share the input.  Exactly one of x (fp16 [n, k]) / x_f32 (fp32 [n, k] written
    by an EARLIER phase of the same chain, scale) is given.  outs: fp32 [n, m_i] tensors (allocated when None);
    outs_zeroed: they hold zeros when the phase starts (the prezero of an earlier phase) so a split-K layer may accumulate
    with atomics; prezero: a tensor this phase zero-fills for a later one; wscales / oscale: fused epilogue
    out = acc * wscale[row] * oscale; x_fresh: x is written by an earlier phase of the chain; publish: a later phase of the
    chain reads the outputs (as x_f32)."""
    layers: list
    x: Optional[torch.Tensor] = None
    x_f32: Optional[Tuple[torch.Tensor, float]] = None
    outs: Optional[List[torch.Tensor]] = None
    outs_zeroed: bool = False
    prezero: Optional[torch.Tensor] = None
    wscales: Optional[list] = None
    oscale: float = 1.0
    x_fresh: bool = False
    publish: bool = False
    results: list = field(default_factory=list)


def chain_key(layer):
    """Layers with equal keys can be phases (or jobs of a phase) of one chain: same kernel instantiation."""
    if isinstance(layer, QTIPLinearTCQ):
        return ("tcq", layer.tlut_bits, layer.KV, 0, nat.QPAL_SPLIT_NONE)
    if isinstance(layer, CombtLinearTCQ) and layer.use_comb_kernel:
        return ("tcq", layer.tlut_bits, layer.KV[0], layer.KV[1], nat.QPAL_SPLIT_COLS)
    if isinstance(layer, VQLinearPackTensorCore):
        return ("lut", layer.lut_bits, layer.vec_sz)
    return None


_WS = {}


def workspace(device):
    """The 2 KiB arrival-counter workspace of a device: zero-filled once, shared by every chain launched there."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if idx not in _WS:
        _WS[idx] = torch.zeros(nat.CHAIN_WS_BYTES, dtype=torch.uint8, device=torch.device("cuda", idx))
    return _WS[idx]


def chain_error(device):
    """!= 0 after a synchronisation: a dependency wait of a chain launch gave up (1 + phase index)."""
    return int(workspace(device).view(torch.int32)[65].item())


class GemvChain:
    def __init__(self, phases, n, key, device):
        assert 1 <= n <= MAX_CHAIN_BATCH
        self.phases, self.n, self.key, self.device = phases, n, key, torch.device(device)
        lib = nat.lib()
        ncu = torch.cuda.get_device_properties(self.device).multi_processor_count
        self._keep = []
        cph = (nat.ChainPhase * len(phases))()
        for pi, ph in enumerate(phases):
            k = ph.layers[0].in_features
            assert all(l.in_features == k and chain_key(l) == key for l in ph.layers)
            assert (ph.x is None) != (ph.x_f32 is None), "exactly one of x / x_f32"
            xp = x32p = None
            xscale = 1.0
            if ph.x is not None:
                assert ph.x.is_cuda and ph.x.dtype == torch.float16 and ph.x.is_contiguous() and tuple(ph.x.shape) == (n, k)
                xp = ph.x.data_ptr()
            else:
                t, xscale = ph.x_f32
                assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and tuple(t.shape) == (n, k)
                x32p = t.data_ptr()
            if ph.outs is None:
                ph.outs = [torch.empty((n, l.out_features), dtype=torch.float32, device=self.device) for l in ph.layers]
            ph.results = ph.outs
            tcq = key[0] == "tcq"
            jobs = ((nat.TcqJob if tcq else nat.LutJob) * len(ph.layers))()
            for j, (l, out) in enumerate(zip(ph.layers, ph.outs)):
                m = l.out_features
                assert out.is_cuda and out.dtype == torch.float32 and tuple(out.shape) == (n, m) and out.stride(1) == 1
                ldo = out.stride(0) if n > 1 else m
                ws = None
                if ph.wscales is not None and ph.wscales[j] is not None:
                    w = ph.wscales[j]
                    assert w.is_cuda and w.dtype == torch.float16 and w.is_contiguous() and w.numel() == m
                    ws = w.data_ptr()
                    self._keep.append(w)
                common = dict(m=m, k=k, out_zeroed=1 if ph.outs_zeroed else 0, wscale=ws, oscale=float(ph.oscale), ldo=ldo,
                              x_had=0, x_post=1.0, x_su=None, x_f32=x32p, x_f32_scale=float(xscale),
                              x_fresh=1 if ph.x_fresh else 0, publish=1 if ph.publish else 0)
                if tcq:
                    if key[4] == nat.QPAL_SPLIT_NONE:
                        c1, c2 = l.trellis, None
                    else:
                        c1, c2 = l.trellis1, l.trellis2
                    jobs[j] = nat.TcqJob(out=out.data_ptr(), c1=c1.data_ptr(), c2=c2.data_ptr() if c2 is not None else None,
                                         x=xp, tlut=l.tlut.data_ptr(), kv=0, **common)
                    self._keep += [c1, c2, l.tlut]
                else:
                    jobs[j] = nat.LutJob(out=out.data_ptr(), qweight=l.qweight.data_ptr(), x=xp, lut=l.lut.data_ptr(), **common)
                    self._keep += [l.qweight, l.lut]
            self._keep.append(jobs)
            cph[pi].njobs = len(ph.layers)
            if tcq:
                cph[pi].tcq_jobs = ctypes.cast(jobs, ctypes.POINTER(nat.TcqJob))
            else:
                cph[pi].lut_jobs = ctypes.cast(jobs, ctypes.POINTER(nat.LutJob))
            if ph.prezero is not None:
                z = ph.prezero
                nbytes = z.numel() * z.element_size()
                assert z.is_cuda and z.is_contiguous() and nbytes % 16 == 0 and z.data_ptr() % 16 == 0
                cph[pi].prezero, cph[pi].prezero_bytes = z.data_ptr(), nbytes
        nbytes = lib.qpal_chain_blob_bytes(len(phases))
        self._host = (ctypes.c_char * nbytes)()
        if key[0] == "tcq":
            rc = lib.qpal_tcq_chain_build(self._host, nbytes, cph, len(phases), n, key[1], key[2], key[3], key[4], ncu)
        else:
            rc = lib.qpal_lut_chain_build(self._host, nbytes, cph, len(phases), n, key[1], key[2], ncu)
        nat.check(rc, "qpal_chain_build")
        self._dev = torch.frombuffer(self._host, dtype=torch.uint8).to(self.device)
        self._ws = workspace(self.device)

    @property
    def nphases(self):
        return len(self.phases)

    def launch(self, dbg=None):
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            rc = nat.lib().qpal_chain_launch(self._dev.data_ptr(), self._host, self._ws.data_ptr(),
                                             dbg.data_ptr() if dbg is not None else None, stream)
        nat.check(rc, "qpal_chain_launch")


def chainable(phase, n):
    """Can this phase be part of a chain?  (one codec over all its layers, batch <= 8)"""
    if n > MAX_CHAIN_BATCH:
        return None
    keys = {chain_key(l) for l in phase.layers}
    if len(keys) != 1 or None in keys or len(phase.layers) > 8:
        return None
    return next(iter(keys))


def build_chains(plan, n, device, min_len=1):
    """Partition `plan` (Phases in execution order) into maximal runs of one codec -> list of GemvChain / Phase: a Phase that no
    chain can take (SIMT packing, unequal comb parts, mixed codecs inside one phase, a shape the chain planner refuses) is
    returned as is for the caller to run with multi_gemv."""
    out, run, run_key = [], [], None

    def flush():
        nonlocal run, run_key
        if run:
            if len(run) >= min_len:
                try:
                    out.append(GemvChain(run, n, run_key, device))
                except nat.QpalError:
                    out.extend(run)
            else:
                out.extend(run)
        run, run_key = [], None

    for ph in plan:
        key = chainable(ph, n)
        if key is None:
            flush()
            out.append(ph)
            continue
        if key != run_key:
            flush()
            run_key = key
        run.append(ph)
    flush()
    return out
