"""qpalette_amd — MI355X (gfx950) native implementation of Q-Palette's dequant-matmul hot path.

Layout:
  csrc/            hand-written HIP kernels + the C-ABI (include/qpal.h) -> libqpal_hip.so
  _native.py       ctypes binding of the C-ABI (fails loudly when the library is missing)
  ops.py           ``torch.ops.ours_lib.*`` operator surface of the reference (lib/linear/__init__.py)
  linear/          nn.Module mirror of lib/linear/{tcq,comb,vq}_linear.py
  mem_op.py        quantizer-string grammar, Llama layer shapes, synthetic packed weights
                   (lib/utils/mem_op.py:2-307)
  hadamard.py      get_hadK (generated Paley factors), matmul_hadU*_cuda, the one-launch `rotate` (lib/utils/matmul_had.py)
  linear/incoherent_linear.py  IncoherentLinear / IncoherentMLP / IncoherentSdpaAttention (lib/linear/incoherent_linear.py)
  packers.py       pack_trellis / pack_qweight / pack_qweight_{sq,vq}_simt on the C-ABI's host-side encoders
  shard.py         row-sharding of packed layers across GPUs (torch.distributed / RCCL)

There is deliberately no CPU implementation here: the CPU restatement lives in /oracle and is test
infrastructure only.
"""
from . import _native  # noqa: F401
from . import ops  # noqa: F401
from . import mem_op  # noqa: F401
from . import shard  # noqa: F401
from . import hadamard  # noqa: F401
from . import packers  # noqa: F401
from .linear import (  # noqa: F401
    IncoherentLinear,
    IncoherentMLP,
    IncoherentSdpaAttention,
    make_linear,
    CombLinearTCQ,
    CombtLinearTCQ,
    QTIPLinearTCQ,
    VQLinearPackSIMT,
    VQLinearPackTensorCore,
    make_linear_from_info,
    multi_gemv,
    share_codebooks,
)

__version__ = "0.1.0"
