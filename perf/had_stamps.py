#!/usr/bin/env python3
"""Phase stamps (s_memtime, 100 MHz) of the rotation kernel; needs the -DHAD_STAMPS build:
QPAL_LIB=perf/libqpal_stamps.so python perf/had_stamps.py"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import qpalette_amd as qp

had = qp.hadamard
lib = qp._native.lib()
dev = torch.device("cuda")
dbg = torch.zeros(16 * 8, dtype=torch.int64, device=dev)
lib.qpal_debug_had_stamps(ctypes.c_void_p(dbg.data_ptr()))
for n in (4096, 14336, 28672):
    hadK, K = had.get_hadK(n)
    h = None if hadK is None else hadK.T.contiguous().half().to(dev)
    x = torch.randn(1, n, device=dev).half()
    su = (torch.randint(0, 2, (n,), device=dev) * 2 - 1).half()
    out = torch.empty_like(x)
    for _ in range(3):
        dbg.zero_()
        had.rotate(x, hadK=h, K=K, su=su, post_scale=1 / 64, out=out)
        torch.cuda.synchronize()
    st = dbg.cpu().view(16, 8)
    nw = int((st[:, 0] > 0).sum())
    t0 = int(st[:nw, 0].min())
    print(f"n={n}: waves {nw}; per-wave stamps in us since first wave start "
          "(0 start, 1/2 pass0 done/after barrier, 3/4 pass1, 5/6 pass2, 7 end)")
    for w in range(nw):
        print("   wave", w, " ".join(f"{(int(v) - t0) / 100:6.2f}" if v > 0 else "   -  " for v in st[w]))
