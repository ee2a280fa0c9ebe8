#!/usr/bin/env python3
"""Profiling helper: the SwiGLU rotation of a Llama-8B down_proj input (14336 = 28 x 512), eager calls (run under rocprofv3)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import qpalette_amd as qp


def main():
    dev = torch.device("cuda", 0)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 14336
    had = qp.hadamard
    hadK, K = had.get_hadK(n)
    hT = hadK.T.contiguous().half().to(dev)
    ug = torch.randn(1, 2 * n, device=dev)
    su = (torch.randint(0, 2, (n,), device=dev) * 2 - 1).half()
    for i in range(50):
        y = had.rotate(ug, hadK=hT, K=K, su=su, post_scale=1 / 64, in_mode=had.IN_SWIGLU_F32)
    torch.cuda.synchronize()
    print(float(y.float().abs().mean()))


if __name__ == "__main__":
    main()
