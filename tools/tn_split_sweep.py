"""dW-shaped TN GEMM (M = 256, N = 1024, K = 24000: frame_dim_reshape_0's weight gradient at C2) over tile shapes and split-K
factors: where the 85-94 TF of the step's dominant kernel come from when the same loop reaches 121-124 TF on shapes with enough
output tiles."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sdumc_amd import ops
from gemm_wide_check import timeit

dev = "cuda"
M, N, K = 256, 1024, 24000
g = torch.Generator(device=dev).manual_seed(1)
A = torch.randn(K, M, device=dev, generator=g)
B = torch.randn(K, N, device=dev, generator=g)
C = torch.empty(M, N, device=dev)
ref = None
for tile in (0, 1, 2, 11, 13, 14):
    line = []
    for sk in (0, 2, 4, 8, 16, 32, 64):
        try:
            fn = lambda: ops.gemm(ops.TN, A, B, M, N, K, C_out=C, tile=tile, splitk=sk)
            fn()
            torch.cuda.synchronize()
            if ref is None:
                ref = C.clone()
            err = float((C - ref).abs().max() / ref.abs().max())
            us = timeit(fn, reps=10, rounds=2)
            line.append(f"sk={sk:2d}: {us:6.1f}us {2.0 * M * N * K / us / 1e6:5.1f}TF" + ("" if err < 1e-5 else f" ERR {err:.1e}"))
        except Exception as e:      # a combination the plan refuses
            line.append(f"sk={sk:2d}: -")
    print(f"tile {tile:2d}: " + " | ".join(line), flush=True)
