"""launch time of sdumc_gemm_p3_nt on one shape: python tools/p3_time.py M K tile_m [splitk] [mask]   (SDUMC_P3_DBG ablations)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdumc_amd import ops, _lib  # noqa: E402
from tools.p3_check import timeit  # noqa: E402

M, K, tm = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
sk = int(sys.argv[4]) if len(sys.argv) > 4 else 1
mask = len(sys.argv) > 5 and sys.argv[5] == "mask"
g = torch.Generator(device="cuda").manual_seed(1)
X, W, b = torch.randn(M, K, device="cuda", generator=g), torch.randn(256, K, device="cuda", generator=g) / K ** 0.5, torch.randn(256, device="cuda", generator=g)
X3, W3 = ops.p3_split(X), ops.p3_split_frag(W)
bits = None
if mask:
    bits = torch.randint(0, 16, (M, K // 4), device="cuda", generator=g, dtype=torch.uint8)
c = torch.empty(M, 256, device="cuda")
launch, _ = ops.gemm_p3_nt_call(X3, W3, M, 256, K, bias=b, tile_m=tm, splitk=sk, C_out=c, bits=bits, scale=2.0, act=ops.ACT_TANH if mask else ops.ACT_NONE)
t = timeit(launch, reps=100)
print(f"M={M} K={K} tile_m={tm} splitk={sk} mask={mask} dbg={os.environ.get('SDUMC_P3_DBG', '0')}: {t:7.1f} us = {2.0 * M * 256 * K / t / 1e6:6.1f} TF")
