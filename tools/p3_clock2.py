"""in-kernel stamps of gemm_p3_nt in the step's form (tile_m = 0: 64-row tiles, 256-thread workgroups; library built with -DSDUMC_P3_DBG=8):
python tools/p3_clock2.py M K [mask]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdumc_amd import ops  # noqa: E402
from tools.p3_check import timeit  # noqa: E402

M, K = int(sys.argv[1]), int(sys.argv[2])
mask = len(sys.argv) > 3 and sys.argv[3] == "mask"
g = torch.Generator(device="cuda").manual_seed(1)
X, W, b = torch.randn(M, K, device="cuda", generator=g), torch.randn(256, K, device="cuda", generator=g) / K ** 0.5, torch.randn(256, device="cuda", generator=g)
X3, W3 = ops.p3_split(X), ops.p3_split_frag(W)
bits = torch.randint(0, 16, (M, K // 4), device="cuda", generator=g, dtype=torch.uint8) if mask else None
c = torch.empty(M, 256, device="cuda")
launch, _ = ops.gemm_p3_nt_call(X3, W3, M, 256, K, bias=b, tile_m=0, splitk=1, C_out=c, bits=bits, scale=2.0, act=ops.ACT_TANH if mask else ops.ACT_NONE)
t = timeit(launch, reps=100)
torch.cuda.synchronize()
st = c.view(torch.int32)[::64, :4].cpu().long() & 0xFFFFFFFF
cyc, real = st[:, 0].double(), st[:, 1].double()
mhz = cyc / real * 100.0
pro, epi = st[:, 2].double() / 100, st[:, 3].double() / 100
tiles = (M + 63) // 64
print(f"M={M} K={K} mask={mask}: launch {t:.1f} us, {tiles} tiles = {tiles / 512:.2f} rounds of 512 slots; per workgroup {float(real.mean()) / 100:.1f} us "
      f"(min {float(real.min()) / 100:.1f}, max {float(real.max()) / 100:.1f}), clock {float(mhz.mean()):.0f} MHz; prologue {float(pro.mean()):.2f} us (max {float(pro.max()):.2f}), "
      f"epilogue {float(epi.mean()):.2f} us (max {float(epi.max()):.2f}), loop {float((real / 100 - pro - epi).mean()):.2f} us")
