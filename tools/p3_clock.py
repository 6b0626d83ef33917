"""in-kernel clock stamps of gemm_p3_nt (a library built with -DSDUMC_P3_DBG=8 [+ ablation bits]): python tools/p3_clock.py M K tile_m"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdumc_amd import ops  # noqa: E402
from tools.p3_check import timeit  # noqa: E402

M, K, tm = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
g = torch.Generator(device="cuda").manual_seed(1)
X, W, b = torch.randn(M, K, device="cuda", generator=g), torch.randn(256, K, device="cuda", generator=g) / K ** 0.5, torch.randn(256, device="cuda", generator=g)
X3, W3 = ops.p3_split(X), ops.p3_split_frag(W)
c = torch.empty(M, 256, device="cuda")
launch, _ = ops.gemm_p3_nt_call(X3, W3, M, 256, K, bias=b, tile_m=tm, splitk=1, C_out=c)
t = timeit(launch, reps=100)
torch.cuda.synchronize()
st = c.view(torch.int32)[::tm, :4].cpu().long() & 0xFFFFFFFF
cyc, real = st[:, 0].double(), st[:, 1].double()
mhz = cyc / real * 100.0
pro, epi = st[:, 2].double() / 100, st[:, 3].double() / 100
print(f"M={M} K={K} tile_m={tm}: launch {t:.1f} us; per workgroup: {float(real.mean()) / 100:.1f} us (min {float(real.min()) / 100:.1f}, max {float(real.max()) / 100:.1f}), "
      f"shader clock {float(mhz.mean()):.0f} MHz (min {float(mhz.min()):.0f}, max {float(mhz.max()):.0f}); prologue {float(pro.mean()):.2f} us (max {float(pro.max()):.2f}), "
      f"epilogue {float(epi.mean()):.2f} us (max {float(epi.max()):.2f})")
