"""In-kernel timeline of the grouped GEMM (SDUMC_GG_DBG=5 build path): s_memrealtime stamps per workgroup.
usage: SDUMC_GG_DBG=5 python3 tools/gg_stamps.py [leg]"""
import ctypes as C
import os
import sys
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdumc_amd import ops, _lib  # noqa: E402

dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
leg = sys.argv[1] if len(sys.argv) > 1 else "p1"
B, Ta, Tt, Tv, D = 64, 375, 32, 225, 256


def rn(*s):
    return torch.randn(*s, device=dev, generator=g)


if leg == "p1":
    ps = [{"A": rn(24000, 256), "B": rn(24000, 1024)}]
else:
    ps = [{"A": rn(B * Ta, D), "B": rn(B * Ta, 1024), "colsum": torch.zeros(D, device=dev)},
          {"A": rn(B * Tv, D), "B": rn(B * Tv, 1024), "colsum": torch.zeros(D, device=dev)},
          {"A": rn(B * Tt, D), "B": rn(B * Tt, 4096), "A1": rn(B * Tt, D), "B1": rn(B * Tt, 4096), "colsum": torch.zeros(D, device=dev)}]
n = len(ps)
arr = (_lib.GGProblem * n)()
for i, q in enumerate(ps):
    q["C"] = torch.empty(q["A"].shape[1], q["B"].shape[1], device=dev)
need = None
ws = torch.zeros(600 << 20, dtype=torch.uint8, device=dev)
for _ in range(5):
    ops.gemm_group_tn(ps, workspace=ws)
torch.cuda.synchronize()
# locate the stamps: after (nwg + units) slots
SLOT = 256 * 128 + 256
BN, BK = 128, 16
units = sum(((q["B"].shape[1] + BN - 1) // BN) for q in ps)
nwg = 256
off = (nwg + units) * SLOT * 4
st = ws[off: off + nwg * 96].view(torch.int64).cpu().numpy().reshape(nwg, 12)
# stamps: (realtime, shader clock) pairs at: start, [first stage landed, loop done] per piece
rt, sc = st[:, 0::2].astype(np.float64), st[:, 1::2].astype(np.float64)
np.set_printoptions(linewidth=250, precision=1, suppress=True)
for w in (0, 1, 100, 255):
    print(w, "realtime us", (rt[w] - rt[:, 0].min()) * 0.01, " clock GHz over first loop %.3f" % ((sc[w, 2] - sc[w, 1]) / ((rt[w, 2] - rt[w, 1]) * 10.0) ))
ghz = (sc[:, 2] - sc[:, 1]) / ((rt[:, 2] - rt[:, 1]) * 10.0)
print("in-loop clock: mean %.3f GHz min %.3f max %.3f" % (ghz.mean(), ghz.min(), ghz.max()))
print("first loop: mean %.1f us = %.0f cycles" % (((rt[:, 2] - rt[:, 1]) * 0.01).mean(), (sc[:, 2] - sc[:, 1]).mean()))
