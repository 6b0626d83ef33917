"""One shape for a counter pass: the 5-site rows launch (sdumc_gemm_rows256), forward form (masked + bias + tanh) or dX form
(accumulate), N launches.  usage: python3 tools/rows_pmc_target.py fwd|dx [launches]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from sdumc_amd import ops  # noqa: E402

dev = "cuda"
form = sys.argv[1] if len(sys.argv) > 1 else "fwd"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
g = torch.Generator(device=dev).manual_seed(1)
Ms = [48000, 28800, 28800, 4096, 4096]
W = [torch.randn(256, 256, device=dev, generator=g) / 16 for _ in range(4)]
bias = torch.randn(256, device=dev, generator=g)
A = [torch.randn(M, 256, device=dev, generator=g) for M in Ms]
C = [torch.randn(M, 256, device=dev, generator=g) for M in Ms]
if form == "fwd":
    bits = [torch.randint(0, 16, (M, 64), device=dev, generator=g, dtype=torch.uint8) for M in Ms]
    probs = [{"A": a, "B": W[i % 4], "C": c, "bits": b, "scale": 2.0, "bias": bias, "act": ops.ACT_TANH} for i, (a, c, b) in enumerate(zip(A, C, bits))]
else:
    probs = [{"A": a, "B": W[i % 4], "C": c, "accumulate": True} for i, (a, c) in enumerate(zip(A, C))]
for _ in range(n):
    ops.gemm_rows256(probs)
torch.cuda.synchronize()
