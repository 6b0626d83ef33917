import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sdumc_amd import ops
M, D = 48000, 256
x2 = torch.randn(M, D, device="cuda"); W = torch.randn(D, D, device="cuda") / 16; C = torch.empty(M, D, device="cuda")
A4 = torch.randn(4096, 4096, device="cuda"); B4 = torch.randn(4096, 4096, device="cuda"); C4 = torch.empty(4096, 4096, device="cuda")
for _ in range(5):
    ops.gemm(ops.NT, x2, W, M, D, D, C_out=C, splitk=1)
    ops.gemm(ops.NT, A4, B4, 4096, 4096, 4096, C_out=C4, splitk=1, tile=2)
torch.cuda.synchronize()
