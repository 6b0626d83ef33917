"""PMC target: the keys-audio NT shape (M 48000, N = K = 256, mask + bias + tanh) and the 4096^3 reference on the wide
kernels, a few launches each (rocprofv3 --pmc ... -- python3 tools/gemm_wide_pmc_target.py <tile>)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sdumc_amd import ops, _lib
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 13
M, D = 48000, 256
x = torch.randn(24000, D, device="cuda"); W = torch.randn(D, D, device="cuda") / 16; b = torch.randn(D, device="cuda")
C = torch.empty(M, D, device="cuda")
d = _lib.make_dropout(True, 3, 0.5, M, D, 1, seed=77)
bits = ops.dropout_bits(d, 1)
A4 = torch.randn(4096, 4096, device="cuda"); B4 = torch.randn(4096, 4096, device="cuda"); C4 = torch.empty(4096, 4096, device="cuda")
xa = torch.randn(24000, 1024, device="cuda"); Wa = torch.randn(D, 1024, device="cuda") / 32; Ca = torch.empty(24000, D, device="cuda")
for _ in range(4):
    ops.gemm(ops.NT, x, W, M, D, D, bias=b, C_out=C, tile=tile, splitk=0, act=ops.ACT_TANH, a_row_mod=24000, a_drop=d, ab_drop_bits=[bits])
    ops.gemm(ops.NT, x, W, M, D, D, bias=b, C_out=C, tile=tile, splitk=0, a_row_mod=24000)
    ops.gemm(ops.NT, xa, Wa, 24000, D, 1024, bias=b, C_out=Ca, tile=tile, splitk=0)
    ops.gemm(ops.NT, A4, B4, 4096, 4096, 4096, C_out=C4, splitk=1, tile=tile)
torch.cuda.synchronize()
