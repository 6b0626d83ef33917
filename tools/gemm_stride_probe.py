"""Does a power-of-two leading dimension cost the fp32 GEMM anything (L2 / HBM channel aliasing)?  Same shapes with
lda/ldb = K and K + 64."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sdumc_amd import ops

def bench(M, N, K, pad, tile, reps=20):
    ld = K + pad
    A = torch.randn(M, ld, device="cuda"); B = torch.randn(N, ld, device="cuda"); C = torch.empty(M, N, device="cuda")
    for _ in range(3): ops.gemm(ops.NT, A, B, M, N, K, C_out=C, lda=ld, ldb=ld, tile=tile, splitk=1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): ops.gemm(ops.NT, A, B, M, N, K, C_out=C, lda=ld, ldb=ld, tile=tile, splitk=1)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print(f"M={M:6d} N={N:5d} K={K:5d} ld=K+{pad:<3d} tile={tile}: {us:8.1f} us {2.0*M*N*K/us/1e6:7.1f} TF", flush=True)

for M, N, K in ((4096, 4096, 4096), (65536, 256, 1024), (24000, 256, 1024), (48000, 256, 256)):
    for pad in (0, 64, 32, 16):
        bench(M, N, K, pad, 2)
bench(4096, 4096, 4096, 0, 1); bench(4096, 4096, 4096, 64, 1)
