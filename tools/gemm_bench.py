"""GEMM micro-benchmark on the launch stream (HIP events): shapes of the SDUMC step + a square reference."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sdumc_amd import ops

def bench(layout, M, N, K, tile=0, splitk=0, reps=20, **kw):
    dev = "cuda"
    if layout == ops.NT: A, B = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
    elif layout == ops.NN: A, B = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev)
    else: A, B = torch.randn(K, M, device=dev), torch.randn(K, N, device=dev)
    C = torch.empty(M, N, device=dev)
    for _ in range(3): ops.gemm(layout, A, B, M, N, K, C_out=C, tile=tile, splitk=splitk, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): ops.gemm(layout, A, B, M, N, K, C_out=C, tile=tile, splitk=splitk, **kw)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print(f"{['NT','NN','TN'][layout]} M={M:6d} N={N:5d} K={K:6d} tile={tile} splitk={splitk:3d}: {us:8.1f} us  {2.0*M*N*K/us/1e6:7.1f} TF", flush=True)

if __name__ == "__main__":
    NT, NN, TN = ops.NT, ops.NN, ops.TN
    bench(NT, 4096, 4096, 4096, tile=1, splitk=1)
    bench(NT, 8192, 8192, 2048, tile=1, splitk=1)
    bench(NT, 4096, 4096, 4096, tile=2, splitk=1)
    for t, s in ((1, 1), (2, 1), (0, 0)):
        bench(NT, 24000, 256, 1024, tile=t, splitk=s)
    for t, s in ((2, 1), (1, 4), (1, 8), (1, 16), (0, 0)):
        bench(NT, 2048, 256, 4096, tile=t, splitk=s)
    for t, s in ((1, 1), (2, 1)):
        bench(NT, 48000, 256, 256, tile=t, splitk=s)
        bench(NN, 48000, 256, 256, tile=t, splitk=s)
    for s in (16, 24, 32, 0):
        bench(TN, 256, 1024, 24000, tile=1, splitk=s)
    for s in (48, 96, 0):
        bench(TN, 256, 256, 48000, tile=1, splitk=s)
    for t, s in ((2, 1), (3, 1), (0, 0)):
        bench(NT, 128, 256, 256, tile=t, splitk=s)
        bench(NT, 128, 256, 896, tile=t, splitk=s)
        bench(NT, 896, 256, 256, tile=t, splitk=s)
