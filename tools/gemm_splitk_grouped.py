"""Split-K sweep for the GROUPED key-projection dW launches of the step (2 sites: groups = 2, M = N = 256, long K)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdumc_amd import ops
def bench(K, s, groups=2, reps=30):
    A = [torch.randn(K, 256, device="cuda") for _ in range(groups)]
    B = [torch.randn(K, 256, device="cuda") for _ in range(groups)]
    C = [torch.empty(256, 256, device="cuda") for _ in range(groups)]
    for _ in range(5): ops.gemm(ops.TN, A, B, 256, 256, K, C_out=C, tile=2, splitk=s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): ops.gemm(ops.TN, A, B, 256, 256, K, C_out=C, tile=2, splitk=s)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print(f"TN grouped x{groups} M=N=256 K={K:6d} splitk={s:3d}: {us:7.1f} us {2.0*groups*256*256*K/us/1e6:6.1f} TF", flush=True)
bench(48000, 16); bench(48000, 16)
for K in (48000, 28800, 4096):
    for s in (0, 4, 8, 12, 16, 20, 24, 32, 40, 48, 64):
        if s and K // s < 256: continue
        bench(K, s)
