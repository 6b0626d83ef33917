"""Split-K sweep for the long-K TN (dW) shapes of the step, 64x64 tiles: slices vs time (the plan's choice is splitk=0)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gemm_bench import bench
from sdumc_amd import ops
for M, N, K in ((256, 1024, 24000), (256, 1024, 14400), (256, 256, 48000), (256, 256, 28800), (256, 4096, 2048)):
    for s in (0, 5, 8, 10, 12, 16, 20, 24, 32, 40, 48, 64, 80):
        if K // s < 256 if s else False:
            continue
        bench(ops.TN, M, N, K, tile=2, splitk=s)
    for s in (8, 16, 32):
        bench(ops.TN, M, N, K, tile=1, splitk=s)
