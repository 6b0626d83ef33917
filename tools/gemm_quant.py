"""Wave-quantisation probe: 64x64-tile NT GEMM, N=256, K in {256,1024}, M swept so that the tile count crosses
multiples of the 1024 resident workgroup slots (256 CUs x 4)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gemm_bench import bench
from sdumc_amd import ops
for K in (1024, 256):
    for tiles in (512, 768, 1024, 1280, 1500, 1792, 2048, 2560, 3000, 3072, 4096):
        M = tiles * 16
        bench(ops.NT, M, 256, K, tile=2, splitk=1)
    bench(ops.NT, 24000, 256, K, tile=2, splitk=2)
