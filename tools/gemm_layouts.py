"""NT vs NN vs TN on the shapes of the step (plain operands): how much do the row-contiguous (ds_read_b32) layouts cost?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gemm_bench import bench
from sdumc_amd import ops
for M, N, K in ((48000, 256, 256), (28800, 256, 256), (24000, 256, 1024), (4096, 4096, 4096)):
    bench(ops.NT, M, N, K, tile=2, splitk=1)
    bench(ops.NN, M, N, K, tile=2, splitk=1)
for M, N, K, s in ((256, 256, 48000, 32), (256, 1024, 24000, 16), (256, 256, 28800, 32), (4096, 4096, 4096, 1)):
    bench(ops.TN, M, N, K, tile=2, splitk=s)
    bench(ops.NT, M, N, K, tile=2, splitk=s)
