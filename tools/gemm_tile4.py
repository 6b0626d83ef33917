import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gemm_bench import bench
from sdumc_amd import ops
for layout in (ops.NT, ops.NN):
    for M, N, K in ((48000, 256, 256), (28800, 256, 256), (24000, 256, 1024), (14400, 256, 1024)):
        for tile in (2, 4):
            bench(layout, M, N, K, tile=tile, splitk=1)
for M, N, K, s in ((256, 256, 48000, 32), (256, 1024, 24000, 16)):
    for tile in (2, 4):
        bench(ops.TN, M, N, K, tile=tile, splitk=s)
