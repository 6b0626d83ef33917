"""64x64 vs 128x128 tiles on the wide shapes of the generic transformer layer (E = 1024, 4E = 4096, M = T*B = 16384)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gemm_bench import bench
from sdumc_amd import ops
for layout in (ops.NT, ops.NN, ops.TN):
    for M, N, K in ((16384, 1024, 1024), (16384, 4096, 1024), (16384, 1024, 4096), (1024, 1024, 16384), (4096, 1024, 16384), (1024, 4096, 16384), (4096, 512, 512), (2048, 1024, 1024)):
        if layout != ops.TN and K == 16384: continue
        if layout == ops.TN and K != 16384: continue
        for tile in (2, 1):
            bench(layout, M, N, K, tile=tile, splitk=0)
