"""Step time at tiny batch sizes = the latency floor of the ~160-launch dependency chain."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sdumc_amd import _lib, engine
dev = torch.device("cuda", 0)
flat, lay = bench.init_flat_params(engine, dev)
for B, T in ((4, (8, 4, 8, 4)), (64, (8, 4, 8, 4)), (64, bench.T_MOSEI)):
    batch = [torch.randn(B, T[i], bench.DIMS[i], device=dev) for i in range(4)] + [torch.rand(B, device=dev)]
    for graph in (False, True):
        step = engine.TrainStep(flat.clone(), B, T, bench.DIMS, seed=1)
        step.set_batch(*batch)
        if graph: step.capture()
        for _ in range(10): step.run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        N = 50
        for _ in range(N): step.run()
        torch.cuda.synchronize()
        print(f"B={B} T={T} graph={graph}: {1e3*(time.perf_counter()-t0)/N:.3f} ms/step", flush=True)
