"""How long does the host need to ISSUE one step (no sync) vs how long the GPU needs to run it?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sdumc_amd import _lib, engine
dev = torch.device("cuda", 0)
flat, lay = bench.init_flat_params(engine, dev)
batch = [t.to(dev) for t in bench.synthetic_shard(bench.B_PER_GPU, 0)]
step = engine.TrainStep(flat, bench.B_PER_GPU, bench.T_MOSEI, bench.DIMS, seed=1)
step.set_batch(*batch)
for conc in (1, 0):
    _lib.lib.sdumc_set_concurrency(conc)
    for _ in range(5): step.launch()
    torch.cuda.synchronize()
    N = 20
    t0 = time.perf_counter()
    for _ in range(N): step.launch()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"concurrency={conc}: host issue {1e3*(t1-t0)/N:.3f} ms/step, total {1e3*(t2-t0)/N:.3f} ms/step", flush=True)
