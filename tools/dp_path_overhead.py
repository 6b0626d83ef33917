"""Single-GPU cost of the data-parallel code path (DataParallelStep with world_size 1: separate forward / loss /
backward / Adam calls from Python, no collectives) against the fused TrainStep the N = 1 bench line uses."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sdumc_amd import engine
from sdumc_amd.trainer import DataParallelStep
dev = torch.device("cuda", 0)
def timeit(run, n=100, w=20):
    for _ in range(w): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): run()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
batch = [t.to(dev) for t in bench.synthetic_shard(bench.B_PER_GPU, 0)]
flat, lay = bench.init_flat_params(engine, dev)
ts = engine.TrainStep(flat, bench.B_PER_GPU, bench.T_MOSEI, bench.DIMS, seed=2024); ts.set_batch(*batch)
print("fused TrainStep        ms/step", round(timeit(ts.run), 4))
flat2, _ = bench.init_flat_params(engine, dev)
dp = DataParallelStep(flat2, bench.B_PER_GPU, bench.T_MOSEI, bench.DIMS, seed=2024, exact=True); dp.set_batch(*batch)
print("DataParallelStep (W=1) ms/step", round(timeit(dp.step), 4))
be = dp.be
def phased():
    be.forward(); be.loss_backward(); be.backward_phase(0); be.backward_phase(1); be.adam(1.0)
print("same with backward in two phases", round(timeit(phased), 4))
