"""Where the data-parallel step spends its time on ONE GPU with a one-rank RCCL communicator (the only RCCL available
on a 1-GPU box): host enqueue time and GPU step time for the fused TrainStep, the DP step without collectives, and the
DP step with each collective enabled in turn."""
import os, sys, time, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sdumc_amd import engine
from sdumc_amd.trainer import DataParallelStep

dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29561", rank=0, world_size=1, device_id=dev)


def timeit(run, n=100, w=20):
    for _ in range(w): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): run()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return round((t1 - t0) / n * 1e3, 3), round((t2 - t0) / n * 1e3, 3)


batch = [t.to(dev) for t in bench.synthetic_shard(bench.B_PER_GPU, 0)]
flat, lay = bench.init_flat_params(engine, dev)
ts = engine.TrainStep(flat, bench.B_PER_GPU, bench.T_MOSEI, bench.DIMS, seed=2024); ts.set_batch(*batch)
print("fused TrainStep                 host/gpu ms", timeit(ts.run), flush=True)
flat2, _ = bench.init_flat_params(engine, dev)
dp = DataParallelStep(flat2, bench.B_PER_GPU, bench.T_MOSEI, bench.DIMS, seed=2024, exact=True, force_collectives=True)
dp.set_batch(*batch)
be, B = dp.be, bench.B_PER_GPU
print("DP step, all collectives        host/gpu ms", timeit(dp.step), flush=True)
dp.collect = False; dp.overlap = False
print("DP step, no collectives         host/gpu ms", timeit(dp.step), flush=True)
dp.collect = True; dp.overlap = False
print("DP step, exchange + one flat AR host/gpu ms", timeit(dp.step), flush=True)


def grads_only(mode):
    def run():
        be.forward(); be.loss_backward()
        if mode == "flat":
            g = be.backward(); dist.all_reduce(g)
        elif mode == "split_sync":
            e = be.backward_phase(0); dist.all_reduce(e); l = be.backward_phase(1); dist.all_reduce(l)
        elif mode == "split_async":
            e = be.backward_phase(0); h = dist.all_reduce(e, async_op=True); l = be.backward_phase(1); dist.all_reduce(l); h.wait()
        elif mode == "phases":
            be.backward_phase(0); be.backward_phase(1)
        else:
            be.backward()
        be.adam(1.0)
    return run


for m in ("none", "phases", "flat", "split_sync", "split_async"):
    print(f"local loss, grads {m:12s}     host/gpu ms", timeit(grads_only(m)), flush=True)

x = torch.zeros(lay.live, device=dev)
print("bare all_reduce 15 MB           host/gpu ms", timeit(lambda: dist.all_reduce(x)), flush=True)
y = torch.zeros(9000, device=dev)
print("bare all_reduce 36 KB           host/gpu ms", timeit(lambda: dist.all_reduce(y)), flush=True)
dist.destroy_process_group()
