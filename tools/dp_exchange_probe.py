"""Cost of the pieces of the DP exactness exchange on one GPU (one-rank RCCL communicator)."""
import os, sys, time, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sdumc_amd import engine
from sdumc_amd.trainer import DataParallelStep
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29563", rank=0, world_size=1, device_id=dev)
def timeit(run, n=200, w=30):
    for _ in range(w): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): run()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return round((t1 - t0) / n * 1e3, 3), round((t2 - t0) / n * 1e3, 3)
batch = [t.to(dev) for t in bench.synthetic_shard(bench.B_PER_GPU, 0)]
flat, lay = bench.init_flat_params(engine, dev)
dp = DataParallelStep(flat, bench.B_PER_GPU, bench.T_MOSEI, bench.DIMS, seed=2024, exact=True, force_collectives=True)
dp.set_batch(*batch)
be, B = dp.be, bench.B_PER_GPU
recs = torch.empty(1, 2 * B * 64 + B + 3, device=dev)
def variant(mode):
    def run():
        be.forward()
        if mode == "local":
            be.loss_backward()
        else:
            rec = be.dp_pack()
            if mode == "pack+gather": dist.all_gather_into_tensor(recs, rec)
            elif mode == "pack+allreduce": recs[0].copy_(rec); dist.all_reduce(recs)
            else: recs[0].copy_(rec)
            ssd, feats, labels2 = be.dp_unpack(recs, 1)
            be.loss_backward(ssd, feats, labels2, (0, B))
        be.backward(); be.adam(1.0)
    return run
for m in ("local", "pack+copy", "pack+gather", "pack+allreduce", "local"):
    print(f"{m:16s} host/gpu ms", timeit(variant(m)), flush=True)
x = torch.zeros(8259, device=dev); y = torch.zeros(1, 8259, device=dev)
def chain(k, coll):
    def run():
        for _ in range(k):
            x.add_(1.0)
            if coll: dist.all_gather_into_tensor(y, x)
            y.add_(1.0)
    return run
print("20 x (add, add)            ", timeit(chain(20, False)), flush=True)
print("20 x (add, all_gather, add)", timeit(chain(20, True)), flush=True)
dist.destroy_process_group()
