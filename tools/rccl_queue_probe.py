"""Does creating the RCCL communicator slow the three-lane step down (hardware-queue sharing)?
usage: rccl_queue_probe.py [--no-dist] [--lanes-first]   (env GPU_MAX_HW_QUEUES is read by the HIP runtime)"""
import os, sys, time, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sdumc_amd import engine
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
if "--dist-first" in sys.argv:
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29562", rank=0, world_size=1, device_id=dev)
elif "--dist-first-lazy" in sys.argv:
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29562", rank=0, world_size=1)
batch = [t.to(dev) for t in bench.synthetic_shard(bench.B_PER_GPU, 0)]
flat, lay = bench.init_flat_params(engine, dev)
ts = engine.TrainStep(flat, bench.B_PER_GPU, bench.T_MOSEI, bench.DIMS, seed=2024); ts.set_batch(*batch)
if "--lanes-first" in sys.argv:
    ts.run(); torch.cuda.synchronize()
print("workspace ptr %x params %x audio %x" % (ts.workspace.data_ptr(), flat.data_ptr(), ts.audio.data_ptr()))
if "--no-dist" not in sys.argv and not dist.is_initialized():
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29562", rank=0, world_size=1, device_id=dev)
    if "--use-comm" in sys.argv:
        x = torch.zeros(1024, device=dev); dist.all_reduce(x); torch.cuda.synchronize()
for _ in range(20): ts.run()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(100): ts.run()
torch.cuda.synchronize()
print("ARGS", sys.argv[1:], "GPU_MAX_HW_QUEUES", os.environ.get("GPU_MAX_HW_QUEUES"), "ms/step", round((time.perf_counter() - t0) * 10, 4), flush=True)
if dist.is_initialized():
    dist.destroy_process_group()
