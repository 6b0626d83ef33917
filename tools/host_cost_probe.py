"""Pure host cost per step (enqueue of short bursts behind a synchronised device: no queue back-pressure) of the static resident step
and of the in-place epoch loop, with the C call's share.  python tools/host_cost_probe.py"""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from sdumc_amd import engine, _lib  # noqa: E402
from sdumc_amd.data import DeviceFeatureStore  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
B, T, DIMS = 64, bench.T_MOSEI, bench.DIMS
flat, lay = bench.init_flat_params(engine, dev)
print("cpu affinity:", len(os.sched_getaffinity(0)), "cpus; this process on cpu", os.sched_getcpu() if hasattr(os, "sched_getcpu") else "?")
# static
batches = [[t.to(dev) for t in bench.synthetic_shard(B, 0, k=k)] for k in range(2)]
ts, run = bench.resident_step(engine, flat.clone(), batches)
for _ in range(10):
    run()
torch.cuda.synchronize()
acc = []
for _ in range(6):
    t0 = time.perf_counter()
    for _ in range(8):
        run()
    acc.append((time.perf_counter() - t0) / 8)
    torch.cuda.synchronize()
print("static resident step: host %.3f ms/step (bursts of 8: %s)" % (1e3 * min(acc), ", ".join("%.3f" % (1e3 * a) for a in acc)))
# the C call alone
io, dims, cfg = ts.io, ts.dims, ts.cfg
st = _lib.current_stream()
acc = []
for _ in range(6):
    t0 = time.perf_counter()
    for _ in range(8):
        _lib.lib.sdumc_train_step(C.byref(dims), C.byref(io), C.byref(cfg), st)
    acc.append((time.perf_counter() - t0) / 8)
    torch.cuda.synchronize()
print("  sdumc_train_step alone: %.3f ms/call" % (1e3 * min(acc)))
# epoch, in place
store = DeviceFeatureStore.synthetic(2048, T, DIMS, seed=1234, device=dev, planes=True)
g = torch.Generator().manual_seed(7)
tr = engine.FusedTrainer(flat.clone(), DIMS, capacity=(B, T), seed=2024)
plans = [store.plan_epoch([torch.randperm(len(store), generator=g)[:B] for _ in range(8)]) for _ in range(8)]
for p in plans[:2]:
    tr.run_epoch(store, p)
torch.cuda.synchronize()
acc = []
for p in plans[2:]:
    t0 = time.perf_counter()
    tr.run_epoch(store, p)
    acc.append((time.perf_counter() - t0) / 8)
    torch.cuda.synchronize()
print("in-place epoch loop: host %.3f ms/step (bursts of 8: %s), %d step objects" % (1e3 * min(acc), ", ".join("%.3f" % (1e3 * a) for a in acc), len(tr._steps)))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
tr.run_epoch(store, plans[3])
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(8)
