"""Host cost of making a per-shape arena step object (FusedTrainer._get on a new shape) and of an epoch with / without the objects cached."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from sdumc_amd import engine  # noqa: E402
from sdumc_amd.data import DeviceFeatureStore  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
B, T, DIMS = 64, bench.T_MOSEI, bench.DIMS
flat, lay = bench.init_flat_params(engine, dev)
store = DeviceFeatureStore.synthetic(2048, T, DIMS, seed=1234, device=dev, planes=True)
g = torch.Generator().manual_seed(7)
for nb in (100, 200):
    batches = [torch.randperm(len(store), generator=g)[:B] for _ in range(nb + 10)]
    tr = engine.FusedTrainer(flat.clone(), DIMS, capacity=(B, T), seed=2024)
    pw, pt = store.plan_epoch(batches[:10]), store.plan_epoch(batches[10:])
    tr.run_epoch(store, pw)
    torch.cuda.synchronize()
    n0 = len(tr._steps)
    t0 = time.perf_counter()
    tr.run_epoch(store, pt)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    new = len(tr._steps) - n0
    print(f"{nb} batches, {new} new shapes: host {1e3 * (t1 - t0) / nb:.3f} ms/step, wall {1e3 * (t2 - t0) / nb:.3f} ms/step")
    t0 = time.perf_counter()
    tr.run_epoch(store, pt)      # the same epoch again: every shape cached
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"   again (cached): host {1e3 * (t1 - t0) / nb:.3f} ms/step, wall {1e3 * (t2 - t0) / nb:.3f} ms/step")
# the creation alone
tr = engine.FusedTrainer(flat.clone(), DIMS, capacity=(B, T), seed=2024)
shapes = [(64, (375 - i, 32, 225 - (i % 7), 32 - (i % 3))) for i in range(50)]
t0 = time.perf_counter()
for s in shapes:
    tr._get(*s)
print(f"FusedTrainer._get on a new shape: {1e3 * (time.perf_counter() - t0) / len(shapes):.3f} ms each")
import cProfile, pstats
shapes = [(64, (300 - i, 32, 225 - (i % 7), 32 - (i % 3))) for i in range(50)]
pr = cProfile.Profile(); pr.enable()
for s in shapes:
    tr._get(*s)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
