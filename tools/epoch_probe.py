"""Where a ragged epoch's time goes: python tools/epoch_probe.py [--bf16] [--profile] [--mode epoch|one|static]
host time per step (enqueue) vs wall time per step, optionally under cProfile."""
import argparse
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from sdumc_amd import engine  # noqa: E402
from sdumc_amd.data import DeviceFeatureStore  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--bf16", action="store_true")
ap.add_argument("--profile", action="store_true")
ap.add_argument("--mode", default="epoch")
ap.add_argument("--nb", type=int, default=200)
ap.add_argument("--wgs", type=int, default=0)
ap.add_argument("--fixed", action="store_true", help="every utterance at full length: one batch shape")
ap.add_argument("--gather", action="store_true", help="padded copies instead of in-place row maps")
args = ap.parse_args()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
B, T, DIMS = 64, bench.T_MOSEI, bench.DIMS
flat, lay = bench.init_flat_params(engine, dev)
hf = engine.bf16_mode(args.bf16, DIMS) == 2
store = DeviceFeatureStore.synthetic(2048, T, DIMS, seed=1234, device=dev, bf16=hf, planes=not hf, min_frac=1.0 if args.fixed else 0.25)
g = torch.Generator().manual_seed(7)
batches = [torch.randperm(len(store), generator=g)[:B] for _ in range(args.nb + 10)]
tr = engine.FusedTrainer(flat, DIMS, capacity=(B, T), seed=2024, bf16=args.bf16, prefetch_workgroups=args.wgs, inplace=not args.gather)
plan_w, plan_t = store.plan_epoch(batches[:10]), store.plan_epoch(batches[10:])
# every shape's step object up front (host set-up is not what is being measured)
for (b, t) in plan_w.shapes + plan_t.shapes:
    tr._get(b, t)


def epoch(plan):
    if args.mode == "epoch":
        tr.run_epoch(store, plan)
    else:
        for i in range(len(plan)):
            o = plan.offsets[i]
            tr.step_from_store(store, plan.idx_d[o:o + plan.shapes[i][0]].cpu())


epoch(plan_w)
torch.cuda.synchronize()
pr = cProfile.Profile() if args.profile else None
t0 = time.perf_counter()
if pr:
    pr.enable()
epoch(plan_t)
if pr:
    pr.disable()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
n = len(plan_t)
print(f"mode {args.mode} gather {args.gather} bf16 {args.bf16} fixed {args.fixed} PF_MODE {os.environ.get('SDUMC_PF_MODE', '1')} wgs {args.wgs}: "
      f"host {1e3 * (t1 - t0) / n:.4f} ms/step, wall {1e3 * (t2 - t0) / n:.4f} ms/step, loss {float(tr.state.losses[0]):.5f}")
if pr:
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
