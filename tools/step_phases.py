"""Wall time of the pieces of one train step, each fenced by a device synchronise (so launch ramp-up is included and
nothing overlaps across pieces): forward, loss + its backward, utterance-level backward (phase 0), frame-level backward
(phase 1), Adam.  The sum exceeds the fused step because the fused step has no fences."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sdumc_amd import engine
from sdumc_amd.trainer import DataParallelStep
dev = torch.device("cuda", 0)
batch = [t.to(dev) for t in bench.synthetic_shard(bench.B_PER_GPU, 0)]
flat, lay = bench.init_flat_params(engine, dev)
dp = DataParallelStep(flat, bench.B_PER_GPU, bench.T_MOSEI, bench.DIMS, seed=2024, exact=True); dp.set_batch(*batch)
be = dp.be
pieces = [("forward", be.forward), ("loss+backward of loss", be.loss_backward), ("backward utterance-level", lambda: be.backward_phase(0)),
          ("backward frame-level", lambda: be.backward_phase(1)), ("adam", lambda: be.adam(1.0))]
acc = {k: 0.0 for k, _ in pieces}
N = 50
for it in range(N + 10):
    for k, f in pieces:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        f()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        if it >= 10: acc[k] += dt
for k in acc: print(f"{k:28s} {acc[k] / N * 1e3:7.3f} ms")
print("sum", round(sum(acc.values()) / N * 1e3, 3))
