"""One-off probe: how many torch threads make the CPU oracle's train step fastest on this host."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import sdumc_oracle as O
DIMS, T, B = (1024, 4096, 1024, 4096), (375, 32, 225, 32), 64
P = O.init_params(DIMS, seed=0)
batch = O.synthetic_batch(B, T, DIMS, seed=1234)
for n in [int(a) for a in sys.argv[1:]] or [8, 16, 32, 64]:
    torch.set_num_threads(n)
    st = {}
    t0 = time.perf_counter(); O.train_step(P, st, *batch, mode="native", step=0); t1 = time.perf_counter()
    O.train_step(P, st, *batch, mode="native", step=1); t2 = time.perf_counter()
    print(f"threads={n} first={t1-t0:.2f}s second={t2-t1:.2f}s -> {B/(t2-t1):.1f} samples/s", flush=True)
    if t2 - t1 > 30: break
