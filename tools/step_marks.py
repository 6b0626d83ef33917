"""Where the wall time of one C2 train step goes WITHOUT a profiler attached: the engine's debug marks (events on the caller's
stream, include/sdumc_hip.h sdumc_debug_marks) averaged over a few steps.  usage: python tools/step_marks.py [--bf16]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sdumc_amd import engine, _lib

bf16 = "--bf16" in sys.argv
dev = torch.device("cuda:0")
flat, lay = bench.init_flat_params(engine, dev)
# the headline's loop: bench.N_RESIDENT resident batches in turn, the next step's keep-bits generated in this step's middle
batches = [[t.to(dev) for t in bench.synthetic_shard(bench.B_PER_GPU, 0, k=k)] for k in range(bench.N_RESIDENT)]
step, run = bench.resident_step(engine, flat, batches, bf16=bf16)
step.run = run
for _ in range(12):
    step.run()
torch.cuda.synchronize()
_lib.lib.sdumc_debug_marks(1)
names = ["start", "frame-level fwd done (chain A launch)", "chain fwd A done", "site-1 attention done (chain B launch)",
         "chain fwd B done", "losses done", "chain bwd B done", "site-1 pooling bwd done (chain bwd A launch)",
         "chain bwd A done", "frame-level bwd done (before Adam)", "Adam done"]
acc = [0.0] * 11
N = 20
for _ in range(N):
    step.run()
    torch.cuda.synchronize()
    ms = (C.c_float * 11)()
    _lib.lib.sdumc_debug_marks_read(ms, 11)
    for i in range(11):
        acc[i] += ms[i]
ms2 = [0.0] * 48
for _ in range(N):
    step.run()
    torch.cuda.synchronize()
    buf = (C.c_float * 48)()
    _lib.lib.sdumc_debug_marks_read(buf, 48)
    for i in range(48):
        ms2[i] += buf[i]
prev = 0.0
for i, n in enumerate(names):
    t = acc[i] / N * 1e3
    print(f"{i:2d} {t:8.1f} us  (+{t - prev:7.1f})  {n}")
    prev = t

for m, mod in enumerate(("audio", "text", "video")):
    print(f"  {mod} lane (frame-level forward): projection done {ms2[28 + 4 * m] / N * 1e3:7.1f}  keep-bits awaited {ms2[29 + 4 * m] / N * 1e3:7.1f}"
          f"  FRA2UTT site done {ms2[30 + 4 * m] / N * 1e3:7.1f}")
print(f"  lane 3 (keep-bits, Cross_Attention key projections) done at {ms2[40] / N * 1e3:7.1f} us")
lane_names = ["pooling bwd + colsum done", "key-projection dX done", "early Cross_Attention dxd awaited", "mask-sum done", "frame dW done"]
for m, mod in enumerate(("audio", "text", "video")):
    print(f"  {mod} lane (frame-level backward):", "  ".join(f"{lane_names[j]} {ms2[12 + 5 * m + j] / N * 1e3:7.1f}" for j in range(5)))
print(f"  lane 3 drained at {ms2[27] / N * 1e3:7.1f} us")

# phase stamps of workgroup 0 inside the clustered utterance-level kernels (chain_cluster.hip), when they are in use
try:
    _lib.lib.sdumc_chain_cluster_trace_.argtypes = [C.c_int, C.POINTER(C.c_double)]
    _lib.lib.sdumc_chain_cluster_trace_(1, None)
    step.run()
    torch.cuda.synchronize()
    out = (C.c_double * 128)()
    _lib.lib.sdumc_chain_cluster_trace_(0, out)
    for k, nm in enumerate(("fwd A", "fwd B", "bwd B", "bwd A")):
        ts = [(i, out[k * 32 + i]) for i in range(32) if out[k * 32 + i] >= 0]
        print(nm, " ".join(f"{i}:{t:.1f}" for i, t in ts))
except AttributeError:
    pass
