"""Standalone rate of sdumc_gemm_rows256 on the C2 shapes against the kernels it replaces (run on the GPU box).
The shapes are the step's (text M = 2 x 64 x 32 = 4096), and every timed launch works on the next of SETS independent operand
sets (together > 256 MB: nothing stays in the Infinity Cache between launches) -- round 3's version re-ran one 100-150 MB working
set and used text M = 6400, so its 101-113 TF were not the step's number.
usage: python tools/rows_bench.py"""
import sys, time
sys.path.insert(0, ".")
import torch
from sdumc_amd import ops

torch.manual_seed(0)
dev = "cuda"


def timeit(fn, reps=30):
    t0 = time.time()
    while time.time() - t0 < 0.5:       # clocks up
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def bits_of(M):
    return torch.randint(0, 16, (M, 64), dtype=torch.uint8, device=dev)


W = [torch.randn(256, 256, device=dev) / 16 for _ in range(4)]
bias = torch.randn(256, device=dev)
class Rot:
    """calls fn(set) on the next operand set each time"""

    def __init__(self, sets, fn):
        self.sets, self.fn, self.i = sets, fn, 0

    def __call__(self):
        self.fn(self.sets[self.i % len(self.sets)])
        self.i += 1


for name, Ms in (("audio site", [48000]), ("video both sites", [28800, 28800]), ("text both", [4096, 4096]),
                 ("audio+video+text dX (5 sites)", [48000, 28800, 28800, 4096, 4096])):
    fl = sum(2.0 * M * 65536 for M in Ms)
    nsets = max(4, int(300e6 / (sum(Ms) * 2048)) + 1)
    sets = []
    for _ in range(nsets):
        A = [torch.randn(M, 256, device=dev) for M in Ms]
        C = [torch.randn(M, 256, device=dev) for M in Ms]
        bts = [bits_of(M) for M in Ms]
        sets.append({"A": A, "C": C,
                     "dx": [{"A": a, "B": W[i % 4], "C": c, "accumulate": True} for i, (a, c) in enumerate(zip(A, C))],
                     "fwd": [{"A": a, "B": W[i % 4], "C": c, "bits": b, "scale": 2.0, "bias": bias, "act": ops.ACT_TANH}
                             for i, (a, c, b) in enumerate(zip(A, C, bts))]})
    t = timeit(Rot(sets, lambda st: ops.gemm_rows256(st["dx"])))
    # the 64x64 NN kernel (one launch per site here; the engine groups sites of equal M)
    def old(st):
        for i, (a, c) in enumerate(zip(st["A"], st["C"])):
            ops.gemm(ops.NN, a, W[i % 4], a.shape[0], 256, 256, C_out=c, accumulate=True)
    t_old = timeit(Rot(sets, old))
    # forward: masked + bias + tanh
    t_f = timeit(Rot(sets, lambda st: ops.gemm_rows256(st["fwd"])))
    del sets
    print(f"{name:32s} {fl / 1e9:6.2f} GF  dX rows {t:7.1f} us {fl / t / 1e6:6.1f} TF | dX 64x64 NN {t_old:7.1f} us {fl / t_old / 1e6:6.1f} TF"
          f" | fwd masked+tanh {t_f:7.1f} us {fl / t_f / 1e6:6.1f} TF")

# ---- bf16 storage ----
print("bf16 storage:")
Wh = [w.to(torch.bfloat16) for w in W]
for name, Ms in (("audio site", [48000]), ("video both sites", [28800, 28800]), ("text both", [4096, 4096]),
                 ("audio+video+text (5 sites)", [48000, 28800, 28800, 4096, 4096])):
    fl = sum(2.0 * M * 65536 for M in Ms)
    A = [torch.randn(M, 256, device=dev).to(torch.bfloat16) for M in Ms]
    C = [torch.randn(M, 256, device=dev).to(torch.bfloat16) for M in Ms]
    probs = [{"A": a, "B": Wh[i % 4], "C": c, "accumulate": True} for i, (a, c) in enumerate(zip(A, C))]
    t = timeit(lambda: ops.gemm_rows256(probs))
    by = sum(M * 512 * 3 for M in Ms)
    def old():
        for i, (a, c) in enumerate(zip(A, C)):
            ops.gemm_bf16(ops.NT, a, Wh[i % 4], a.shape[0], 256, 256, C_out=c, accumulate=True, c_bf16=True)
    t_old = timeit(old)
    probs_f = [{"A": a, "B": Wh[i % 4], "C": c, "bias": bias, "act": ops.ACT_TANH} for i, (a, c) in enumerate(zip(A, C))]
    t_f = timeit(lambda: ops.gemm_rows256(probs_f))
    byf = sum(M * 512 * 2 for M in Ms)
    def oldf():
        for i, (a, c) in enumerate(zip(A, C)):
            ops.gemm_bf16(ops.NT, a, Wh[i % 4], a.shape[0], 256, 256, bias=bias, C_out=c, act=ops.ACT_TANH, c_bf16=True)
    t_oldf = timeit(oldf)
    print(f"{name:32s} dX rows {t:7.1f} us {by / t / 1e6:5.2f} TB/s | dX gemm_bf16 {t_old:7.1f} us | fwd rows {t_f:7.1f} us {byf / t_f / 1e6:5.2f} TB/s"
          f" | fwd gemm_bf16 {t_oldf:7.1f} us")
