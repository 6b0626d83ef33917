"""bf16-storage mode smoke: C3-shaped step (B = 8) against the fp32 path on the GPU, then timing at B = 64."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sdumc_amd import engine
import bench
dims = (1024, 4096, 1024, 4096)
T = (375, 32, 225, 32)
for B in (8, 64):
    bench.DIMS, bench.T_MOSEI = dims, T
    flat, lay = bench.init_flat_params(engine, torch.device("cuda"))
    g = torch.Generator(device="cuda").manual_seed(1)
    batch = [torch.randn(B, T[i], dims[i], device="cuda", generator=g) for i in range(4)] + [torch.rand(B, device="cuda", generator=g) * 6 - 3]
    res = {}
    for mode in (False, "operands", True):
        f = flat.clone()
        ts = engine.TrainStep(f, B, T, dims, seed=5, bf16=mode)
        ts.set_batch(*batch)
        l = ts.run().clone()
        torch.cuda.synchronize()
        res[mode] = (l.cpu(), ts.grads.clone().cpu(), ts.vals.clone().cpu())
        for _ in range(5):
            ts.run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            ts.run()
        torch.cuda.synchronize()
        print(f"B={B} bf16={mode}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms/step  losses {[round(float(x), 4) for x in l[:7]]}", flush=True)
    ref = res[False]
    for mode in ("operands", True):
        l, gr, v = res[mode]
        gerr = float((gr - ref[1]).norm() / ref[1].norm())
        print(f"   vs fp32: loss rel {float(abs(l[0] - ref[0][0]) / ref[0][0]):.2e}, vals max abs {float((v - ref[2]).abs().max()):.2e}, grads rel-norm {gerr:.2e}, finite {bool(torch.isfinite(gr).all())}")
