"""Times the RnC loss kernels (sdumc_rnc_fwd_bwd_rep) at n = 2B rows, dim 64."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdumc_amd import _lib
from sdumc_amd._lib import lib, ptr, check, current_stream
for B in (64, 512):
    n, dim = 2 * B, 64
    f = torch.randn(n, dim, device="cuda"); y = torch.rand(B, device="cuda") * 6 - 3
    loss = torch.empty(1, device="cuda"); df = torch.empty(n, dim, device="cuda")
    ws = torch.empty(lib.sdumc_rnc_workspace_bytes(n), dtype=torch.uint8, device="cuda")
    def run(): check(lib.sdumc_rnc_fwd_bwd_rep(ptr(f), ptr(y), n, dim, 2.0, 0.8, 0, n, ptr(loss), ptr(df), ptr(ws), current_stream()))
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    print(f"n={n}: {e0.elapsed_time(e1) * 1e3 / 50:.1f} us per call (3 kernels), loss {float(loss):.5f}")
