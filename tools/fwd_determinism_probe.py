import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from oracle import sdumc_oracle as O
from sdumc_amd import engine as E
cfg = sys.argv[1] if len(sys.argv) > 1 else "c5"
if cfg == "c5":
    dims, B, Tn = (1024, 1024, 1024, 1024), 32, (512, 512, 512, 512)
else:
    dims, B, Tn = (1024, 4096, 1024, 4096), 64, (375, 32, 225, 32)
bf = os.environ.get("BF", "1") == "1"
P = O.init_params(dims, seed=0)
lay = E.ParamLayout.get(*dims[:3])
flat = torch.zeros(lay.total)
for k, v in lay.views(flat).items():
    v.copy_(P[k])
flat = flat.cuda()
g = torch.Generator(device="cuda").manual_seed(37)
audio, text, video, feat4 = [torch.randn(B, Tn[i], dims[i], device="cuda", generator=g) for i in range(4)]
NAMES = ["vals", "fused", "rnc", "text_hidden", "cross_text"]
from sdumc_amd import _lib
if os.environ.get("SERIAL") == "1":
    _lib.lib.sdumc_set_concurrency(0)
if "BG" in os.environ:
    _lib.lib.sdumc_set_background_lane(int(os.environ["BG"]))
if "HOLD" in os.environ:
    _lib.lib.sdumc_chain_cluster_test_hold_(int(os.environ["HOLD"]))
ref = None
bad = {n: 0 for n in NAMES}
reps = int(os.environ.get("REPS", "30"))
same_nc = os.environ.get("SAME", "0") == "1"
nc = None
for rep in range(reps):
    if nc is None or not same_nc:
        nc = E.NetCall(flat, audio, [text, feat4], video, False, None, bf16=bf)
    if os.environ.get("POISON") == "1":
        nc.workspace.view(torch.float32).fill_(float("nan"))
    out = [t.clone() for t in nc.forward()]
    torch.cuda.synchronize()
    if ref is None:
        ref = out
    for n, a, b in zip(NAMES, ref, out):
        if not torch.equal(a, b):
            bad[n] += 1
            if not torch.isfinite(b).all():
                bad[n] += 1000
            if n in ("text_hidden", "fused") and bad[n] <= 3:
                d = (a != b).reshape(a.shape[0], -1)
                rows = d.any(1).nonzero().flatten().tolist()
                cols = d.any(0).nonzero().flatten().tolist()
                print(n, "rep", rep, "rows", rows[:40], "ncols", len(cols), "cols", cols[:24], "max rel", float(((a - b).abs() / (a.abs() + 1e-9)).max()))
print(cfg, "bf16" if bf else "fp32", "same workspace" if same_nc else "fresh workspaces", "mismatching reps of", reps, bad)
