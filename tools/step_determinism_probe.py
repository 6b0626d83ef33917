"""Run-to-run determinism of the whole fused train step (forward, losses, backward, Adam): the same step REPS times from the same
parameters, optimiser state and dropout counter; the loss vector, the gradient bucket and the updated parameters of every run are
compared bit for bit with the first run's, per gradient tensor.  BF=0|1 (bf16 storage), REPS, CFG=c2|c5."""
import os
import sys

sys.path.insert(0, os.getcwd())
import torch  # noqa: E402

from oracle import sdumc_oracle as O  # noqa: E402  (parameter initialisation only)
from sdumc_amd import _lib  # noqa: E402
from sdumc_amd import engine as E  # noqa: E402

cfg = os.environ.get("CFG", "c2")
dims, B, Tn = ((1024, 1024, 1024, 1024), 32, (512, 512, 512, 512)) if cfg == "c5" else ((1024, 4096, 1024, 4096), 64, (375, 32, 225, 32))
bf = os.environ.get("BF", "1") == "1"
reps = int(os.environ.get("REPS", "100"))
P = O.init_params(dims, seed=0)
lay = E.ParamLayout.get(*dims[:3])
flat0 = torch.zeros(lay.total)
for k, v in lay.views(flat0).items():
    v.copy_(P[k])
flat0 = flat0.cuda()
g = torch.Generator(device="cuda").manual_seed(37)
batch = [torch.randn(B, Tn[i], dims[i], device="cuda", generator=g) for i in range(4)] + [torch.rand(B, device="cuda", generator=g) * 6 - 3]
flat = flat0.clone()
ts = E.TrainStep(flat, B, Tn, dims, seed=5, bf16=bf)
ts.set_batch(*batch)
ref = None
nbad, which = 0, {}
for rep in range(reps):
    flat.copy_(flat0)
    ts.adam_m.zero_()
    ts.adam_v.zero_()
    ts.hyper[1] = 0.0
    ts.rng.set_call(0)
    losses = ts.run().clone()
    torch.cuda.synchronize()
    cur = (losses, ts.grads.clone(), flat.clone())
    if ref is None:
        ref = cur
        continue
    if not all(torch.equal(a.view(torch.int32), b.view(torch.int32)) for a, b in zip(ref, cur)):
        nbad += 1
        gv_r = lay.views(torch.cat([ref[1], torch.zeros(lay.total - lay.live, device="cuda")]))
        gv_c = lay.views(torch.cat([cur[1], torch.zeros(lay.total - lay.live, device="cuda")]))
        bad = [k for k in lay.live_names() if not torch.equal(gv_r[k], gv_c[k])]
        for k in bad:
            which[k] = which.get(k, 0) + 1
        if nbad <= 4:
            worst = max(bad, key=lambda k: float((gv_r[k] - gv_c[k]).abs().max() / (gv_r[k].abs().max() + 1e-30))) if bad else None
            print(f"rep {rep}: losses equal {torch.equal(ref[0], cur[0])}; {len(bad)} gradient tensors differ"
                  + (f", worst {worst}: max rel-to-max {float((gv_r[worst] - gv_c[worst]).abs().max() / (gv_r[worst].abs().max() + 1e-30)):.3g}" if worst else ""))
top = sorted(which.items(), key=lambda kv: -kv[1])[:8]
print(cfg, "bf16" if bf else "fp32", f"lib={os.path.basename(_lib.LIB_PATH)}", f"steps that differ from the first: {nbad} of {reps - 1}; tensors most often different: {top}")
print("cluster error word", _lib.lib.sdumc_chain_cluster_error_())
