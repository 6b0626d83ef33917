"""Run-to-run determinism of the forward pass, tensor by tensor.

Runs the same eval-mode (or train-mode, TRAIN=1) forward REPS times and compares, after every run, each intermediate tensor of
the workspace plan (sdumc_debug_plan_table) with the first run's, in pipeline order: the first tensor that differs is where a
run-to-run difference enters.  Environment: BF=0|1 (bf16 storage), REPS, POISON=1 (NaN-fill the workspace before every run),
SERIAL=1 (one lane), BG=<background-lane mode>, SDUMC_CL_MODE / SDUMC_CHAIN_CLUSTER (read by the library),
NEIGHBOUR=1 (a bandwidth-heavy copy kernel loop on another torch stream while the forward runs)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.getcwd())
import torch  # noqa: E402

from oracle import sdumc_oracle as O  # noqa: E402  (parameter initialisation only)
from sdumc_amd import _lib  # noqa: E402
from sdumc_amd import engine as E  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
if cfg == "c5":
    dims, B, Tn = (1024, 1024, 1024, 1024), 32, (512, 512, 512, 512)
else:
    dims, B, Tn = (1024, 4096, 1024, 4096), 64, (375, 32, 225, 32)
bf = os.environ.get("BF", "1") == "1"
train = os.environ.get("TRAIN", "0") == "1"
reps = int(os.environ.get("REPS", "30"))
P = O.init_params(dims, seed=0)
lay = E.ParamLayout.get(*dims[:3])
flat = torch.zeros(lay.total)
for k, v in lay.views(flat).items():
    v.copy_(P[k])
flat = flat.cuda()
g = torch.Generator(device="cuda").manual_seed(37)
audio, text, video, feat4 = [torch.randn(B, Tn[i], dims[i], device="cuda", generator=g) for i in range(4)]
if os.environ.get("SERIAL") == "1":
    _lib.lib.sdumc_set_concurrency(0)
if "BG" in os.environ:
    _lib.lib.sdumc_set_background_lane(int(os.environ["BG"]))

ORDER = ["x_a", "x_t", "x_v", "keys0a", "keys0t", "keys0v", "attn0a", "attn0t", "attn0v", "pooled0a", "pooled0t", "pooled0v",
         "hpre", "wt", "u1", "u", "att1", "att2", "alpha", "qin", "q", "qp", "keys1a", "keys1t", "keys1v", "attn1a", "attn1t",
         "attn1v", "pooled1a", "pooled1t", "pooled1v", "ca_out", "c1", "c", "h", "e1", "e2", "beta", "z", "vals", "r1", "r"]


def plan_table(nc):
    need = -_lib.lib.sdumc_debug_plan_table(C.byref(nc.dims), None, 0)
    buf = C.create_string_buffer(need)
    assert _lib.lib.sdumc_debug_plan_table(C.byref(nc.dims), buf, need) > 0
    tab = {}
    for line in buf.value.decode().splitlines():
        name, off, n = line.split()
        tab[name] = (int(off), int(n))
    return tab


def snapshot(nc, tab):
    w = nc.workspace.view(torch.float32)
    return {k: w[o:o + n].clone() for k, (o, n) in tab.items()}


nb_stream = torch.cuda.Stream() if os.environ.get("NEIGHBOUR") == "1" else None
nb_a = torch.empty(64 << 20, device="cuda") if nb_stream else None
nb_b = torch.empty(64 << 20, device="cuda") if nb_stream else None

rng = E.RngState(5, "cuda", call=0) if train else None
nc = E.NetCall(flat, audio, [text, feat4], video, train, rng, bf16=bf)
tab = plan_table(nc)
ref = None
first_bad = {}
nbad = 0
fresh = os.environ.get("FRESH") == "1"      # a new NetCall (workspace allocation) per run, as a training loop over shapes does
for rep in range(reps):
    if fresh and rep:
        nc = E.NetCall(flat, audio, [text, feat4], video, train, rng, bf16=bf)
    if rng is not None:
        rng.set_call(0)
    if os.environ.get("POISON") == "1":
        nc.workspace.view(torch.float32).fill_(float("nan"))
    for name in os.environ.get("POISON_ONLY", "").split(","):
        if name:
            o, n = tab[name]
            nc.workspace.view(torch.float32)[o:o + n].fill_(float("nan"))
    torch.cuda.synchronize()
    if nb_stream is not None:
        with torch.cuda.stream(nb_stream):
            for _ in range(6):
                nb_b.copy_(nb_a)
    nc.forward()
    torch.cuda.synchronize()
    snap = snapshot(nc, tab)
    if ref is None:
        ref = snap
        continue
    bad = [k for k in ORDER if k in snap and not torch.equal(snap[k].view(torch.int32), ref[k].view(torch.int32))]
    if bad:
        upstream = [k for k in bad if k.startswith(("x_", "keys", "attn0", "pooled0", "hpre", "wt"))]
        if upstream:
            print(f"rep {rep}: tensors that do NOT depend on stage A differ too: {upstream}")
        nbad += 1
        k = bad[0]
        first_bad[k] = first_bad.get(k, 0) + 1
        if nbad <= 6:
            a, b = ref[k], snap[k]
            d = (a.view(torch.int32) != b.view(torch.int32)).nonzero().flatten()
            ad = (a[d].double() - b[d].double()).abs()
            rel = ad / (a[d].double().abs() + 1e-30)
            print(f"rep {rep}: first differing tensor {k} ({len(bad)} tensors differ: {bad[:8]}); {d.numel()} of {a.numel()} words differ, "
                  f"first at {d[:6].tolist()}, max abs {float(ad.max()):.3g}, max rel {float(rel.max()):.3g}, "
                  f"finite {bool(torch.isfinite(b).all())}")
            if k == "u1":
                # which (modality, row, column, quad component) differ, the two values, and the fp64 truth from hpre and the parameters
                V = nc.V
                hp = snap["hpre"].view(3, V, 256).double().cpu()
                names = ("audio_mlp.0", "text_mlp.0", "video_mlp.0")
                truth = torch.stack([torch.relu(hp[m] @ P[names[m] + ".weight"].double().t() + P[names[m] + ".bias"].double()) for m in range(3)]).reshape(-1)
                idx = d.cpu()
                m_, v_, c_ = idx // (V * 256), (idx // 256) % V, idx % 256
                print("   modality/row/col:", sorted(set(zip(m_.tolist(), v_.tolist())))[:8], "columns", sorted(set(c_.tolist()))[:40])
                print("   quad components:", torch.bincount(c_ % 4, minlength=4).tolist(), "64-col slices:", torch.bincount(c_ // 64, minlength=4).tolist())
                ea = (a.cpu().double()[idx] - truth[idx]).abs()
                eb = (b.cpu().double()[idx] - truth[idx]).abs()
                print(f"   |first run - truth| max {float(ea.max()):.3g}; |this run - truth| max {float(eb.max()):.3g}")
                for t in range(min(6, idx.numel())):
                    print(f"   [{int(idx[t])}] first run {float(a[idx[t]]):.6g} this run {float(b[idx[t]]):.6g} truth {float(truth[idx[t]]):.6g}")
print(cfg, "bf16" if bf else "fp32", "train" if train else "eval", f"CL_MODE={os.environ.get('SDUMC_CL_MODE', '0')}",
      f"CLUSTER={os.environ.get('SDUMC_CHAIN_CLUSTER', '1')}", f"NEIGHBOUR={os.environ.get('NEIGHBOUR', '0')}", "fresh workspaces" if fresh else "one workspace",
      f"runs that differ from the first: {nbad} of {reps - 1}; first differing tensor counts: {first_bad}")
print("cluster error word", _lib.lib.sdumc_chain_cluster_error_())
if int(os.environ.get("SDUMC_CL_MODE", "0")) & (16 | 64):
    dbg = (C.c_uint32 * 256)()
    _lib.lib.sdumc_chain_cluster_debug_read_(dbg, 256)
    print("self-check records (mode 16: cached weight loads != agent-scope loads; mode 64: LDS copy of hpre != global hpre):", dbg[0])
    import struct
    for i in range(min(dbg[0], 20)):
        r = dbg[8 + 12 * i: 8 + 12 * i + 12]
        f = lambda w: struct.unpack("f", struct.pack("I", w))[0]
        print(f"   wg {r[0]} tid {r[1]} tag {r[2]} addr/idx {r[3]:08x}: first {[round(f(w), 6) for w in r[4:8]]} ({[hex(w) for w in r[4:8]]}) second {[round(f(w), 6) for w in r[8:12]]}")
