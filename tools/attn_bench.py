"""The attention-pooling launches of a C2 step, alone: the grouped Cross_Attention pair (three sites, nq = 7; forward partial +
combine, backward + dq reduce) and the three FRA2UTT-site backwards (nq = 1), fp32 and bf16 storage, round-3 kernels against
the round-4 "v2" kernels (sdumc_attnpool_set_v2_) -- results compared bit for bit, times from HIP events over REPS launches that
rotate through SETS independent buffer sets (one set is ~340 MB, more than the 256 MB Infinity Cache, so nothing stays resident).
Prints us per launch pair and the HBM-side rate for the algorithmic bytes of the launch."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sdumc_amd import ops, _lib

dev = "cuda"
B, Dm = 64, 256
V = 2 * B
SITES = [("audio", 375), ("video", 225), ("text", 32)]
REPS = int(os.environ.get("REPS", "40"))
SETS = int(os.environ.get("SETS", "2"))


def make_set(nq, bf16, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    fdt = torch.bfloat16 if bf16 else torch.float32
    fwd, bwd, keep = [], [], []
    for name, T in SITES:
        shared = name != "text"
        xs = B if shared else V
        x = torch.randn(xs, T, Dm, device=dev, generator=g)
        keys = torch.tanh(torch.randn(V, T, Dm, device=dev, generator=g))
        q = torch.randn(V, nq, Dm, device=dev, generator=g) / 4
        xdrop = _lib.make_dropout(True, 23, 0.5, T, Dm, B, call0=4, seed=9)
        bits = None
        if bf16:      # bf16 storage: the masked frames are materialised per virtual sample, no mask in the pooling kernels
            m = ops.dropout_mask(xdrop, 2).view(V, T, Dm)
            x = ((x if not shared else x.repeat(2, 1, 1)) * m).to(fdt).contiguous()
            xs = V
            xdrop = _lib.make_dropout(False, 23, 0.5, T, Dm, B)
            keys = keys.to(fdt)
        else:
            bits = ops.dropout_bits(xdrop, 2)
            xdrop.bits = _lib.ptr(bits)
        odrop = _lib.make_dropout(True, 24, 0.5, nq, Dm, B, call0=4, seed=9)
        attn, pooled, out = torch.empty(V, T, nq, device=dev), torch.empty(V, nq, Dm, device=dev), torch.empty(V, nq, Dm, device=dev)
        a = ops.attnpool_desc(x, keys, q, V, T, nq, xs, nq * Dm, xdrop, odrop, attn, pooled, out, tickets=False)
        a.bf16 = 1 if bf16 else 0
        need = _lib.lib.sdumc_attnpool_fwd_workspace_bytes(V, T, nq)
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        a.workspace, a.workspace_bytes = _lib.ptr(ws), need
        dout = torch.randn(V, nq, Dm, device=dev, generator=g)
        dz, dxd = torch.empty(V, T, Dm, device=dev, dtype=fdt), torch.empty(V, T, Dm, device=dev, dtype=fdt)
        dq = torch.empty(V, nq, Dm, device=dev)
        nb = _lib.lib.sdumc_attnpool_bwd_workspace_bytes(V, T, nq)
        wsb = torch.empty(nb, dtype=torch.uint8, device=dev)
        bb = _lib.AttnPoolBwd()
        bb.f = a
        bb.dout, bb.dz, bb.dxd, bb.dq = _lib.ptr(dout), _lib.ptr(dz), _lib.ptr(dxd), _lib.ptr(dq)
        bb.workspace, bb.workspace_bytes = _lib.ptr(wsb), nb
        fwd.append(a)
        bwd.append(bb)
        keep.append((x, keys, q, bits, attn, pooled, out, ws, dout, dz, dxd, dq, wsb))
    fa = (_lib.AttnPool * 3)(*fwd)
    ba = (_lib.AttnPoolBwd * 3)(*bwd)
    return fa, ba, keep


def run_fwd(fa):
    _lib.check(_lib.lib.sdumc_attnpool_fwd_multi(fa, 3, _lib.current_stream()), "fwd_multi")


def run_bwd(ba):
    _lib.check(_lib.lib.sdumc_attnpool_bwd_multi(ba, 3, _lib.current_stream()), "bwd_multi")


def run_bwd_single(ba):
    for i in range(3):
        _lib.check(_lib.lib.sdumc_attnpool_bwd(C.byref(ba[i]), _lib.current_stream()), "bwd")


def timeit(fn, sets):
    for s in sets:
        fn(s)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for r in range(REPS):
            fn(sets[r % len(sets)])
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / REPS)
    return best


def outputs(keep, which):
    idx = {"fwd": (4, 5, 6), "bwd": (9, 10, 11)}[which]
    return [k[i].clone() for k in keep for i in idx]


def case(nq, bf16):
    sets = [make_set(nq, bf16, 1 + s) for s in range(SETS)]
    rows = sum(V * T for _, T in SITES)
    xrows = sum((B if n != "text" else V) * T for n, T in SITES) if not bf16 else rows
    eb = 2 if bf16 else 4
    fwd_bytes = rows * Dm * eb + xrows * Dm * eb + (0 if bf16 else rows * 64)
    bwd_bytes = fwd_bytes + 2 * rows * Dm * eb
    res = {}
    for v2 in (0, 1, 0, 1):
        _lib.lib.sdumc_attnpool_set_v2_(v2)
        tf = timeit(lambda s: run_fwd(s[0]), sets)
        of = outputs(sets[0][2], "fwd")
        tb = timeit(lambda s: run_bwd(s[1]), sets) if nq == 7 else timeit(lambda s: run_bwd_single(s[1]), sets)
        ob = outputs(sets[0][2], "bwd")
        res.setdefault(v2, []).append((tf, tb, of, ob))
    for v2 in (0, 1):
        tf = min(r[0] for r in res[v2])
        tb = min(r[1] for r in res[v2])
        print(f"nq={nq} {'bf16' if bf16 else 'fp32'} {'v2 ' if v2 else 'old'}: fwd pair {tf:6.1f} us ({fwd_bytes / tf / 1e6:5.2f} TB/s of {fwd_bytes / 1e6:.0f} MB)"
              f" | bwd {'pair' if nq == 7 else '3 launches'} {tb:6.1f} us ({bwd_bytes / tb / 1e6:5.2f} TB/s of {bwd_bytes / 1e6:.0f} MB)", flush=True)
    same_f = all(torch.equal(a, b) for a, b in zip(res[0][0][2], res[1][0][2]))
    same_b = all(torch.equal(a, b) for a, b in zip(res[0][0][3], res[1][0][3]))
    print(f"   v2 == old bit for bit: forward {same_f}, backward {same_b}")
    if not (same_f and same_b):
        for name, la, lb in (("fwd", res[0][0][2], res[1][0][2]), ("bwd", res[0][0][3], res[1][0][3])):
            for i, (a, b) in enumerate(zip(la, lb)):
                if not torch.equal(a, b):
                    print(f"   {name} tensor {i}: max abs diff {float((a.float() - b.float()).abs().max()):.3g}, finite {bool(torch.isfinite(b.float()).all())}")
    _lib.lib.sdumc_attnpool_set_v2_(1)


if __name__ == "__main__":
    only = os.environ.get("CASES")           # e.g. CASES=7f,1f,7h,1h (nq + f|h)
    for bf16 in (False, True):
        for nq in (7, 1):
            if only is None or f"{nq}{'h' if bf16 else 'f'}" in only.split(","):
                case(nq, bf16)
