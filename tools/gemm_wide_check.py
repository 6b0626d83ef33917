"""gemm_wide.hip against gemm_f32.hip on the shapes of the SDUMC step: results (max relative difference) and time per launch,
all tile configurations, in ONE process (interleaved rounds).  tile: 0 = the old plan, 11 = 64x256, 12 = 128x256,
13 = 128x128, 14 = 64x128, 15..18 = the persistent NT variants of 11..14."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sdumc_amd import ops, _lib

NT, NN, TN = ops.NT, ops.NN, ops.TN
dev = "cuda"


def timeit(fn, reps=20, rounds=3):
    best = 1e9
    for _ in range(rounds):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


def rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def case(name, layout, M, N, K, tiles, groups=1, **kw):
    g = torch.Generator(device=dev).manual_seed(1)
    mk = lambda *s: torch.randn(*s, device=dev, generator=g)
    row_mod = kw.pop("row_mod", 0)
    drop = kw.pop("drop", False)
    bias = kw.pop("bias", False)
    act = kw.pop("act", ops.ACT_NONE)
    accumulate = kw.pop("accumulate", False)
    colsum = kw.pop("colsum", False)
    if layout == NT:
        A = [mk(row_mod or M, K) for _ in range(groups)]
        if row_mod and groups > 1:
            A = [A[0]] * groups
        B = [mk(N, K) * 0.05 for _ in range(groups)]
    else:
        A = [mk(K, M) for _ in range(groups)]
        B = [mk(row_mod or K, N) for _ in range(groups)]
        if row_mod and groups > 1:
            B = [B[0]] * groups
    biases = [mk(N) for _ in range(groups)] if bias else None
    args = dict(act=act, accumulate=accumulate)
    if row_mod:
        args["a_row_mod" if layout == NT else "b_row_mod"] = row_mod
    keep = []
    if drop:
        # row space [streams][samples][rows][width]: one "sample" of R rows
        R = M if layout == NT else K
        W = K if layout == NT else N
        d = _lib.make_dropout(True, 3, 0.5, R, W, 1, seed=77)
        bits = []
        for gi in range(groups):
            dd = _lib.make_dropout(True, 3 + 5 * gi, 0.5, R, W, 1, seed=77)
            bits.append(ops.dropout_bits(dd, 1))
        keep.append(bits)
        args["a_drop" if layout == NT else "b_drop"] = d
        args["ab_drop_group_stride"] = 5
        args["ab_drop_bits"] = bits
    flops = 2.0 * M * N * K * groups
    ref = None
    out = []
    Cinit = [mk(M, N) for _ in range(groups)] if accumulate else None
    for t in tiles:
        C0 = [c.clone() for c in Cinit] if accumulate else [torch.empty(M, N, device=dev) for _ in range(groups)]
        cs = [torch.zeros(M, device=dev) for _ in range(groups)] if colsum else None

        def run(Cs=None):
            Cs = Cs if Cs is not None else [c.clone() for c in C0] if accumulate else C0
            ops.gemm(layout, A if groups > 1 else A[0], B if groups > 1 else B[0], M, N, K,
                     bias=(biases if groups > 1 else biases[0]) if bias else None,
                     C_out=Cs if groups > 1 else Cs[0], tile=t, splitk=0, colsum_a=(cs if groups > 1 else cs[0]) if colsum else None,
                     **args)
            return Cs
        Cs = run([c.clone() for c in C0])
        torch.cuda.synchronize()
        res = torch.stack(Cs).clone()
        csr = torch.stack(cs).clone() if colsum else None
        if ref is None:
            ref, cref = res, csr
            err = 0.0
        else:
            err = rel(res, ref)
            if colsum:
                err = max(err, rel(csr, cref))
        us = timeit(lambda: run(C0))
        out.append((t, us, flops / us / 1e6, err))
    print(f"{name:34s} {['NT','NN','TN'][layout]} M={M:6d} N={N:5d} K={K:6d} g={groups}: " +
          "  ".join(f"[t{t}: {us:7.1f}us {tf:6.1f}TF err {err:.1e}]" for t, us, tf, err in out), flush=True)


if __name__ == "__main__":
    NTT = (0, 11, 12, 13, 14, 15, 16, 17, 18)
    TNT = (0, 11, 13, 14)
    case("frame proj audio", NT, 24000, 256, 1024, NTT, bias=True)
    case("frame proj video", NT, 14400, 256, 1024, NTT, bias=True)
    case("frame proj text", NT, 2048, 256, 4096, NTT, bias=True)
    case("frame proj text C2 (both streams)", NT, 6400, 256, 4096, NTT, bias=True)
    case("frame proj text C2 (one stream)", NT, 3200, 256, 4096, NTT, bias=True)
    case("keys audio 1 site", NT, 48000, 256, 256, NTT, bias=True, act=ops.ACT_TANH, drop=True, row_mod=24000)
    case("keys audio 2 sites", NT, 48000, 256, 256, NTT, groups=2, bias=True, act=ops.ACT_TANH, drop=True, row_mod=24000)
    case("keys video 1 site", NT, 28800, 256, 256, NTT, bias=True, act=ops.ACT_TANH, drop=True, row_mod=14400)
    case("keys text 1 site", NT, 4096, 256, 256, NTT, bias=True, act=ops.ACT_TANH, drop=True)
    case("keys dX audio (as NT, accumulate)", NT, 48000, 256, 256, NTT, accumulate=True)
    case("keys dX audio 2 sites", NT, 48000, 256, 256, NTT, groups=2, accumulate=True)
    case("ragged M", NT, 24003, 256, 1024, NTT, bias=True)
    case("frame dW audio", TN, 256, 1024, 24000, TNT, colsum=True)
    case("frame dW video", TN, 256, 1024, 14400, TNT, colsum=True)
    case("frame dW text", TN, 256, 4096, 2048, TNT, colsum=True)
    case("keys dW audio 1 site", TN, 256, 256, 48000, TNT, colsum=True, drop=True, row_mod=24000)
    case("keys dW audio 2 sites", TN, 256, 256, 48000, TNT, groups=2, colsum=True, drop=True, row_mod=24000)
    case("keys dW video 1 site", TN, 256, 256, 28800, TNT, colsum=True, drop=True, row_mod=14400)
    case("ragged K", TN, 256, 1024, 24007, TNT, colsum=True)
    case("square 4096", NT, 4096, 4096, 4096, (1, 12, 13, 16, 17))
    case("square 4096", TN, 4096, 4096, 4096, (1, 11, 13, 14))
    case("dW-like, no split needed", TN, 2048, 2048, 24000, (1, 11, 13, 14))
