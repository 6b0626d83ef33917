"""gemm_p3.hip on the GPU: split / join round trip, the NT kernel against fp64 products (and beside the in-fragment split kernel of
gemm_wide.hip), and launch times on the step's shapes.  python tools/p3_check.py [--time]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdumc_amd import ops, _lib  # noqa: E402


def err(got, ref):
    return float((got.double() - ref).abs().max() / ref.abs().max())


def timeit(fn, reps=30, warm_s=0.3):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < warm_s:
        fn()
        torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    g = torch.Generator(device="cuda").manual_seed(1)
    rn = lambda *s: torch.randn(*s, device="cuda", generator=g)
    # round trip
    x = rn(1000, 1024) * torch.exp2(torch.randint(-20, 20, (1000, 1024), device="cuda", generator=g).float())
    p = ops.p3_split(x)
    assert torch.equal(ops.p3_join(p, 1024), x), "split -> join is not bit-exact"
    print("round trip bit-exact")
    # frame-like: ragged M, bias; every tile height; split-K
    M, K = 9003, 1024
    X, W, b = rn(M, K), rn(256, K) / K ** 0.5, rn(256)
    ref = X.double() @ W.double().t() + b.double()
    X3, W3 = ops.p3_split(X), ops.p3_split_frag(W)
    wide = ops.gemm(ops.NT, X, W, M, 256, K, bias=b, tile=14, splitk=1)
    print(f"frame M={M} K={K}: in-fragment split kernel {err(wide, ref):.2e}")
    for tm in (64, 96, 128):
        for sk in (1, 4):
            c, c3 = ops.gemm_p3_nt(X3, W3, M, 256, K, bias=b, tile_m=tm, splitk=sk, want_p3=True)
            e = err(c, ref)
            same = torch.equal(ops.p3_join(c3, 256), c)
            print(f"  p3 tile_m={tm} splitk={sk}: {e:.2e}  P3 output == fp32 output: {same}")
            assert e < 3e-6 and same
    # key-like: mask, row modulo, bias, tanh
    M2, mod = 36864 + 32, 18432 + 16
    x, W2, b2 = rn(mod, 256), rn(256, 256) / 16, rn(256) * 0.1
    d = _lib.make_dropout(True, 3, 0.5, M2, 256, 1, seed=77)
    bits = ops.dropout_bits(d, 1).view(M2, 64)
    mask = ops.dropout_mask(d, 1).view(M2, 256).double()
    ref2 = torch.tanh((x.double().repeat(2, 1) * mask) @ W2.double().t() + b2.double())
    x3, W23 = ops.p3_split(x), ops.p3_split_frag(W2)
    for tm in (64, 96, 128):
        c = ops.gemm_p3_nt(x3, W23, M2, 256, 256, bias=b2, act=ops.ACT_TANH, a_row_mod=mod, bits=bits, scale=2.0, tile_m=tm)
        e = err(c, ref2)
        print(f"key M={M2} masked tile_m={tm}: {e:.2e}")
        assert e < 3e-6
    if "--time" not in sys.argv:
        return
    print("\nlaunch times (us; TF = fp32-equivalent):")
    for name, M, K, sks in (("audio frame", 24000, 1024, (1,)), ("video frame", 14400, 1024, (1,)), ("text frame", 2048, 4096, (2, 4, 8)),
                            ("audio keys", 48000, 256, (1,)), ("video keys", 28800, 256, (1,)), ("text keys", 4096, 256, (1,))):
        X, W, b = rn(M, K), rn(256, K) / K ** 0.5, rn(256)
        X3, W3 = ops.p3_split(X), ops.p3_split_frag(W)
        gf = 2.0 * M * 256 * K / 1e9
        t = timeit(lambda: ops.gemm(ops.NT, X, W, M, 256, K, bias=b, tile=14 if M >= 8192 else 0, splitk=1 if M >= 8192 else 0))
        print(f"{name:12s} M={M:6d} K={K:5d}: wide split {t:7.1f} us = {gf / t * 1e3:6.1f} TF", end="")
        c = torch.empty(M, 256, device="cuda")
        for tm in (64, 96, 128):
            for sk in sks:
                t = timeit(ops.gemm_p3_nt_call(X3, W3, M, 256, K, bias=b, tile_m=tm, splitk=sk, C_out=c)[0])
                print(f" | p3 {tm}{'/' + str(sk) if sk > 1 else ''} {t:6.1f} us = {gf / t * 1e3:5.1f} TF", end="")
        t = timeit(ops.gemm_p3_nt_call(X3, W3, M, 256, K, bias=b, C_out=c)[0])
        print(f" | auto {t:6.1f}")
        t = timeit(lambda: ops.p3_split(X, out=X3))
        print(f"{'':12s} split of A: {t:6.1f} us = {M * K * 10 / t / 1e6:5.2f} TB/s")


if __name__ == "__main__":
    main()
