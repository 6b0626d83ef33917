"""gemm_b1.hip on the GPU: the NT kernel against fp64 products of the same bf16 operands (and beside gemm_bf16.hip's LDS-staged
kernel), and launch times on the bf16-storage step's shapes.  python tools/b1_check.py [--time]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdumc_amd import ops  # noqa: E402


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
    bf = torch.bfloat16
    for (M, K, act, mod, sk) in ((9003, 1024, ops.ACT_NONE, 0, 0), (9003, 1024, ops.ACT_TANH, 0, 4), (2048, 4096, ops.ACT_NONE, 0, 0),
                                 (700, 128, ops.ACT_RELU, 0, 0), (36864 + 32, 256, ops.ACT_TANH, 18432 + 16, 0), (64, 256, ops.ACT_NONE, 0, 0)):
        X = rn(mod or M, K).to(bf)
        W, b = rn(256, K) / K ** 0.5, rn(256) * 0.1
        Wf = ops.b1_frag(W)
        Xd = X.double().repeat(2, 1) if mod else X.double()
        ref = Xd @ W.to(bf).double().t() + b.double()
        ref = torch.tanh(ref) if act == ops.ACT_TANH else (ref.clamp_min(0) if act == ops.ACT_RELU else ref)
        for od in (torch.float32, bf):
            c = ops.gemm_b1_nt(X, Wf, M, 256, K, bias=b, act=act, a_row_mod=mod, out_dtype=od, splitk=sk)
            e = err(c, ref)
            print(f"M={M} K={K} act={act} mod={mod} splitk={sk} out={od}: {e:.2e}")
            assert e < (6e-3 if od == bf else 2e-5), e
    # two A tensors (the text slot's streams)
    Xa, Xb = rn(1024, 4096).to(bf), rn(1024, 4096).to(bf)
    W, b = rn(256, 4096) / 64, rn(256)
    Wf = ops.b1_frag(W)
    c = ops.gemm_b1_nt(Xa, Wf, 2048, 256, 4096, bias=b, out_dtype=torch.float32, A_second=Xb, second_row0=1024)
    ref = torch.cat([Xa, Xb]).double() @ W.to(bf).double().t() + b.double()
    print(f"two A tensors: {err(c, ref):.2e}")
    assert err(c, ref) < 2e-5
    c1 = ops.gemm_b1_nt(Xa, Wf, 2048, 256, 4096, bias=b, out_dtype=torch.float32, A_second=Xb, second_row0=1024)
    assert torch.equal(c, c1), "not deterministic"
    if "--time" not in sys.argv:
        return
    for name, M, K, mod in (("audio frames", 51200, 1024, 0), ("video frames", 30720, 1024, 0), ("text frames", 2048, 4096, 0),
                            ("keys audio", 153600, 256, 0), ("keys video", 92160, 256, 0), ("keys text", 6144, 256, 0)):
        X = rn(mod or M, K).to(bf)
        W, b = rn(256, K) / K ** 0.5, rn(256)
        Wf, Wb = ops.b1_frag(W), W.to(bf)
        fl = 2.0 * M * 256 * K
        launch, _ = ops.gemm_b1_nt_call(X, Wf, M, 256, K, bias=b, act=ops.ACT_TANH)
        t = timeit(launch)
        t0 = timeit(lambda: ops.gemm_bf16(ops.NT, X, Wb, M, 256, K, bias=b, act=ops.ACT_TANH, c_bf16=True))
        print(f"{name:13s} M={M:6d} K={K:4d}: b1 {t:7.1f} us {fl / t * 1e-6:6.1f} TF   gemm_bf16s {t0:7.1f} us {fl / t0 * 1e-6:6.1f} TF")


if __name__ == "__main__":
    main()
