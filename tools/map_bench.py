"""The two kernels that read a batch IN PLACE through a row map, against the same products on a padded copy (C2 shapes):
   sdumc_gemm_p3_nt (frame projections, A rows by map) and sdumc_gemm_group_tn (frame dW, B rows by map).
   python tools/map_bench.py [utterances in the store]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdumc_amd import ops  # noqa: E402
from tools.p3_check import timeit  # noqa: E402

N_UTT = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
B = 64
g = torch.Generator(device="cuda").manual_seed(1)
gc = torch.Generator().manual_seed(2)


def store(T, d, ragged):
    lens = torch.randint(T // 4, T + 1, (N_UTT,), generator=gc) if ragged else torch.full((N_UTT,), T)
    rows = int(lens.sum())
    X = torch.zeros(rows + 1, d, device="cuda")
    X[:rows] = torch.randn(rows, d, device="cuda", generator=g)
    starts = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(lens, 0)[:-1]])
    return X, starts, lens, rows


def batch_map(starts, lens, zero_row, T):
    idx = torch.randperm(N_UTT, generator=gc)[:B]
    Tp = int(lens[idx].max())
    m = torch.full((B, Tp), zero_row, dtype=torch.int32)
    for b, e in enumerate(idx.tolist()):
        n = int(lens[e])
        m[b, :n] = torch.arange(int(starts[e]), int(starts[e]) + n, dtype=torch.int32)
    return m.reshape(-1).cuda(), Tp


for name, T, d, sk in (("audio", 375, 1024, 1), ("video", 225, 1024, 1), ("text", 32, 4096, 8)):
    for ragged in (False, True):
        X, starts, lens, rows = store(T, d, ragged)
        W = torch.randn(256, d, device="cuda", generator=g) / d ** 0.5
        bias = torch.randn(256, device="cuda", generator=g)
        X3, W3 = ops.p3_split(X), ops.p3_split_frag(W)
        amap, Tp = batch_map(starts, lens, rows, T)
        M = B * Tp
        Xb = X[amap.long()].contiguous()                    # the padded copy
        Xb3 = ops.p3_split(Xb)
        c0, c1 = torch.empty(M, 256, device="cuda"), torch.empty(M, 256, device="cuda")
        l0, _ = ops.gemm_p3_nt_call(Xb3, W3, M, 256, d, bias=bias, C_out=c0, splitk=sk)
        l1, _ = ops.gemm_p3_nt_call(X3, W3, M, 256, d, bias=bias, C_out=c1, splitk=sk, a_map=amap)
        l0(); l1()
        torch.cuda.synchronize()
        assert torch.equal(c0, c1), "mapped frame projection differs from the padded copy's"
        t0, t1 = timeit(l0, reps=50), timeit(l1, reps=50)
        # frame dW: C[256, d] = dx[M, 256]^T . feat[M, d]
        dx = torch.randn(M, 256, device="cuda", generator=g)
        p0 = [dict(A=dx, B=Xb, colsum=torch.empty(256, device="cuda"))]
        p1 = [dict(A=dx, B=X, b_map=amap, K=M, colsum=torch.empty(256, device="cuda"))]
        ops.gemm_group_tn(p0); ops.gemm_group_tn(p1)
        torch.cuda.synchronize()
        assert torch.equal(p0[0]["C"], p1[0]["C"]) and torch.equal(p0[0]["colsum"], p1[0]["colsum"]), "mapped frame dW differs"
        u0, u1 = timeit(lambda: ops.gemm_group_tn(p0), reps=30), timeit(lambda: ops.gemm_group_tn(p1), reps=30)
        # the same two products in bf16 storage (gemm_b1 / sdumc_gemm_group_tn_bf16)
        Xh, Xbh, dxh = X.bfloat16(), Xb.bfloat16(), dx.bfloat16()
        Wb = ops.b1_frag(W)
        ch0, ch1 = torch.empty(M, 256, device="cuda", dtype=torch.bfloat16), torch.empty(M, 256, device="cuda", dtype=torch.bfloat16)
        small = X.numel() * 2 < 0xFFFFFFF0
        b0, _ = ops.gemm_b1_nt_call(Xbh, Wb, M, 256, d, bias=bias, C_out=ch0, splitk=4 if name == "text" else 0)
        b1, _ = ops.gemm_b1_nt_call(Xh, Wb, M, 256, d, bias=bias, C_out=ch1, splitk=4 if name == "text" else 0, a_map=amap, a_map_rows=X.shape[0])
        b2, _ = ops.gemm_b1_nt_call(Xh, Wb, M, 256, d, bias=bias, C_out=ch1, splitk=4 if name == "text" else 0, a_map=amap)
        b0(); b1()
        torch.cuda.synchronize()
        assert torch.equal(ch0, ch1), "mapped bf16 frame projection differs"
        ch1.zero_(); b2()
        torch.cuda.synchronize()
        assert torch.equal(ch0, ch1), "mapped (64-bit) bf16 frame projection differs"
        amap4 = torch.cat([amap, torch.zeros(64, dtype=torch.int32, device="cuda")])
        q0 = [dict(A=dxh, B=Xbh, colsum=torch.empty(256, device="cuda"))]
        q1 = [dict(A=dxh, B=Xh, b_map=amap4, K=M, colsum=torch.empty(256, device="cuda"))]
        ops.gemm_group_tn(q0); ops.gemm_group_tn(q1)
        torch.cuda.synchronize()
        assert torch.equal(q0[0]["C"], q1[0]["C"]) and torch.equal(q0[0]["colsum"], q1[0]["colsum"]), "mapped bf16 frame dW differs"
        v0, v1, v2 = timeit(b0, reps=50), timeit(b1, reps=50), timeit(b2, reps=50)
        w0, w1 = timeit(lambda: ops.gemm_group_tn(q0), reps=30), timeit(lambda: ops.gemm_group_tn(q1), reps=30)
        # fp32: the 64-bit form of the mapped projection beside the descriptor form
        l2, _ = ops.gemm_p3_nt_call(X3, W3, M, 256, d, bias=bias, C_out=c1, splitk=sk, a_map=amap, a_map_rows=X.shape[0])
        c1.zero_(); l2()
        torch.cuda.synchronize()
        assert torch.equal(c0, c1), "mapped (descriptor form) frame projection differs"
        t2 = timeit(l2, reps=50)
        print(f"{name:5s} ragged={ragged!s:5s} bf16: frame proj copy {v0:6.1f} / map32 {v1:6.1f} / map64 {v2:6.1f} us;  frame dW copy {w0:6.1f} / in place {w1:6.1f} us;"
              f"  fp32 frame proj map32 {t2:6.1f} us", flush=True)
        print(f"{name:5s} ragged={ragged!s:5s} M={M:6d} K={d}: frame proj copy {t0:6.1f} us / in place {t1:6.1f} us ({t1 / t0:.3f}x);  "
              f"frame dW copy {u0:6.1f} us / in place {u1:6.1f} us ({u1 / u0:.3f}x); store {X.numel() * 10 / 1e9:.2f} GB", flush=True)
