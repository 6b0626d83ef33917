"""gemm_bf16.hip against torch (fp32 products of the bf16-rounded operands): values and time per launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sdumc_amd import ops
from tools.gemm_wide_check import timeit
NT, TN = ops.NT, ops.TN
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(3)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)


def nt(M, N, K, bias=False, act=ops.ACT_NONE, c_bf16=False, accumulate=False, row_mod=0):
    A = rn(row_mod or M, K).bfloat16()
    B = (rn(N, K) * 0.05).bfloat16()
    b = rn(N) if bias else None
    ref = A.float()[torch.arange(M, device=dev) % (row_mod or M)] @ B.float().t()
    if bias:
        ref = ref + b
    if act == ops.ACT_TANH:
        ref = torch.tanh(ref)
    C0 = (rn(M, N)).to(torch.bfloat16 if c_bf16 else torch.float32) if accumulate else None
    if accumulate:
        ref = ref + C0.float()
    C = C0.clone() if accumulate else torch.empty(M, N, device=dev, dtype=torch.bfloat16 if c_bf16 else torch.float32)
    ops.gemm_bf16(NT, A, B, M, N, K, bias=b, C_out=C, act=act, c_bf16=c_bf16, accumulate=accumulate, a_row_mod=row_mod)
    err = float((C.float() - ref).abs().max() / ref.abs().max())
    us = timeit(lambda: ops.gemm_bf16(NT, A, B, M, N, K, bias=b, C_out=C, act=act, c_bf16=c_bf16, accumulate=accumulate, a_row_mod=row_mod))
    print(f"NT M={M:6d} N={N:5d} K={K:5d} bias={int(bias)} act={act} cbf={int(c_bf16)} acc={int(accumulate)} mod={row_mod}: err {err:.2e}  {us:7.1f} us  {2.0*M*N*K/us/1e6:7.1f} TF", flush=True)


def tn(M, N, K, colsum=False, row_mod=0, splitk=0):
    A = rn(K, M).bfloat16()
    B = rn(row_mod or K, N).bfloat16()
    Bf = B.float()[torch.arange(K, device=dev) % (row_mod or K)]
    ref = A.float().t() @ Bf
    cs = torch.zeros(M, device=dev) if colsum else None
    C = torch.empty(M, N, device=dev)
    ops.gemm_bf16(TN, A, B, M, N, K, C_out=C, colsum_a=cs, b_row_mod=row_mod, splitk=splitk)
    err = float((C - ref).abs().max() / ref.abs().max())
    if colsum:
        err = max(err, float((cs - A.float().sum(0)).abs().max() / A.float().sum(0).abs().max()))
    us = timeit(lambda: ops.gemm_bf16(TN, A, B, M, N, K, C_out=C, colsum_a=cs, b_row_mod=row_mod, splitk=splitk))
    print(f"TN M={M:6d} N={N:5d} K={K:6d} cs={int(colsum)} mod={row_mod} split={splitk}: err {err:.2e}  {us:7.1f} us  {2.0*M*N*K/us/1e6:7.1f} TF", flush=True)


if __name__ == "__main__":
    nt(256, 128, 64)
    nt(300, 256, 128, bias=True)
    nt(24000, 256, 1024, bias=True, c_bf16=True)
    nt(14400, 256, 1024, bias=True, c_bf16=True)
    nt(2048, 256, 4096, bias=True, c_bf16=True)
    nt(48000, 256, 256, bias=True, act=ops.ACT_TANH, c_bf16=True)
    nt(48000, 256, 256, bias=True, act=ops.ACT_TANH, c_bf16=True, row_mod=24000)
    nt(48000, 256, 256, c_bf16=True, accumulate=True)
    nt(28800, 256, 256, bias=True, act=ops.ACT_TANH, c_bf16=True)
    nt(4096, 4096, 4096)
    tn(128, 128, 64)
    tn(256, 256, 1000, colsum=True)
    tn(256, 1024, 24000, colsum=True)
    tn(256, 1024, 24007, colsum=True)
    tn(256, 4096, 2048, colsum=True)
    tn(256, 256, 48000, colsum=True)
    tn(256, 256, 48000, colsum=True, row_mod=24000)
    tn(4096, 4096, 4096, splitk=1)
