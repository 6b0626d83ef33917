"""Isolate the cost of the fusions on the key-projection GEMM shape (M=48000, N=256, K=256)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sdumc_amd import ops
from sdumc_amd._lib import make_dropout

def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps

B, T, D, S = 64, 375, 256, 2
M = S * B * T
x = torch.randn(B * T, D, device="cuda"); x2 = torch.randn(M, D, device="cuda")
W = torch.randn(D, D, device="cuda") / 16; b = torch.randn(D, device="cuda")
C = torch.empty(M, D, device="cuda")
d = make_dropout(True, 21, 0.5, T, D, B, call0=3, seed=9)
db = make_dropout(True, 21, 0.5, T, D, B, call0=3, seed=9); bits = ops.dropout_bits(db, S)
flops = 2.0 * M * D * D
for name, fn in [
    ("plain NT", lambda: ops.gemm(ops.NT, x2, W, M, D, D, C_out=C, splitk=1)),
    ("+bias", lambda: ops.gemm(ops.NT, x2, W, M, D, D, bias=b, C_out=C, splitk=1)),
    ("+bias+tanh", lambda: ops.gemm(ops.NT, x2, W, M, D, D, bias=b, act=ops.ACT_TANH, C_out=C, splitk=1)),
    ("+row_mod", lambda: ops.gemm(ops.NT, x, W, M, D, D, C_out=C, a_row_mod=B * T, splitk=1)),
    ("+row_mod+philox", lambda: ops.gemm(ops.NT, x, W, M, D, D, C_out=C, a_row_mod=B * T, a_drop=d, splitk=1)),
    ("+row_mod+bits", lambda: ops.gemm(ops.NT, x, W, M, D, D, C_out=C, a_row_mod=B * T, a_drop=db, splitk=1)),
    ("all (bits,tanh)", lambda: ops.gemm(ops.NT, x, W, M, D, D, bias=b, act=ops.ACT_TANH, C_out=C, a_row_mod=B * T, a_drop=db, splitk=1)),
    ("bf16 plain NT", lambda: ops.gemm(ops.NT, x2, W, M, D, D, C_out=C, splitk=1, bf16=True)),
    ("bf16 NT tile128", lambda: ops.gemm(ops.NT, x2, W, M, D, D, C_out=C, splitk=1, bf16=True, tile=1)),
    ("bf16 all (bits,tanh)", lambda: ops.gemm(ops.NT, x, W, M, D, D, bias=b, act=ops.ACT_TANH, C_out=C, a_row_mod=B * T, a_drop=db, splitk=1, bf16=True)),
    ("NN plain", lambda: ops.gemm(ops.NN, x2, W, M, D, D, C_out=C, splitk=1)),
    ("NN accumulate", lambda: ops.gemm(ops.NN, x2, W, M, D, D, C_out=C, splitk=1, accumulate=True)),
    ("TN plain s=0", lambda: ops.gemm(ops.TN, x2, C, D, D, M, splitk=0)),
    ("TN bits s=0", lambda: ops.gemm(ops.TN, x2, x, D, D, M, splitk=0, b_row_mod=B * T, b_drop=db)),
    ("TN philox s=0", lambda: ops.gemm(ops.TN, x2, x, D, D, M, splitk=0, b_row_mod=B * T, b_drop=d)),
    ("TN bits+colsum", lambda: ops.gemm(ops.TN, x2, x, D, D, M, splitk=0, b_row_mod=B * T, b_drop=db, colsum_a=b)),
]:
    us = t(fn)
    print(f"{name:22s} {us:8.1f} us {flops / us / 1e6:7.1f} TF", flush=True)
