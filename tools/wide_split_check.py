"""gemm_wide.hip's NT launches with fp32 products on the bf16 matrix pipe (three-way operand split, six MFMA terms) against the
same launches on v_mfma_f32_32x32x2_f32: error of both against an fp64 product of the same operands, and time, on the forward
shapes of a C2 step.  usage: python3 tools/wide_split_check.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from sdumc_amd import ops, _lib  # noqa: E402
from tools.gemm_wide_check import timeit  # noqa: E402

dev = "cuda"
g = torch.Generator(device=dev).manual_seed(3)


def case(name, M, N, K, tile, tanh=False, drop=False, row_mod=0, splitk=0):
    A = torch.randn(row_mod or M, K, device=dev, generator=g)
    B = torch.randn(N, K, device=dev, generator=g) / K ** 0.5
    bias = torch.randn(N, device=dev, generator=g)
    kw = dict(act=ops.ACT_TANH if tanh else ops.ACT_NONE)
    Ad = A.double()
    if row_mod:
        kw["a_row_mod"] = row_mod
        Ad = Ad.repeat(M // row_mod, 1)
    if drop:
        d = _lib.make_dropout(True, 3, 0.5, M, K, 1, seed=77)
        bits = ops.dropout_bits(d, 1)
        kw["a_drop"] = d
        kw["ab_drop_bits"] = bits
        Ad = Ad * ops.dropout_mask(d, 1).view(M, K).double()
    ref = Ad @ B.double().t() + bias.double()
    if tanh:
        ref = torch.tanh(ref)
    C = torch.empty(M, N, device=dev)
    res = {}
    for split in (0, 1, 0, 1):
        _lib.lib.sdumc_set_split_(15 if split else 0)
        run = lambda: ops.gemm(ops.NT, A, B, M, N, K, bias=bias, C_out=C, tile=tile, splitk=splitk, **kw)
        run()
        torch.cuda.synchronize()
        err = float((C.double() - ref).abs().max() / ref.abs().max())
        res.setdefault(split, []).append((timeit(run), err))
    fl = 2.0 * M * N * K
    for split in (0, 1):
        t = min(r[0] for r in res[split])
        print(f"{name:28s} M={M:6d} K={K:5d} tile {tile} {'split bf16 x6' if split else 'fp32 MFMA    '} {t:7.1f} us = {fl / t / 1e6:6.1f} TF   "
              f"max |C - fp64| / max |C| {res[split][0][1]:.2e}", flush=True)
    _lib.lib.sdumc_set_split_(int(os.environ.get("SDUMC_SPLIT", 15)))


for t in (11, 12, 13):
    case("frame proj audio", 24000, 256, 1024, t)
    case("frame proj video", 14400, 256, 1024, t)
case("keys video (CA site)", 28800, 256, 256, 14, tanh=True, drop=True, row_mod=14400)
case("keys audio (CA site)", 48000, 256, 256, 14, tanh=True, drop=True, row_mod=24000)
case("keys audio (CA site)", 48000, 256, 256, 12, tanh=True, drop=True, row_mod=24000)
case("frame proj audio", 24000, 256, 1024, 14)
case("frame proj video", 14400, 256, 1024, 14)
case("frame proj text (split-K 8)", 4096, 256, 4096, 14, splitk=8)
case("keys audio (CA site)", 48000, 256, 256, 13, tanh=True, drop=True, row_mod=24000)
case("keys video (CA site)", 28800, 256, 256, 13, tanh=True, drop=True, row_mod=14400)
