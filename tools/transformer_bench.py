#!/usr/bin/env python3
"""Times one TransformerEncoderLayer (sdumc_amd.transformers_encoder) forward and forward+backward at the
per-rank shape of BASELINE configs[4] (C5: T=512, E=1024, H=8, batch 256 over 8 GPUs = 32 per rank) and prints
one JSON line with the algorithmic FLOP rate against the fp32 MFMA peak.  Secondary measurement (SURVEY §8a
row A11 is not on the SDUMC step); the headline number stays bench.py's.

  python tools/transformer_bench.py [--T 512 --B 32 --E 1024 --H 8 --steps 10 --no-dropout]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from sdumc_amd import _lib, transformers_encoder as te  # noqa: E402

PEAK_F32_TFLOPS = 157.3   # MI355X dense fp32 MFMA (MI355X_MICROARCH.md)
PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--T", type=int, default=512)
    ap.add_argument("--B", type=int, default=32)
    ap.add_argument("--E", type=int, default=1024)
    ap.add_argument("--H", type=int, default=8)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-dropout", action="store_true")
    ap.add_argument("--bf16", action="store_true", help="bf16 operands / fp32 accumulate in every product (BASELINE configs[4] dtype)")
    a = ap.parse_args()
    p = 0.0 if a.no_dropout else 0.1
    te.set_bf16(a.bf16)
    torch.manual_seed(0)
    lay = te.TransformerEncoderLayer(a.E, num_heads=a.H, attn_dropout=p, relu_dropout=p, res_dropout=p,
                                     attn_mask=True).cuda().train()
    x = torch.randn(a.T, a.B, a.E, device="cuda", requires_grad=True)
    R = torch.randn(a.T, a.B, a.E, device="cuda")
    M = a.T * a.B
    fwd_flops = 24.0 * M * a.E * a.E + 4.0 * M * a.T * a.E

    def fwd():
        return lay(x)

    def fwd_bwd():
        y = lay(x)
        y.backward(R)
        x.grad = None
        for q in lay.parameters():
            q.grad = None

    def timeit(fn):
        for _ in range(a.warmup):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.steps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.steps

    with torch.no_grad():
        ms_f = timeit(fwd)
    ms_fb = timeit(fwd_bwd)
    # per-variant GEMM time from the library's own HIP-event profiler (one extra step)
    _lib.lib.sdumc_profile_enable(1)
    fwd_bwd()
    torch.cuda.synchronize()
    ent = (_lib.ProfEntry * 32)()
    n = _lib.lib.sdumc_profile_report(ent, 32)
    gemms = {ent[i].name.decode(): {"launches": ent[i].launches, "ms": round(ent[i].total_ms, 4),
                                    "tflops": round(ent[i].total_flops / (ent[i].total_ms * 1e9), 2)}
             for i in range(n) if ent[i].launches}
    _lib.lib.sdumc_profile_enable(0)
    print(json.dumps({
        "what": "TransformerEncoderLayer fwd / fwd+bwd", "T": a.T, "B": a.B, "E": a.E, "H": a.H, "dropout": p,
        "fwd_ms": round(ms_f, 4), "fwd_bwd_ms": round(ms_fb, 4),
        "fwd_tflops": round(fwd_flops / (ms_f * 1e9), 2), "fwd_bwd_tflops": round(3 * fwd_flops / (ms_fb * 1e9), 2),
        "peak_tflops": PEAK_BF16_TFLOPS if a.bf16 else PEAK_F32_TFLOPS,
        "fwd_bwd_frac": round(3 * fwd_flops / (ms_fb * 1e9) / (PEAK_BF16_TFLOPS if a.bf16 else PEAK_F32_TFLOPS), 3),
        "tokens_per_s": round(M / (ms_fb * 1e-3)), "dtype": "bf16 operands / f32 accumulate, f32 storage" if a.bf16 else "f32", "gemm_variants": gemms}))


if __name__ == "__main__":
    main()
