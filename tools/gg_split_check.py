"""The grouped weight-gradient launch with fp32 products on the bf16 matrix pipe (three-way operand split, six MFMA terms)
against the same launch on v_mfma_f32_32x32x2_f32: error of both against an fp64 product of the same operands, and time, on
the C2 problems of one backward.  usage: python3 tools/gg_split_check.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from sdumc_amd import ops, _lib  # noqa: E402
import tools.gg_bench_problems as GP  # noqa: E402

dev = "cuda"


def ref64(q):
    A, B = q["A"].double(), q["B"].double()
    K = A.shape[0]
    if q.get("b_row_mod"):
        B = B.repeat((K + B.shape[0] - 1) // B.shape[0], 1)[:K]
    if q.get("bits") is not None:
        bits = q["bits"]
        cols = torch.arange(B.shape[1], device=dev)
        keep = ((bits[:, cols // 4].int() >> (cols % 4)) & 1).double()
        B = B * keep * q["scale"]
    C = A.t() @ B
    if q.get("A1") is not None:
        C = C + q["A1"].double().t() @ q["B1"].double()
    cs = A.sum(0) + (q["A1"].double().sum(0) if q.get("A1") is not None else 0)
    return C, cs


def errs(ps, refs):
    worst, worst_cs = 0.0, 0.0
    for q, (C, cs) in zip(ps, refs):
        # against the fp64 product: largest error over the entries, relative to what an entry's terms add up to in magnitude
        scale = float(C.abs().max())
        worst = max(worst, float((q["C"].double() - C).abs().max()) / scale)
        if q.get("colsum") is not None:
            worst_cs = max(worst_cs, float((q["colsum"].double() - cs).abs().max()) / float(cs.abs().max()))
    return worst, worst_cs


for name, mk in (("frame dW", GP.frame_problems), ("key dW, 6 sites (masked)", GP.key_problems), ("utterance-level dW", GP.utt_problems),
                 ("all = frame + key + utt", lambda: GP.frame_problems() + GP.key_problems() + GP.utt_problems())):
    ps = mk()
    refs = [ref64(q) for q in ps]
    f = GP.flops(ps)
    out = []
    res = {}
    for split in (0, 1, 0, 1):
        _lib.lib.sdumc_set_split_(15 if split else 0)
        for q in ps:
            q.pop("C", None)
            if q.get("colsum") is not None:
                q["colsum"].zero_()
        ops.gemm_group_tn(ps)
        torch.cuda.synchronize()
        e, ecs = errs(ps, refs)
        t = GP.timeit(lambda: ops.gemm_group_tn(ps), 20)
        res.setdefault(split, []).append((t, e, ecs))
    for split in (0, 1):
        t = min(r[0] for r in res[split])
        e, ecs = res[split][0][1], res[split][0][2]
        print(f"{name:28s} {f / 1e9:7.2f} GF  {'split bf16 x6' if split else 'fp32 MFMA    '} {t:8.1f} us = {f / t / 1e6:6.1f} TF   "
              f"max |C - fp64| / max |C| {e:.2e}   column sums {ecs:.2e}", flush=True)
_lib.lib.sdumc_set_split_(int(os.environ.get("SDUMC_SPLIT", 15)))
