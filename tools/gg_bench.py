"""Grouped weight-gradient GEMM (sdumc_gemm_group_tn) vs the per-layer split-K GEMMs on the C2 shapes of one backward.
usage: python3 tools/gg_bench.py [reps]"""
import sys
import os
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdumc_amd import ops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
legs = sys.argv[2].split(",") if len(sys.argv) > 2 else None
only_new = len(sys.argv) > 3 and sys.argv[3] == "new"
from tools.gg_bench_problems import B, Ta, Tt, Tv, D, dev, g, rn, bits, timeit, flops, frame_problems, key_problems, utt_problems  # noqa: E402,F811


def old_path(ps):
    """the round-2 way: one split-K TN GEMM (auto plan) per problem (segments accumulate)"""
    from sdumc_amd._lib import make_dropout
    for q in ps:
        A, Bm = q["A"], q["B"]
        K, M, N = A.shape[0], A.shape[1], Bm.shape[1]
        kw = {}
        if q.get("bits") is not None:
            d = make_dropout(True, 0, 0.5, 1, N, 1)
            d.bits = q["bits"].data_ptr()
            kw = dict(b_drop=d)
        q.setdefault("C", torch.empty(M, N, device=dev))
        ops.gemm(ops.TN, A, Bm, M, N, K, C_out=q["C"], splitk=0, colsum_a=q.get("colsum"), b_row_mod=q.get("b_row_mod", 0), **kw)
        if q.get("A1") is not None:
            ops.gemm(ops.TN, q["A1"], q["B1"], M, N, q["A1"].shape[0], C_out=q["C"], splitk=0, colsum_a=q.get("colsum"),
                     accumulate=True)


def run(name, mk):
    if legs is not None and name.split()[0] not in legs:
        return
    ps = mk()
    f = flops(ps)
    if only_new:
        t_new = timeit(lambda: ops.gemm_group_tn(ps), reps)
        print(f"{name:28s} {f / 1e9:7.2f} GF  grouped {t_new:8.1f} us = {f / t_new / 1e6:6.1f} TF", flush=True)
        return
    t_old = timeit(lambda: old_path(ps), reps)
    c_old = [q["C"].clone() for q in ps]
    for q in ps:
        q.pop("C", None)
    t_new = timeit(lambda: ops.gemm_group_tn(ps), reps)
    err = max(float((q["C"] - c).abs().max() / c.abs().max()) for q, c in zip(ps, c_old))
    print(f"{name:28s} {f / 1e9:7.2f} GF  per-layer {t_old:8.1f} us = {f / t_old / 1e6:6.1f} TF   grouped {t_new:8.1f} us = "
          f"{f / t_new / 1e6:6.1f} TF   max rel diff {err:.1e}", flush=True)


def to_bf16(ps):
    out = []
    for q in ps:
        r = dict(q)
        for k in ("A", "B", "A1", "B1"):
            if r.get(k) is not None:
                r[k] = r[k].bfloat16()
        r.pop("bits", None)
        r.pop("scale", None)
        out.append(r)
    return out


def old_bf16(ps):
    for q in ps:
        A, Bm = q["A"], q["B"]
        K, M, N = A.shape[0], A.shape[1], Bm.shape[1]
        q.setdefault("C", torch.empty(M, N, device=dev))
        ops.gemm_bf16(ops.TN, A, Bm, M, N, K, C_out=q["C"], splitk=0, colsum_a=q.get("colsum"), b_row_mod=q.get("b_row_mod", 0))
        if q.get("A1") is not None:
            ops.gemm_bf16(ops.TN, q["A1"], q["B1"], M, N, q["A1"].shape[0], C_out=q["C"], splitk=0, colsum_a=q.get("colsum"), accumulate=True)


def run_bf16(name, mk):
    if legs is not None and name.split()[0] not in legs:
        return
    ps = to_bf16(mk())
    f = flops(ps)
    t_old = timeit(lambda: old_bf16(ps), reps)
    c_old = [q["C"].clone() for q in ps]
    for q in ps:
        q.pop("C", None)
    t_new = timeit(lambda: ops.gemm_group_tn(ps), reps)
    err = max(float((q["C"] - c).abs().max() / c.abs().max()) for q, c in zip(ps, c_old))
    print(f"{name:28s} {f / 1e9:7.2f} GF  per-layer {t_old:8.1f} us = {f / t_old / 1e6:6.1f} TF   grouped {t_new:8.1f} us = "
          f"{f / t_new / 1e6:6.1f} TF   max rel diff {err:.1e}", flush=True)


run_bf16("hframe bf16 frame dW", frame_problems)
run_bf16("hkey bf16 key dW 6 sites", lambda: [dict(q, b_row_mod=0, B=torch.cat([q["B"], q["B"]])) for q in key_problems()])
run_bf16("hall bf16 frame + key", lambda: frame_problems() + [dict(q, b_row_mod=0, B=torch.cat([q["B"], q["B"]])) for q in key_problems()])
run("p1 K=24000 N=1024", lambda: [{"A": rn(24000, 256), "B": rn(24000, 1024)}])
run("z1 zeros K=24000 N=1024", lambda: [{"A": torch.zeros(24000, 256, device=dev), "B": torch.zeros(24000, 1024, device=dev)}])
run("o1 ones K=24000 N=1024", lambda: [{"A": torch.ones(24000, 256, device=dev), "B": torch.ones(24000, 1024, device=dev)}])
run("p2 K=3000 N=8192", lambda: [{"A": rn(3000, 256), "B": rn(3000, 8192)}])
run("p3 K=750 N=32768", lambda: [{"A": rn(750, 256), "B": rn(750, 32768)}])
run("p4 K=24000 N=1024 Bmod=512", lambda: [{"A": rn(24000, 256), "B": rn(512, 1024), "b_row_mod": 512}])
run("frame dW (G2)", frame_problems)
run("audio frame dW only", lambda: frame_problems()[:1])
run("key dW, 6 sites (masked)", key_problems)
run("keyca dW CA sites + utt", lambda: key_problems((1,)) + utt_problems()[:12])
run("utt utterance-level dW", utt_problems)
run("all = frame + key + utt", lambda: frame_problems() + key_problems() + utt_problems())
