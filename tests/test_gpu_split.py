import os
"""fp32 products on the bf16 matrix pipe (sdumc_hip.h: sdumc_set_split_; csrc/gemm_group.hip has the arithmetic).

Every fp32 GEMM kernel of the frame-level part splits its operands exactly into three bf16 parts and accumulates six of the
nine part products -- each exact in fp32 -- with v_mfma_f32_32x32x16_bf16 instead of eight v_mfma_f32_32x32x2_f32.  The claim
tested here, kernel family by kernel family and on the whole step: the results are AS CLOSE TO AN FP64 PRODUCT of the same
operands as those of the fp32-MFMA kernels (the error of a product, < 2^-23 |a b|, is the size of one fp32 rounding), i.e. this is
fp32 arithmetic at a different summation order, not reduced precision.  Bounds: both forms within 3e-6 of fp64 (relative to the
largest entry), the split form within 1.5x the fp32-MFMA form's own error + 1e-7."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ABS_BOUND = 3e-6


@pytest.fixture(scope="module")
def env():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from sdumc_amd import ops, _lib
    return ops, _lib


def both(lib, run):
    """run() under sdumc_set_split_(0) and (15): [result tensors] of each"""
    out = {}
    try:
        for mask in (0, 15):
            lib.sdumc_set_split_(mask)
            out[mask] = [t.clone() for t in run()]
            torch.cuda.synchronize()
    finally:
        lib.sdumc_set_split_(int(os.environ.get("SDUMC_SPLIT", 15)))
    return out[0], out[15]


def err(got, ref):
    return float((got.double() - ref).abs().max() / ref.abs().max())


def judge(f32, split, refs, what):
    for i, (a, b, r) in enumerate(zip(f32, split, refs)):
        ea, eb = err(a, r), err(b, r)
        assert ea < ABS_BOUND and eb < ABS_BOUND, f"{what}[{i}]: fp32 MFMA {ea:.2e}, split {eb:.2e} against fp64"
        assert eb <= 1.5 * ea + 1e-7, f"{what}[{i}]: split {eb:.2e} against fp64 is not as close as the fp32 MFMAs' {ea:.2e}"
        assert not torch.equal(a, b) or a.numel() < 64, f"{what}[{i}]: the two forms are bit-identical -- is the switch connected?"


def test_grouped_weight_gradients(env):
    """sdumc_gemm_group_tn: frame-like (two K segments, column sums), key-like (fused dropout on B, row modulo), a ragged
    utterance-level problem, an accumulating one -- stream-K partials and the ordered reduce included."""
    ops, _lib = env
    g = torch.Generator(device="cuda").manual_seed(3)
    rn = lambda *s: torch.randn(*s, device="cuda", generator=g)
    K = 6000
    bits = torch.randint(0, 16, (2 * K, 64), device="cuda", generator=g, dtype=torch.uint8)
    C_acc = rn(128, 320)
    mk = lambda: [
        {"A": rn(K, 256), "B": rn(K, 1024), "A1": rn(1000, 256), "B1": rn(1000, 1024), "colsum": torch.zeros(256, device="cuda")},
        {"A": rn(2 * K, 256), "B": rn(K, 256), "b_row_mod": K, "bits": bits, "scale": 2.0, "colsum": torch.zeros(256, device="cuda")},
        {"A": rn(900, 128), "B": rn(900, 320), "C": C_acc.clone(), "accumulate": True},
        {"A": rn(131, 64), "B": rn(131, 128), "colsum": torch.zeros(64, device="cuda")},
    ]
    g.manual_seed(3)
    ps = mk()
    refs = []
    for q in ps:
        A, B = q["A"].double(), q["B"].double()
        if q.get("b_row_mod"):
            B = B.repeat(2, 1)
        if q.get("bits") is not None:
            cols = torch.arange(B.shape[1], device="cuda")
            B = B * ((q["bits"][:, cols // 4].int() >> (cols % 4)) & 1).double() * q["scale"]
        C = A.t() @ B
        cs = A.sum(0)
        if q.get("A1") is not None:
            C = C + q["A1"].double().t() @ q["B1"].double()
            cs = cs + q["A1"].double().sum(0)
        if q.get("accumulate"):
            C = C + C_acc.double()
        refs.append((C, cs))

    def run():
        for q in ps:
            if q.get("accumulate"):
                q["C"] = C_acc.clone()
            else:
                q.pop("C", None)
            if q.get("colsum") is not None:
                q["colsum"].zero_()
        ops.gemm_group_tn(ps)
        return [q["C"] for q in ps] + [q["colsum"] for q in ps if q.get("colsum") is not None]

    f32, split = both(_lib.lib, run)
    all_refs = [r[0] for r in refs] + [r[1] for q, r in zip(ps, refs) if q.get("colsum") is not None]
    judge(f32[:4], split[:4], all_refs[:4], "dW")
    for a, b, r in zip(f32[4:], split[4:], all_refs[4:]):      # the column sums are fp32 VALU sums in both forms
        assert err(a, r) < ABS_BOUND and err(b, r) < ABS_BOUND


def test_wide_tile_projections(env):
    """sdumc_gemm, NT wide tiles: a frame projection (ragged M, bias) and a key projection (fused dropout on A through keep-bits,
    row modulo, bias, tanh)."""
    ops, _lib = env
    g = torch.Generator(device="cuda").manual_seed(5)
    rn = lambda *s: torch.randn(*s, device="cuda", generator=g)
    M1, K1 = 9003, 1024
    X, W1, b1 = rn(M1, K1), rn(256, K1) / K1 ** 0.5, rn(256)
    ref1 = X.double() @ W1.double().t() + b1.double()
    M2, mod = 36864, 18432
    x, W2, b2 = rn(mod, 256), rn(256, 256) / 16, rn(256) * 0.1
    d = _lib.make_dropout(True, 3, 0.5, M2, 256, 1, seed=77)
    bits = ops.dropout_bits(d, 1)
    ref2 = torch.tanh((x.double().repeat(2, 1) * ops.dropout_mask(d, 1).view(M2, 256).double()) @ W2.double().t() + b2.double())

    def run():
        c1 = ops.gemm(ops.NT, X, W1, M1, 256, K1, bias=b1, tile=14, splitk=1)
        c2 = ops.gemm(ops.NT, x, W2, M2, 256, 256, bias=b2, act=ops.ACT_TANH, a_row_mod=mod, a_drop=d, ab_drop_bits=[bits], tile=13, splitk=1)
        return [c1, c2]

    f32, split = both(_lib.lib, run)
    judge(f32, split, [ref1, ref2], "projection")


def test_rows_launch(env):
    """sdumc_gemm_rows256: the masked forward form (bias, tanh, row modulo) and the accumulating dX form, ragged M."""
    ops, _lib = env
    g = torch.Generator(device="cuda").manual_seed(7)
    rn = lambda *s: torch.randn(*s, device="cuda", generator=g)
    M, mod = 9000, 4500
    x, W, b = rn(mod, 256) * 0.5, rn(256, 256) / 16, rn(256)
    keep = torch.rand(M, 256, device="cuda", generator=g) >= 0.5
    bits = (keep.view(M, 64, 4).int() * torch.tensor([1, 2, 4, 8], device="cuda", dtype=torch.int32)).sum(-1).to(torch.uint8)
    ref_f = torch.tanh((x.double().repeat(2, 1) * keep.double() * 2.0) @ W.double() + b.double())
    M2 = 5003
    dz, C0 = rn(M2, 256), rn(M2, 256)
    ref_b = C0.double() + dz.double() @ W.double()

    def run():
        f = ops.gemm_rows256([{"A": x, "B": W, "M": M, "a_row_mod": mod, "bits": bits, "scale": 2.0, "bias": b, "act": ops.ACT_TANH}])[0]
        c = C0.clone()
        ops.gemm_rows256([{"A": dz, "B": W, "C": c, "accumulate": True}])
        return [f, c]

    f32, split = both(_lib.lib, run)
    judge(f32, split, [ref_f, ref_b], "rows")


def test_train_step_split_against_fp32_mfma(env):
    """One C2-shaped step (B = 8) with every product on the bf16 matrix pipe against the same step on the fp32 MFMAs: losses to
    1e-5, every live gradient tensor to 2e-4 of its largest entry (measured: ~1e-6; two fp32 evaluations of the same graph at different
    summation orders: the bound the fp32 step meets against the fp64 oracle elsewhere is 1e-3)."""
    ops, _lib = env
    from sdumc_amd import engine as E
    from oracle import sdumc_oracle as O
    dims, B, Tn = (1024, 4096, 1024, 4096), 8, (375, 32, 225, 32)
    P = O.init_params(dims, seed=0)
    lay = E.ParamLayout.get(*dims[:3])
    flat = torch.zeros(lay.total)
    for k, v in lay.views(flat).items():
        v.copy_(P[k])
    flat = flat.cuda()
    g = torch.Generator(device="cuda").manual_seed(11)
    audio, text, video, feat4 = [torch.randn(B, Tn[i], dims[i], device="cuda", generator=g) for i in range(4)]
    vals = torch.rand(B, device="cuda", generator=g) * 6 - 3
    res = {}
    try:
        for mask in (0, 15):
            _lib.lib.sdumc_set_split_(mask)
            ts = E.TrainStep(flat.clone(), B, Tn, dims, seed=5)
            ts.set_batch(audio, text, video, feat4, vals)
            losses = ts.run().cpu().clone()
            res[mask] = (losses, ts.grads.clone().cpu())
            del ts
    finally:
        _lib.lib.sdumc_set_split_(int(os.environ.get("SDUMC_SPLIT", 15)))
    np.testing.assert_allclose(res[15][0].numpy(), res[0][0].numpy(), rtol=1e-5, atol=1e-6)
    worst = 0.0
    for k in lay.live_names():
        off, shape, _ = lay.entries[k]
        n = int(np.prod(shape))
        a, b = res[0][1][off:off + n].double(), res[15][1][off:off + n].double()
        # (absolute floor 1e-8 beside gradient entries up to ~1: orgin_linear_change.2.bias feeds an L2 normalisation, its gradient
        #  is zero in exact arithmetic and 7e-10 of rounding residue in either form)
        d, m = float((a - b).abs().max()), float(a.abs().max())
        worst = max(worst, d / max(m, 1e-12))
        assert d <= 2e-4 * m + 1e-8, f"{k}: |split - fp32 MFMA| {d:.2e} at max |g| {m:.2e}"
    assert worst > 0.0, "the two steps are bit-identical -- is the switch connected?"
