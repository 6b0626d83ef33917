"""csrc/gemm_p3.hip: fp32 GEMMs on operands split ONCE PER TENSOR into three bf16 planes (include/sdumc_hip.h: sdumc_gemm_p3).

Replaces, in the fp32 step, F.linear of frame_dim_reshape_{0,1,2} (model :193-195, :282-284) and of the input_proj key projections
(model :60, :82) that the wide NT kernel used to take.  Same arithmetic as the in-kernel split (six exact bf16 x bf16 part products
per fp32 product, fp32 accumulation), so the same bar: as close to an fp64 product of the same operands as the fp32-MFMA kernel."""
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


def err(got, ref):
    return float((got.double() - ref).abs().max() / ref.abs().max())


def test_split_join_round_trip_is_bit_exact(env):
    ops, _ = env
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(777, 1024, device="cuda", generator=g) * torch.exp2(torch.randint(-60, 60, (777, 1024), device="cuda", generator=g).float())
    x[0, :8] = torch.tensor([0.0, -0.0, 1.0, -1.0, 1.0 + 2.0 ** -23, 16777215.0, 3.0e38, 2.0 ** -100], device="cuda")
    p = ops.p3_split(x)
    assert p.shape == (777, 6 * 1024) and p.dtype == torch.uint8
    back = ops.p3_join(p, 1024)
    assert torch.equal(back, x), "planes -> fp32 must reproduce every value"
    nz = x != 0
    assert torch.equal(back.view(torch.int32)[nz], x.view(torch.int32)[nz])      # bit for bit; only -0 comes back as +0 (its planes are -0, +0, +0)
    assert float(back[0, 1]) == 0.0
    # the planes are what tests/test_split_arithmetic.py's split3 computes: plane 0 = bf16(x) (round to nearest even)
    planes = p.view(torch.bfloat16).view(777, 128, 3, 8)
    assert torch.equal(planes[:, :, 0, :].reshape(777, 1024), x.to(torch.bfloat16))
    r1 = x - planes[:, :, 0, :].reshape(777, 1024).float()
    assert torch.equal(planes[:, :, 1, :].reshape(777, 1024), r1.to(torch.bfloat16))


@pytest.mark.parametrize("tile_m", [0, 64, 96, 128])
def test_frame_projection_against_fp64(env, tile_m):
    """ragged M, bias, K = 1024: every tile form, with and without K split over workgroups; the P3 copy of the output is the split of
    the fp32 output; results are bit-identical from run to run"""
    ops, _ = env
    g = torch.Generator(device="cuda").manual_seed(5)
    rn = lambda *s: torch.randn(*s, device="cuda", generator=g)
    M, K = 9003, 1024
    X, W, b = rn(M, K), rn(256, K) / K ** 0.5, rn(256)
    ref = X.double() @ W.double().t() + b.double()
    X3, W3 = ops.p3_split(X), ops.p3_split_frag(W)
    f32 = ops.gemm(ops.NT, X, W, M, 256, K, bias=b, tile=14, splitk=1)      # the in-kernel split form (wide tiles)
    e_ref = err(f32, ref)
    for sk in (1, 4):
        c, c3 = ops.gemm_p3_nt(X3, W3, M, 256, K, bias=b, tile_m=tile_m, splitk=sk, want_p3=True)
        e = err(c, ref)
        assert e < ABS_BOUND and e <= 1.5 * e_ref + 1e-7, (tile_m, sk, e, e_ref)
        assert torch.equal(ops.p3_join(c3, 256), c)
        again = ops.gemm_p3_nt(X3, W3, M, 256, K, bias=b, tile_m=tile_m, splitk=sk)
        assert torch.equal(again, c)


@pytest.mark.parametrize("tile_m", [0, 96, 128])
def test_key_projection_with_fused_dropout_against_fp64(env, tile_m):
    """keep-bits on A (all three planes masked), row modulo (the two streams share x), bias, tanh, K = 256"""
    ops, _lib = env
    g = torch.Generator(device="cuda").manual_seed(7)
    rn = lambda *s: torch.randn(*s, device="cuda", generator=g)
    M2, mod = 36864 + 32, 18432 + 16
    x, W2, b2 = rn(mod, 256), rn(256, 256) / 16, rn(256) * 0.1
    d = _lib.make_dropout(True, 3, 0.5, M2, 256, 1, seed=77)
    bits = ops.dropout_bits(d, 1).view(M2, 64)
    mask = ops.dropout_mask(d, 1).view(M2, 256).double()
    ref = torch.tanh((x.double().repeat(2, 1) * mask) @ W2.double().t() + b2.double())
    c = ops.gemm_p3_nt(ops.p3_split(x), ops.p3_split_frag(W2), M2, 256, 256, bias=b2, act=ops.ACT_TANH, a_row_mod=mod, bits=bits, scale=2.0,
                       tile_m=tile_m)
    assert err(c, ref) < ABS_BOUND


def test_two_a_tensors_in_one_launch(env):
    """the text slot: the two streams' features are separate tensors, their projections adjacent rows of one x (K = 4096, split K)"""
    ops, _ = env
    g = torch.Generator(device="cuda").manual_seed(9)
    rn = lambda *s: torch.randn(*s, device="cuda", generator=g)
    K = 4096
    A0, A1, W, b = rn(2048, K), rn(1024, K), rn(256, K) / K ** 0.5, rn(256)
    ref = torch.cat([A0, A1]).double() @ W.double().t() + b.double()
    W3 = ops.p3_split_frag(W)
    for sk in (1, 4):
        c = ops.gemm_p3_nt(ops.p3_split(A0), W3, 3072, 256, K, bias=b, splitk=sk, A3_second=ops.p3_split(A1), second_row0=2048)
        assert err(c, ref) < ABS_BOUND
    one = ops.gemm_p3_nt(ops.p3_split(torch.cat([A0, A1])), W3, 3072, 256, K, bias=b, splitk=4)
    assert torch.equal(one, c), "two tensors must give the bits of the concatenated one"


def test_edges_of_the_split_on_every_kernel_family(env):
    """include/sdumc_hip.h, CONTRACT AT THE EDGES: a NaN operand gives NaN wherever its row / column feeds; +-Inf and finite values
    that round to Inf in bf16 (|x| >= 0x1.FEp127) give NaN too (the fp32 MFMAs keep an infinity); every other output stays right."""
    ops, _lib = env
    lib = _lib.lib
    g = torch.Generator(device="cuda").manual_seed(11)
    rn = lambda *s: torch.randn(*s, device="cuda", generator=g)
    big = float(torch.tensor([0x7F7F8000], dtype=torch.int32).view(torch.float32))      # rounds to Inf in its first plane
    M, K = 512, 256
    X, W = rn(M, K), rn(256, K) / 16
    X[3, 5], X[7, 9], X[11, 200], X[15, 17] = float("nan"), float("inf"), -float("inf"), big
    bad = [3, 7, 11, 15]
    good = [r for r in range(M) if r not in bad]
    ref = X[good].double() @ W.double().t()
    # (1) operands split once per tensor
    c = ops.gemm_p3_nt(ops.p3_split(X), ops.p3_split_frag(W), M, 256, K)
    assert torch.isnan(c[bad]).all() and err(c[good], ref) < ABS_BOUND
    # (2) wide NT tiles, in-kernel split; and the same call on the fp32 MFMAs: NaN stays NaN, the infinities stay infinities
    c = ops.gemm(ops.NT, X, W, M, 256, K, tile=14, splitk=1)
    assert torch.isnan(c[bad]).all() and err(c[good], ref) < ABS_BOUND
    try:
        lib.sdumc_set_split_(0)
        c0 = ops.gemm(ops.NT, X, W, M, 256, K, tile=14, splitk=1)
    finally:
        lib.sdumc_set_split_(int(__import__("os").environ.get("SDUMC_SPLIT", "15")))
    assert torch.isnan(c0[3]).all() and torch.isinf(c0[7]).all() and torch.isinf(c0[11]).all() and not torch.isnan(c0[[7, 11]]).any()
    assert err(c0[good], ref) < ABS_BOUND
    # (3) the rows launch (dxd += dz W): A = X
    Wk = rn(256, 256) / 16
    c = ops.gemm_rows256([{"A": X, "B": Wk, "M": M}])[0]
    assert torch.isnan(c[bad]).all() and err(c[good], X[good].double() @ Wk.double()) < ABS_BOUND
    # (4) the grouped weight gradients (contraction over the rows): a bad row of A poisons its output ROW m, a bad row of B column n
    A, B = rn(700, 64), rn(700, 128)
    A[100, 3], A[200, 9], B[300, 17] = float("inf"), float("nan"), big
    q = [{"A": A, "B": B}]
    ops.gemm_group_tn(q)
    C = q[0]["C"]
    ok_r = [r for r in range(64) if r not in (3, 9)]
    ok_c = [c_ for c_ in range(128) if c_ != 17]
    assert torch.isnan(C[3]).all() and torch.isnan(C[9]).all() and torch.isnan(C[:, 17]).all()
    Ad, Bd = A.double().clone(), B.double().clone()
    refC = Ad[:, ok_r].t() @ Bd[:, ok_c]
    assert err(C[ok_r][:, ok_c], refC) < ABS_BOUND


def test_step_on_feature_planes_equals_the_step_on_fp32_features(env):
    """One C2-shaped train step (B = 8) with the frame / key projections on planes (the default) against the same step with
    planes=False (the in-kernel split): two fp32 evaluations at different summation orders -- losses 1e-5, every gradient tensor 2e-4 of
    its largest entry (the bar of tests/test_gpu_split.py::test_train_step_split_against_fp32_mfma); and the planes step is bit-reproducible."""
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
    feats = [torch.randn(B, Tn[i], dims[i], device="cuda", generator=g) for i in range(4)]
    vals = torch.rand(B, device="cuda", generator=g) * 6 - 3
    res = {}
    for planes in (True, False, True):
        ts = E.TrainStep(flat.clone(), B, Tn, dims, seed=5, planes=planes)
        assert (ts._planes is not None) == planes
        ts.set_batch(*feats, vals)
        losses = ts.run().cpu().clone()
        res.setdefault(planes, []).append((losses, ts.grads.clone().cpu()))
        del ts
    assert torch.equal(res[True][0][0], res[True][1][0]) and torch.equal(res[True][0][1], res[True][1][1])
    np.testing.assert_allclose(res[True][0][0].numpy(), res[False][0][0].numpy(), rtol=1e-5, atol=1e-6)
    differ = 0
    for k in lay.live_names():
        off, shape, _ = lay.entries[k]
        n = int(np.prod(shape))
        a, b = res[False][0][1][off:off + n].double(), res[True][0][1][off:off + n].double()
        d, m = float((a - b).abs().max()), float(a.abs().max())
        differ += d > 0
        assert d <= 2e-4 * m + 1e-8, f"{k}: |planes - in-kernel split| {d:.2e} at max |g| {m:.2e}"
    assert differ > 0, "the two steps are bit-identical -- did the planes path run?"
