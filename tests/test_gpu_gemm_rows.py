"""GPU parity tests of the persistent B-stationary rows GEMM (sdumc_gemm_rows256, csrc/gemm_rows.hip): the key projections
keys = tanh(drop(x) W^T + b) of FRA2UTT_new / Cross_Attention (model :60, :82) and their input gradients dxd += dz W under
loss.backward() (main :149).  Checked against fp64 matmuls at 2e-5 and for bit-identical repeats."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from sdumc_amd import ops as o
    return o


def close(got, want, tol=2e-5, msg=""):
    got = got.detach().cpu().double().numpy()
    want = want.detach().cpu().double().numpy()
    scale = max(1.0, np.abs(want).max())
    np.testing.assert_allclose(got, want, rtol=tol, atol=tol * scale, err_msg=msg)


def keep_bits(M, g, p=0.5):
    """uint8 [M, 64], bit e of byte q = keep column 4q + e (as sdumc_dropout_bits writes them); and the 0/1 mask."""
    m = torch.rand(M, 256, generator=g) >= p
    b = (m.reshape(M, 64, 4).to(torch.int32) * torch.tensor([1, 2, 4, 8], dtype=torch.int32)).sum(-1).to(torch.uint8)
    return b, m.double()


def ref_of(q, mask, C0):
    A, B = q["A"].cpu().double(), q["B"].cpu().double()
    M = q.get("M", A.shape[0])
    mod = q.get("a_row_mod", 0)
    rows = torch.arange(M) % mod if mod else torch.arange(M)
    Av = A[rows]
    if mask is not None:
        Av = Av * mask * q.get("scale", 1.0)
    out = Av @ B
    if q.get("bias") is not None:
        out = out + q["bias"].cpu().double()
    if q.get("accumulate"):
        out = out + C0.cpu().double()
    if q.get("act", 0) == 2:
        out = torch.tanh(out)
    return out


def run_case(ops, specs, seed, repeat=False):
    g = torch.Generator().manual_seed(seed)
    probs, masks, c0s = [], [], []
    for sp in specs:
        M, mod = sp["M"], sp.get("mod", 0)
        q = {"A": (torch.randn(mod or M, 256, generator=g) * 0.5).cuda(), "B": (torch.randn(256, 256, generator=g) / 16).cuda(),
             "M": M, "a_row_mod": mod, "act": sp.get("act", 0)}
        if sp.get("bias"):
            q["bias"] = torch.randn(256, generator=g).cuda()
        mask = None
        if sp.get("mask"):
            bits, mask = keep_bits(M, g)
            q["bits"] = bits.cuda()
            q["scale"] = 2.0
        C0 = torch.randn(M, 256, generator=g).cuda()
        if sp.get("accumulate"):
            q["accumulate"] = True
        q["C"] = C0.clone()
        probs.append(q)
        masks.append(mask)
        c0s.append(C0)
    out = [c.clone() for c in ops.gemm_rows256(probs)]
    torch.cuda.synchronize()
    for q, mask, C0, got, sp in zip(probs, masks, c0s, out, specs):
        close(got, ref_of(q, mask, C0), msg=str(sp))
    if repeat:
        for q, C0 in zip(probs, c0s):
            q["C"].copy_(C0)
        again = ops.gemm_rows256(probs)
        torch.cuda.synchronize()
        for a, b in zip(out, again):
            assert torch.equal(a, b)
    return out


def test_plain_products_whole_and_ragged_tiles(ops):
    # one tile, a ragged last tile, fewer tiles than CUs, more tiles than CUs (several tiles per workgroup)
    for M in (64, 50, 1000, 64 * 300 + 17):
        run_case(ops, [{"M": M}], seed=M)


def test_bias_and_tanh_epilogue(ops):
    run_case(ops, [{"M": 3200, "bias": True, "act": 2}], seed=1)
    run_case(ops, [{"M": 777, "bias": True}], seed=2)


def test_key_projection_with_fused_input_dropout(ops):
    """keys = tanh(drop(x) W^T + b): keep-bits on A's virtual rows, the two streams reading the same frames (row modulo)."""
    run_case(ops, [{"M": 2 * 1500, "mod": 1500, "mask": True, "bias": True, "act": 2}], seed=3, repeat=True)
    run_case(ops, [{"M": 2 * 64 * 40 + 2 * 13, "mod": 64 * 40 + 13, "mask": True, "bias": True, "act": 2}], seed=4)


def test_input_gradient_accumulates_into_c(ops):
    """dxd += dz W."""
    run_case(ops, [{"M": 6400, "accumulate": True}], seed=5, repeat=True)
    run_case(ops, [{"M": 1234, "accumulate": True}], seed=6)


def test_several_problems_in_one_launch(ops):
    """A launch's tile list is cut into one contiguous range per workgroup: ranges that cross problem boundaries reload B."""
    run_case(ops, [{"M": 64 * 90 + 5, "accumulate": True}, {"M": 64 * 37, "accumulate": True}, {"M": 100, "accumulate": True},
                   {"M": 64 * 200 + 63, "accumulate": True}], seed=7, repeat=True)
    run_case(ops, [{"M": 3000, "mod": 1500, "mask": True, "bias": True, "act": 2}, {"M": 64, "mask": True, "act": 2},
                   {"M": 64 * 129, "mask": True, "bias": True, "act": 2}], seed=8)


def test_c2_shapes(ops):
    """The audio sites of a C2 step: 48000 virtual rows of 24000 frames."""
    run_case(ops, [{"M": 48000, "mod": 24000, "mask": True, "bias": True, "act": 2}], seed=9)
    run_case(ops, [{"M": 48000, "accumulate": True}, {"M": 28800, "accumulate": True}], seed=10)


def run_pool(ops, specs, seed):
    """C = A B + sum_i pool_w[r, i] * pool_g[r // T, i, :] -- dxd = dz W + the attention pooling's own input gradient in one pass."""
    g = torch.Generator().manual_seed(seed)
    probs, refs = [], []
    for (V, T, nq) in specs:
        M = V * T
        A, B = (torch.randn(M, 256, generator=g) * 0.5).cuda(), (torch.randn(256, 256, generator=g) / 16).cuda()
        pw = torch.softmax(torch.randn(V, T, nq, generator=g), dim=1).reshape(M, nq).contiguous().cuda()
        pg = (torch.randn(V, nq, 256, generator=g) * (torch.rand(V, nq, 256, generator=g) > 0.3)).cuda()
        probs.append({"A": A, "B": B, "pool_w": pw, "pool_g": pg, "pool_T": T, "C": torch.full((M, 256), 7.0).cuda()})
        refs.append(A.double().cpu() @ B.double().cpu()
                    + torch.einsum("vti,vic->vtc", pw.double().cpu().view(V, T, nq), pg.double().cpu()).reshape(M, 256))
    out = [c.clone() for c in ops.gemm_rows256(probs)]
    for got, ref, sp in zip(out, refs, specs):
        close(got, ref, msg=str(sp))
    again = ops.gemm_rows256(probs)
    for a, b in zip(out, again):
        assert torch.equal(a, b)


def test_pooling_term_as_an_extra_k_tile(ops):
    """Sites of 7 queries and of 1; samples of 375 / 225 frames (tiles straddle sample boundaries), of 32 (two samples per tile), of 63 and 64;
    a ragged last tile; several problems per launch."""
    run_pool(ops, [(4, 375, 7)], seed=21)
    run_pool(ops, [(6, 225, 1)], seed=22)
    run_pool(ops, [(10, 32, 7), (3, 64, 7), (5, 63, 1)], seed=23)
    run_pool(ops, [(128, 375, 7), (128, 225, 7)], seed=24)      # the C2 audio and video sites


def run_fold(ops, specs, seed, accumulate=False, masked=True):
    """C[r] (+)= sum_s keep_s[r] . (A[s R + r] B + pooling term) * scale: the mask-sum of the input dropouts folded into the launch."""
    g = torch.Generator().manual_seed(seed)
    probs, refs = [], []
    for (fold, Bn, T, nq) in specs:
        R, V = Bn * T, fold * Bn
        M = fold * R
        A, B = (torch.randn(M, 256, generator=g) * 0.5).cuda(), (torch.randn(256, 256, generator=g) / 16).cuda()
        pw = torch.softmax(torch.randn(V, T, nq, generator=g), dim=1).reshape(M, nq).contiguous().cuda()
        pg = (torch.randn(V, nq, 256, generator=g) * (torch.rand(V, nq, 256, generator=g) > 0.3)).cuda()
        bits, mask = keep_bits(M, g)
        C0 = torch.randn(R, 256, generator=g).cuda()
        q = {"A": A, "B": B, "pool_w": pw, "pool_g": pg, "pool_T": T, "fold": fold, "C": C0.clone(), "accumulate": accumulate}
        if masked:
            q["c_bits"], q["c_scale"] = bits.cuda(), 2.0
        probs.append(q)
        dxd = (A.double().cpu() @ B.double().cpu()
               + torch.einsum("vti,vic->vtc", pw.double().cpu().view(V, T, nq), pg.double().cpu()).reshape(M, 256))
        if masked:
            dxd = dxd * mask * 2.0
        refs.append(dxd.view(fold, R, 256).sum(0) + (C0.double().cpu() if accumulate else 0.0))
    out = [c.clone() for c in ops.gemm_rows256(probs)]
    for got, ref, sp in zip(out, refs, specs):
        assert got.shape == ref.shape
        close(got, ref, msg=str(sp))
    return out


def test_mask_sum_folded_into_the_launch(ops):
    """Two streams over shared frames (fold 2), separate frames (fold 1); with and without keep-bits; onto an existing C; ragged tiles;
    bit-identical repeats."""
    a = run_fold(ops, [(2, 5, 375, 7)], seed=31)
    b = run_fold(ops, [(2, 5, 375, 7)], seed=31)
    assert torch.equal(a[0], b[0])
    run_fold(ops, [(2, 6, 225, 1)], seed=32, accumulate=True)
    run_fold(ops, [(1, 9, 32, 7), (1, 9, 32, 7)], seed=33)
    run_fold(ops, [(1, 7, 63, 1), (2, 3, 100, 7)], seed=34, accumulate=True)
    run_fold(ops, [(2, 4, 375, 7)], seed=35, masked=False)
    run_fold(ops, [(2, 64, 375, 7)], seed=36)                       # the C2 audio Cross_Attention site
    run_fold(ops, [(2, 64, 225, 1)], seed=37, accumulate=True)      # the C2 video FRA2UTT site onto it


def run_fold_bf16(ops, specs, seed, accumulate=False, masked=True):
    """The same on bf16 storage: A (dz), B ([256 n][256 k]: C = A B^T) and C (dx) bf16, attention weights / masked dout fp32."""
    g = torch.Generator().manual_seed(seed)
    bf = torch.bfloat16
    probs, refs = [], []
    fold = specs[0][0]
    for (f, Bn, T, nq) in specs:
        assert f == fold
        R, V = Bn * T, fold * Bn
        M = fold * R
        A, B = (torch.randn(M, 256, generator=g) * 0.5).to(bf).cuda(), (torch.randn(256, 256, generator=g) / 16).to(bf).cuda()
        pw = torch.softmax(torch.randn(V, T, nq, generator=g), dim=1).reshape(M, nq).contiguous().cuda()
        pg = (torch.randn(V, nq, 256, generator=g) * (torch.rand(V, nq, 256, generator=g) > 0.3)).cuda()
        bits, mask = keep_bits(M, g)
        C0 = torch.randn(R, 256, generator=g).to(bf).cuda()
        q = {"A": A, "B": B, "pool_w": pw, "pool_g": pg, "pool_T": T, "fold": fold, "C": C0.clone(), "accumulate": accumulate}
        if masked:
            q["c_bits"], q["c_scale"] = bits.cuda(), 2.0
        probs.append(q)
        dxd = (A.double().cpu() @ B.double().cpu().t()
               + torch.einsum("vti,vic->vtc", pw.double().cpu().view(V, T, nq), pg.double().cpu()).reshape(M, 256))
        if masked:
            dxd = dxd * mask * 2.0
        refs.append(dxd.view(fold, R, 256).sum(0) + (C0.double().cpu() if accumulate else 0.0))
    out = [c.clone() for c in ops.gemm_rows256(probs)]
    for got, ref, sp in zip(out, refs, specs):
        assert got.dtype == bf and got.shape == ref.shape
        close(got, ref, tol=6e-3, msg=str(sp))      # (one bf16 rounding of the output)
    return out


def test_bf16_mask_sum_folded_into_the_launch(ops):
    a = run_fold_bf16(ops, [(2, 5, 375, 7)], seed=41)
    b = run_fold_bf16(ops, [(2, 5, 375, 7)], seed=41)
    assert torch.equal(a[0], b[0])
    run_fold_bf16(ops, [(2, 6, 225, 1)], seed=42, accumulate=True)
    run_fold_bf16(ops, [(1, 9, 32, 7), (1, 9, 32, 7)], seed=43)
    run_fold_bf16(ops, [(1, 7, 63, 1), (1, 3, 100, 7)], seed=44, accumulate=True)
    run_fold_bf16(ops, [(2, 4, 375, 7), (2, 3, 70, 1)], seed=45, masked=False)
    run_fold_bf16(ops, [(2, 64, 375, 7), (2, 64, 225, 7)], seed=46)      # the C2 audio and video Cross_Attention sites, one launch
    run_fold_bf16(ops, [(2, 64, 225, 1)], seed=47, accumulate=True)
    run_fold_bf16(ops, [(1, 128, 32, 1)], seed=48, accumulate=True)     # the C2 text slot


def test_pooling_term_refuses_what_it_cannot_tile(ops):
    from sdumc_amd._lib import SdumcError
    A, B = torch.zeros(100, 256).cuda(), torch.zeros(256, 256).cuda()
    pw, pg = torch.zeros(100, 7).cuda(), torch.zeros(2, 7, 256).cuda()
    with pytest.raises(SdumcError):      # 50-frame samples: a 64-row tile can span three
        ops.gemm_rows256([{"A": A, "B": B, "pool_w": pw, "pool_g": pg, "pool_T": 50}])
    with pytest.raises(SdumcError):      # with accumulate
        ops.gemm_rows256([{"A": A, "B": B, "pool_w": pw, "pool_g": pg, "pool_T": 100, "accumulate": True, "C": torch.zeros(100, 256).cuda()}])


def test_mixed_variants_are_refused(ops):
    g = torch.Generator().manual_seed(0)
    A = torch.randn(128, 256, generator=g).cuda()
    B = torch.randn(256, 256, generator=g).cuda()
    bits, _ = keep_bits(128, g)
    with pytest.raises(RuntimeError):
        ops.gemm_rows256([{"A": A, "B": B, "bits": bits.cuda()}, {"A": A, "B": B}])
    with pytest.raises(RuntimeError):
        ops.gemm_rows256([{"A": A, "B": B, "bits": bits.cuda(), "accumulate": True}])


# ---- bf16 storage (sdumc_gemm_rows256_bf16) ----
def run_case_bf16(ops, specs, seed, repeat=False):
    g = torch.Generator().manual_seed(seed)
    probs, c0s = [], []
    for sp in specs:
        M, mod = sp["M"], sp.get("mod", 0)
        q = {"A": (torch.randn(mod or M, 256, generator=g) * 0.5).to(torch.bfloat16).cuda(),
             "B": (torch.randn(256, 256, generator=g) / 16).to(torch.bfloat16).cuda(),      # [n][k]
             "M": M, "a_row_mod": mod, "act": sp.get("act", 0)}
        if sp.get("bias"):
            q["bias"] = torch.randn(256, generator=g).cuda()
        C0 = torch.randn(M, 256, generator=g).to(torch.bfloat16).cuda()
        if sp.get("accumulate"):
            q["accumulate"] = True
        q["C"] = C0.clone()
        probs.append(q)
        c0s.append(C0)
    out = [c.clone() for c in ops.gemm_rows256(probs)]
    torch.cuda.synchronize()
    for q, C0, got, sp in zip(probs, c0s, out, specs):
        qq = dict(q)
        qq["B"] = q["B"].float().t().contiguous()      # the fp32 reference takes B as [k][n]
        qq["A"] = q["A"].float()
        want = ref_of(qq, None, C0.float())
        close(got.float(), want, tol=6e-3, msg=str(sp))      # one bf16 rounding of the result (2^-9 relative)
    if repeat:
        for q, C0 in zip(probs, c0s):
            q["C"].copy_(C0)
        again = ops.gemm_rows256(probs)
        torch.cuda.synchronize()
        for a, b in zip(out, again):
            assert torch.equal(a, b)


def test_bf16_plain_ragged_and_paired_tiles(ops):
    # one tile (its pair partner is nothing), odd and even tile counts per workgroup, a ragged last tile
    for M in (64, 50, 128, 1000, 64 * 300 + 17, 64 * 513):
        run_case_bf16(ops, [{"M": M}], seed=M)


def test_bf16_key_projection_and_input_gradient(ops):
    run_case_bf16(ops, [{"M": 48000, "bias": True, "act": 2}], seed=21, repeat=True)
    run_case_bf16(ops, [{"M": 3000, "mod": 1500, "bias": True, "act": 2}], seed=22)
    run_case_bf16(ops, [{"M": 48000, "accumulate": True}, {"M": 28800, "accumulate": True}, {"M": 100, "accumulate": True}],
                  seed=23, repeat=True)
    run_case_bf16(ops, [{"M": 64 * 37 + 5, "accumulate": True}, {"M": 64 * 90, "accumulate": True}], seed=24)


def test_row_strides_wider_than_256(ops):
    """A, B and C as column slices of wider tensors (lda / ldb / ldc > 256)."""
    g = torch.Generator().manual_seed(31)
    M = 64 * 5 + 9
    Aw = (torch.randn(M, 320, generator=g) * 0.5).cuda()
    Bw = (torch.randn(256, 384, generator=g) / 16).cuda()
    Cw = torch.randn(M, 512, generator=g).cuda()
    C0 = Cw.clone()
    A, B, C = Aw[:, 32:288], Bw[:, 64:320], Cw[:, 128:384]
    assert A.stride(0) == 320 and B.stride(0) == 384 and C.stride(0) == 512
    ops.gemm_rows256([{"A": A, "B": B, "C": C, "accumulate": True}])
    torch.cuda.synchronize()
    want = C0[:, 128:384].cpu().double() + A.cpu().double() @ B.cpu().double()
    close(C, want)
    assert torch.equal(Cw[:, :128], C0[:, :128]) and torch.equal(Cw[:, 384:], C0[:, 384:])     # nothing outside the slice is touched


def test_results_do_not_depend_on_memory_latency(ops):
    """The kernels never drain their vector-memory queue: every wait is a counted s_waitcnt vmcnt(n).  A copy loop on a second
    stream keeps HBM busy (loads land late and out of their usual rhythm); the results must stay bit-identical to a quiet run."""
    g = torch.Generator().manual_seed(41)
    M = 64 * 700 + 11
    A = (torch.randn(M, 256, generator=g) * 0.5).cuda()
    B = (torch.randn(256, 256, generator=g) / 16).cuda()
    C0 = torch.randn(M, 256, generator=g).cuda()
    bits, _ = keep_bits(M, g)
    bits = bits.cuda()
    bias = torch.randn(256, generator=g).cuda()
    Ah, Bh, C0h = A.to(torch.bfloat16), B.to(torch.bfloat16), C0.to(torch.bfloat16)

    def run_all():
        outs = []
        c = C0.clone()
        ops.gemm_rows256([{"A": A, "B": B, "C": c, "accumulate": True}])
        outs.append(c)
        outs.append(ops.gemm_rows256([{"A": A, "B": B, "bits": bits, "scale": 2.0, "bias": bias, "act": 2}])[0])
        c = C0h.clone()
        ops.gemm_rows256([{"A": Ah, "B": Bh, "C": c, "accumulate": True}])
        outs.append(c)
        outs.append(ops.gemm_rows256([{"A": Ah, "B": Bh, "bias": bias, "act": 2}])[0])
        torch.cuda.synchronize()
        return outs

    quiet = run_all()
    src = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
    dst = torch.empty_like(src)
    side = torch.cuda.Stream()
    for rep in range(6):
        with torch.cuda.stream(side):
            for _ in range(12):
                dst.copy_(src, non_blocking=True)
        busy = run_all()
        side.synchronize()
        for a, b in zip(quiet, busy):
            assert torch.equal(a.view(torch.int16) if a.dtype == torch.bfloat16 else a, b.view(torch.int16) if b.dtype == torch.bfloat16 else b)
