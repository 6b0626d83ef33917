"""GPU parity tests of the grouped weight-gradient GEMM (sdumc_gemm_group_tn, csrc/gemm_group.hip): the dW = dz^T x products of
model :282-284, :60, :82, :293-368 under loss.backward() (main :149), all problems of a backward phase in one persistent
stream-K launch + one ordered reduce.  Checked against fp64 matmuls at 2e-5 and for bit-identical repeats."""
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


def keep_bits(K, N, g):
    """uint8 [K, N/4], bit e of byte q = keep column 4q + e; and the 0/1 mask it encodes."""
    m = (torch.rand(K, N, generator=g) >= 0.5)
    b = (m.reshape(K, N // 4, 4).to(torch.int32) * torch.tensor([1, 2, 4, 8], dtype=torch.int32)).sum(-1).to(torch.uint8)
    return b, m.double()


def ref_of(q, masks):
    A, B = q["A"].cpu().double(), q["B"].cpu().double()
    K = q.get("K", A.shape[0])
    M, N = q.get("M", A.shape[1]), q.get("N", B.shape[1])

    def seg(A, B, K, mod, mask, scale):
        A, B = A[:K, :M], B[:, :N]
        rows = torch.arange(K) % mod if mod else torch.arange(K)
        Bv = B[rows]
        if mask is not None:
            Bv = Bv * mask[:K, :N] * scale
        return A.T @ Bv, A.sum(0)
    w, cs = seg(A, B, K, q.get("b_row_mod", 0), masks.get("m0"), q.get("scale", 1.0))
    if q.get("A1") is not None:
        w1, cs1 = seg(q["A1"].cpu().double(), q["B1"].cpu().double(), q.get("K1", q["A1"].shape[0]), q.get("b_row_mod1", 0),
                      masks.get("m1"), q.get("scale", 1.0))
        w, cs = w + w1, cs + cs1
    return w, cs


def make_problems(specs, seed):
    g = torch.Generator().manual_seed(seed)
    probs, masks = [], []
    for sp in specs:
        M, N, K = sp["M"], sp["N"], sp["K"]
        mod = sp.get("mod", 0)
        q = {"A": torch.randn(K, M, generator=g).cuda(), "B": torch.randn(mod or K, N, generator=g).cuda(), "b_row_mod": mod}
        mk = {}
        if sp.get("K1"):
            mod1 = sp.get("mod1", 0)
            q["A1"] = torch.randn(sp["K1"], M, generator=g).cuda()
            q["B1"] = torch.randn(mod1 or sp["K1"], N, generator=g).cuda()
            q["b_row_mod1"] = mod1
        if sp.get("mask"):
            b, m = keep_bits(K, N, g)
            q["bits"], mk["m0"], q["scale"] = b.cuda(), m, 2.0
            if sp.get("K1"):
                b1, m1 = keep_bits(sp["K1"], N, g)
                q["bits1"], mk["m1"] = b1.cuda(), m1
        if sp.get("cs", True):
            q["colsum"] = torch.zeros(M).cuda()
        probs.append(q)
        masks.append(mk)
    return probs, masks


def check_all(ops, specs, seed, tol=2e-5):
    probs, masks = make_problems(specs, seed)
    outs = ops.gemm_group_tn(probs)
    first = [o.clone() for o in outs]
    first_cs = [q["colsum"].clone() if q.get("colsum") is not None else None for q in probs]
    for i, (q, mk) in enumerate(zip(probs, masks)):
        w, cs = ref_of(q, mk)
        close(outs[i], w, tol, f"problem {i} {specs[i]}")
        if q.get("colsum") is not None:
            close(q["colsum"], cs, tol, f"colsum of problem {i}")
    # bit-identical repeats (the slab sums run in a fixed order)
    for _ in range(2):
        for q in probs:
            q["C"].fill_(float("nan"))
        outs = ops.gemm_group_tn(probs)
        for i, q in enumerate(probs):
            assert torch.equal(outs[i], first[i]), f"problem {i} not reproducible"
            if first_cs[i] is not None:
                assert torch.equal(q["colsum"], first_cs[i])
    return probs, masks


def test_single_problem_shapes(ops):
    """one problem per call: whole-tile pieces, split tiles, ragged K, narrow M / N"""
    for sp in [dict(M=256, N=256, K=128), dict(M=256, N=1024, K=3000), dict(M=128, N=256, K=896), dict(M=64, N=64, K=130),
               dict(M=256, N=768, K=128), dict(M=64, N=128, K=77), dict(M=8, N=12, K=40), dict(M=256, N=128, K=16),
               dict(M=512, N=256, K=1000), dict(M=260, N=132, K=333)]:
        check_all(ops, [sp], 11 + sp["K"])


def test_row_mod_mask_and_segments(ops):
    """the fusions of the frame-level problems: x shared by the two streams (row modulo), keep-bits of the input dropout on B,
    two K segments (the two streams' text tensors), every combination"""
    specs = [dict(M=256, N=256, K=4000, mod=2000, mask=True),
             dict(M=256, N=256, K=1500, K1=1300, mask=True),
             dict(M=256, N=512, K=1000, K1=1000),
             dict(M=256, N=256, K=999, K1=77, mod=333, mod1=40, mask=True),
             dict(M=256, N=1024, K=2048, K1=2048, cs=False)]
    for sp in specs:
        check_all(ops, [sp], 5 + sp["K"])
    check_all(ops, specs, 99)


def test_backward_phase_mix(ops):
    """a phase's problem list at reduced sizes: long-K frame problems next to the utterance-level layers (K = 128 / 896)"""
    specs = [dict(M=256, N=1024, K=6000), dict(M=256, N=1024, K=3600), dict(M=256, N=2048, K=512, K1=512),
             dict(M=256, N=256, K=12000, mod=6000, mask=True), dict(M=256, N=256, K=12000, mod=6000, mask=True)]
    specs += [dict(M=256, N=256, K=128) for _ in range(8)]
    specs += [dict(M=256, N=768, K=128), dict(M=128, N=256, K=896), dict(M=256, N=896, K=128), dict(M=64, N=128, K=128),
              dict(M=64, N=64, K=128)]
    check_all(ops, specs, 123)


def test_more_problems_than_one_launch_takes(ops):
    specs = [dict(M=256, N=256, K=128 + 16 * i) for i in range(30)]
    check_all(ops, specs, 7)


def test_accumulate(ops):
    g = torch.Generator().manual_seed(5)
    A, B = torch.randn(5000, 256, generator=g).cuda(), torch.randn(5000, 384, generator=g).cuda()
    C0, cs0 = torch.randn(256, 384, generator=g), torch.randn(256, generator=g)
    for K in (5000, 64):     # split tile / whole-tile piece
        q = {"A": A[:K], "B": B[:K], "C": C0.clone().cuda(), "colsum": cs0.clone().cuda(), "accumulate": True}
        ops.gemm_group_tn([q])
        close(q["C"], C0.double() + A[:K].cpu().double().T @ B[:K].cpu().double())
        close(q["colsum"], cs0.double() + A[:K].cpu().double().sum(0))


def test_full_size_frame_problems_property(ops):
    """C2's frame-level weight gradients at full size (B = 64): linearity in K -- the gradient over all rows equals the sum of
    the gradients over two halves computed by separate calls -- and agreement with the per-layer split-K GEMM"""
    g = torch.Generator().manual_seed(17)
    K, M, N = 24000, 256, 1024
    A, B = torch.randn(K, M, generator=g).cuda(), torch.randn(K, N, generator=g).cuda()
    whole = ops.gemm_group_tn([{"A": A, "B": B}])[0]
    h0 = ops.gemm_group_tn([{"A": A[:12000], "B": B[:12000]}])[0]
    h1 = ops.gemm_group_tn([{"A": A[12000:], "B": B[12000:]}])[0]
    close(whole, (h0.double() + h1.double()), 1e-5)
    old = ops.gemm(ops.TN, A, B, M, N, K, splitk=0)
    close(whole, old.double(), 1e-5)


def make_bf16(specs, seed):
    g = torch.Generator().manual_seed(seed)
    probs = []
    for sp in specs:
        M, N, K, mod = sp["M"], sp["N"], sp["K"], sp.get("mod", 0)
        q = {"A": torch.randn(K, M, generator=g).bfloat16().cuda(), "B": torch.randn(mod or K, N, generator=g).bfloat16().cuda(),
             "b_row_mod": mod}
        if sp.get("K1"):
            q["A1"] = torch.randn(sp["K1"], M, generator=g).bfloat16().cuda()
            q["B1"] = torch.randn(sp["K1"], N, generator=g).bfloat16().cuda()
        if sp.get("cs", True):
            q["colsum"] = torch.zeros(M).cuda()
        probs.append(q)
    return probs


def test_bf16_storage_problems(ops):
    """sdumc_gemm_group_tn_bf16: bf16 operands, fp32 accumulation and output, against fp64 products of the same bf16 values
    (2e-5: only the summation differs), bit-identical repeats; shapes of the frame-level problems at reduced K, ragged K,
    the row modulo, two K segments, narrow M / N, and a mix in one launch"""
    specs = [dict(M=256, N=1024, K=3000), dict(M=256, N=256, K=4096, mod=2048), dict(M=256, N=512, K=1000, K1=1030),
             dict(M=128, N=256, K=77), dict(M=64, N=64, K=640), dict(M=256, N=128, K=64), dict(M=8, N=16, K=200),
             dict(M=256, N=2048, K=512, K1=512, cs=False)]
    for sp in specs:
        probs = make_bf16([sp], 3 + sp["K"])
        outs = ops.gemm_group_tn(probs)
        first = outs[0].clone()
        w, cs = ref_of(probs[0], {})
        close(outs[0], w, 2e-5, str(sp))
        if probs[0].get("colsum") is not None:
            close(probs[0]["colsum"], cs, 2e-5, "colsum " + str(sp))
        probs[0]["C"].fill_(float("nan"))
        assert torch.equal(ops.gemm_group_tn(probs)[0], first)
    probs = make_bf16(specs, 77)
    outs = ops.gemm_group_tn(probs)
    for i, q in enumerate(probs):
        w, cs = ref_of(q, {})
        close(outs[i], w, 2e-5, f"mixed launch, problem {i}")
        if q.get("colsum") is not None:
            close(q["colsum"], cs, 2e-5)
