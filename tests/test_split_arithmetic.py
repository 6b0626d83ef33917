"""CPU restatement of the arithmetic behind "fp32 products on the bf16 matrix pipe" (DESIGN.md section 4,
sdumc_amd/csrc/gemm_group.hip): the three-way bf16 split of an fp32 value is EXACT, every part product is exact in fp32, and the
six products the kernels keep differ from the fp32 product by less than one fp32 rounding.  No GPU, no library: torch's
bfloat16 cast is round-to-nearest-even, the same conversion as v_cvt_pk_bf16_f32."""
import numpy as np
import torch


def split3(a):
    """a (fp32) -> a0, a1, a2 (bf16 values held in fp32), exactly as split2() in the kernels"""
    a0 = a.to(torch.bfloat16).to(torch.float32)
    r1 = a - a0                                   # exact in fp32 (asserted below)
    a1 = r1.to(torch.bfloat16).to(torch.float32)
    r2 = r1 - a1                                  # exact
    a2 = r2.to(torch.bfloat16).to(torch.float32)
    return a0, a1, a2, r1, r2


def sample(n, seed):
    g = torch.Generator().manual_seed(seed)
    mant = torch.randn(n, generator=g)
    expo = torch.randint(-30, 30, (n,), generator=g).float()
    edge = torch.tensor([1.0, -1.0, 1.0 + 2.0 ** -23, 1.0 - 2.0 ** -24, 255.0 / 256.0, 3.0e38, 1.2e-30, 0.0, -0.0,
                         float(np.float32(1.0) / np.float32(3.0)), 16777215.0, 0.1])
    return torch.cat([mant * torch.exp2(expo), edge])


def test_three_bf16_parts_sum_to_the_fp32_value_exactly():
    a = sample(200000, 1)
    a0, a1, a2, r1, r2 = split3(a)
    ad, a0d, a1d, a2d = a.double(), a0.double(), a1.double(), a2.double()
    assert torch.equal(r1.double(), ad - a0d), "a - bf16(a) is not exact in fp32"
    assert torch.equal(r2.double(), ad - a0d - a1d), "the second residual is not exact in fp32"
    assert torch.equal(a0d + a1d + a2d, ad), "three parts do not reproduce the value"
    # sizes of the parts (round to nearest): |a1| <= 2^-8 |a|, |a2| <= 2^-16 |a|
    nz = ad != 0
    assert float((a1d[nz].abs() / ad[nz].abs()).max()) <= 2.0 ** -8
    assert float((a2d[nz].abs() / ad[nz].abs()).max()) <= 2.0 ** -16


def test_part_products_are_exact_in_fp32_and_six_of_nine_are_within_one_rounding():
    a, b = sample(200000, 2), sample(200000, 3).flip(0)
    pa, pb = split3(a)[:3], split3(b)[:3]
    exact = a.double() * b.double()
    inrange = (exact.abs() < 1e30) & (exact.abs() > 1e-25)       # (no overflow, no part product in the subnormal range)
    for x in pa:
        for y in pb:
            prod = x * y                                        # fp32 multiply: 8-bit x 8-bit significands fit 24 bits
            assert torch.equal(prod.double()[inrange], (x.double() * y.double())[inrange]), "a bf16 x bf16 product is not exact in fp32"
    six = (pa[2].double() * pb[0].double() + pa[0].double() * pb[2].double() + pa[1].double() * pb[1].double()
           + pa[1].double() * pb[0].double() + pa[0].double() * pb[1].double() + pa[0].double() * pb[0].double())
    rel = ((six - exact).abs() / exact.abs())[inrange]
    assert float(rel.max()) < 2.0 ** -23, f"dropped terms {float(rel.max()):.3e} of the product"
    # ... which is the bound of ONE fp32 rounding of the exact product (half an ulp <= 2^-24 relative, an ulp 2^-23)


def test_a_dot_product_on_six_terms_is_as_accurate_as_the_fp32_fma_chain():
    """K = 4096 terms accumulated in fp32 (the kernels' accumulators), per-product error of the six-term form included, against the
    same sum as an fp32 FMA chain (what v_mfma_f32_32x32x2_f32 computes), both against fp64."""
    g = torch.Generator().manual_seed(5)
    K, N = 4096, 512
    a, b = torch.randn(N, K, generator=g), torch.randn(N, K, generator=g)
    ref = (a.double() * b.double()).sum(1)
    # fp32 FMA chain in k order
    chain = torch.zeros(N, dtype=torch.float64)
    acc = torch.zeros(N)
    for k in range(K):
        acc = torch.addcmul(acc, a[:, k], b[:, k])              # one rounding per step (fused in hardware; here two, a superset)
    chain = acc.double()
    # six exact part products per k, summed into an fp32 accumulator smallest first
    pa, pb = split3(a)[:3], split3(b)[:3]
    acc = torch.zeros(N)
    for k in range(K):
        for x, y in ((2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)):
            acc = acc + pa[x][:, k] * pb[y][:, k]
    six = acc.double()
    scale = float((a.double().abs() * b.double().abs()).sum(1).max())
    e_chain, e_six = float((chain - ref).abs().max()) / scale, float((six - ref).abs().max()) / scale
    # (this emulation rounds after EVERY one of the 6 K additions -- an upper bound on what an MFMA's block sum of 16 products per
    #  instruction commits; on the GPU the six-term kernels measure as close to fp64 as the fp32-MFMA kernels, tests/test_gpu_split.py)
    assert e_six < 1e-6 and e_six <= 4.0 * e_chain + 1e-8, (e_chain, e_six)


def six_term_product(a, b):
    """the six kept part products of a * b summed in fp32, smallest first (what one k step of the kernels adds)"""
    pa, pb = split3(a)[:3], split3(b)[:3]
    acc = torch.zeros_like(a)
    for x, y in ((2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)):
        acc = acc + pa[x] * pb[y]
    return acc


def test_contract_at_the_edges_of_the_split():
    """include/sdumc_hip.h, "CONTRACT AT THE EDGES": what the split arithmetic returns for NaN, +-Inf, values that round to Inf in
    bf16, -0 and tiny values -- restated on the CPU (the GPU cases are in tests/test_gpu_split.py)."""
    FLT_MAX = float(np.finfo(np.float32).max)
    one = torch.tensor([1.5])
    # NaN stays NaN
    assert torch.isnan(six_term_product(torch.tensor([float("nan")]), one)).all()
    # +-Inf -> NaN (the first part is Inf, the residual Inf - Inf is NaN); the fp32 product would be +-Inf
    for v in (float("inf"), -float("inf")):
        a0, a1, a2, r1, r2 = split3(torch.tensor([v]))
        assert torch.isinf(a0).all() and torch.isnan(r1).all()
        assert torch.isnan(six_term_product(torch.tensor([v]), one)).all()
        assert torch.isinf(torch.tensor([v]) * one).all()
    # the largest fp32 whose first part is still finite: 0x7F7F7FFF (just below 0x1.FEp127 + half a bf16 ulp); above it -> Inf -> NaN
    edge_ok = torch.tensor([0x7F7F7FFF], dtype=torch.int32).view(torch.float32)
    edge_bad = torch.tensor([0x7F7F8000], dtype=torch.int32).view(torch.float32)
    assert float(edge_bad) > 3.3895e38 and float(edge_bad) < FLT_MAX
    a0, a1, a2, r1, r2 = split3(edge_ok)
    assert torch.isfinite(a0).all() and torch.equal((a0.double() + a1.double() + a2.double()), edge_ok.double())
    assert torch.isinf(split3(edge_bad)[0]).all()
    assert torch.isnan(six_term_product(edge_bad, torch.tensor([1e-3]))).all()      # fp32: a finite 3.39e35
    assert torch.isnan(six_term_product(torch.tensor([FLT_MAX]), torch.tensor([1e-3]))).all()
    # -0 and +0: every part is a zero, the product sum is a zero
    for z in (0.0, -0.0):
        parts = split3(torch.tensor([z]))[:3]
        assert all(float(p) == 0.0 for p in parts)
        assert float(six_term_product(torch.tensor([z]), one)) == 0.0
    # tiny values.  bf16 has fp32's exponent range but its subnormals stop at 2^-133 (fp32: 2^-149), and the matrix pipe may flush
    # subnormal bf16 inputs.  (i) |x| >= 2^-102: every non-zero part is a multiple of ulp(x) >= 2^-125, i.e. a NORMAL bf16: exact,
    # flush or not.  (ii) below: a part under 2^-126 may be lost: absolute error per value < 2^-125.
    g = torch.Generator().manual_seed(11)
    x = torch.cat([torch.randn(20000, generator=g) * 2.0 ** -101, torch.tensor([2.0 ** -102, -(2.0 ** -102) * (1 + 2.0 ** -23)])])
    x = x[x.abs() >= 2.0 ** -102]
    a0, a1, a2, _, _ = split3(x)
    assert torch.equal(a0.double() + a1.double() + a2.double(), x.double())
    for part in (a0, a1, a2):
        nz = part != 0
        assert (part[nz].abs() >= 2.0 ** -126).all(), "a part of a value >= 2^-102 must be a normal bf16"
    tiny = torch.cat([torch.randn(20000, generator=g) * 2.0 ** -120, torch.tensor([1.1e-36, -3.3e-38, 1e-40, 2.0 ** -126, 2.0 ** -149])])
    parts = split3(tiny)[:3]
    flushed = [torch.where(p.abs() < 2.0 ** -126, torch.zeros_like(p), p) for p in parts]
    for ps in (parts, flushed):
        err = (ps[0].double() + ps[1].double() + ps[2].double() - tiny.double()).abs()
        assert float(err.max()) < 2.0 ** -125
