"""Long-run reproducibility guards of the HEADLINE arithmetic (fp32 tensors, products on the bf16 matrix pipe) and the
split switch as a per-context option.

Round 4 found run-to-run differences (wrong low halves of packed fp32 VALU results beside bf16 MFMA workgroups: DESIGN.md
section 7) first in bf16 storage, then -- once the fp32 GEMMs had bf16-MFMA neighbours -- in fp32 storage too.  The device build
carries no packed fp32 operations since; the trigger was fenced by elimination, not isolated, so these tests are what keeps the
default mode honest: 200 identical train steps / forwards must be bit-identical to the first, the last 60 beside a
bandwidth-heavy neighbour on another stream."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DIMS, T_C2 = (1024, 4096, 1024, 4096), (375, 32, 225, 32)


@pytest.fixture(scope="module")
def E():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from sdumc_amd import engine
    return engine


def flat_from(E, P, dims):
    lay = E.ParamLayout.get(*dims[:3])
    flat = torch.zeros(lay.total)
    for k, v in lay.views(flat).items():
        v.copy_(P[k])
    return flat.cuda(), lay


class Neighbour:
    """a copy loop on a side stream: HBM traffic and workgroups of another kernel beside the step"""

    def __init__(self):
        self.side = torch.cuda.Stream()
        self.src, self.dst = torch.empty(32 << 20, device="cuda"), torch.empty(32 << 20, device="cuda")

    def poke(self):
        with torch.cuda.stream(self.side):
            for _ in range(4):
                self.dst.copy_(self.src)


@pytest.mark.parametrize("planes", [True, False])
def test_fp32_split_train_step_is_bit_reproducible_over_200_runs(E, planes):
    """200 C2 train steps (B = 64, fp32 tensors, the default split arithmetic) from IDENTICAL state -- parameters, Adam moments,
    step counter and dropout call counter reset before each: the 8 loss values, the whole flat gradient bucket and the updated
    parameters are bit-identical to the first run's.  planes=True: the step over a resident batch (bench.py's headline, the epoch
    path: projections on bf16 planes, csrc/gemm_p3.hip); False: the set_batch-per-step path (in-kernel split, csrc/gemm_wide.hip)."""
    from oracle import sdumc_oracle as O
    from sdumc_amd import _lib
    assert _lib.lib.sdumc_get_split_() == 15 or True      # (an SDUMC_SPLIT=<mask> suite run guards that mask instead)
    B = 64
    P = O.init_params(DIMS, seed=0)
    flat0, lay = flat_from(E, P, DIMS)
    g = torch.Generator(device="cuda").manual_seed(41)
    feats = [torch.randn(B, T_C2[i], DIMS[i], device="cuda", generator=g) for i in range(4)]
    vals = torch.rand(B, device="cuda", generator=g) * 6 - 3
    flat = flat0.clone()
    ts = E.TrainStep(flat, B, T_C2, DIMS, seed=5, planes=planes)
    assert (ts._planes is not None) == planes
    ts.set_batch(*feats, vals)
    nb = Neighbour()
    ref, bad = None, []
    for rep in range(200):
        flat.copy_(flat0)
        ts.adam_m.zero_()
        ts.adam_v.zero_()
        ts.hyper[1] = 0.0
        ts.rng.set_call(0)
        torch.cuda.synchronize()
        if rep >= 140:
            nb.poke()
        losses = ts.run().clone()
        out = (losses, ts.grads.clone(), flat.clone())
        torch.cuda.synchronize()
        if ref is None:
            ref = out
            assert torch.isfinite(losses).all() and float(out[1].abs().max()) > 0
        elif not all(torch.equal(a, b) for a, b in zip(ref, out)):
            bad.append(rep)
    assert not bad, f"{len(bad)} of 199 fp32 (split) train steps differed from the first: runs {bad[:10]}"


def test_fp32_split_c4_eval_forward_is_bit_reproducible_over_200_runs(E):
    """200 eval-mode forwards of C4's global batch (B = 512, fp32 tensors, both streams) -- the shape whose pooling kernels
    showed one or two differing elements of cross_text in round 4 -- all five outputs bit-identical to the first run's."""
    from oracle import sdumc_oracle as O
    B = 512
    P = O.init_params(DIMS, seed=0)
    flat, lay = flat_from(E, P, DIMS)
    g = torch.Generator(device="cuda").manual_seed(43)
    audio, text, video, feat4 = [torch.randn(B, T_C2[i], DIMS[i], device="cuda", generator=g) for i in range(4)]
    nc = E.NetCall(flat, audio, [text, feat4], video, False, None, planes=True)
    assert nc._planes is not None
    nb = Neighbour()
    ref, bad = None, []
    for rep in range(200):
        if rep >= 140:
            nb.poke()
        out = [t.clone() for t in nc.forward()]
        torch.cuda.synchronize()
        if ref is None:
            ref = out
            assert all(torch.isfinite(t).all() for t in out)
        elif not all(torch.equal(a, b) for a, b in zip(ref, out)):
            bad.append(rep)
    assert not bad, f"{len(bad)} of 199 fp32 (split) C4 forwards differed from the first: runs {bad[:10]}"


def test_split_is_an_option_of_the_execution_context(E):
    """sdumc_ctx_set_option(SDUMC_OPT_SPLIT): a context with split = 0 computes the step on the fp32 MFMAs -- bit-identical to the
    same step under the process-wide sdumc_set_split_(0) -- while steps on the default context before and after keep the bf16-pipe
    products; the process default (sdumc_get_split_) never changes."""
    from oracle import sdumc_oracle as O
    from sdumc_amd import _lib
    lib = _lib.lib
    dims, B, Tn = (1024, 4096, 1024, 4096), 4, (130, 32, 70, 32)
    P = O.init_params(dims, seed=4)
    batch = [t.cuda() for t in O.synthetic_batch(B, Tn, dims, seed=6)]
    default_mask = lib.sdumc_get_split_()

    def run(ctx=None):
        flat, lay = flat_from(E, P, dims)
        ts = E.TrainStep(flat, B, Tn, dims, seed=9, ctx=ctx)
        ts.set_batch(*batch)
        losses = ts.run().clone()
        torch.cuda.synchronize()
        return losses, ts.grads.clone(), flat.clone()

    before = run()
    ctx = E.ExecContext()
    ctx.set_option("split", 0)
    with_ctx = run(ctx)
    assert lib.sdumc_get_split_() == default_mask
    after = run()
    try:
        lib.sdumc_set_split_(0)
        with_global = run()
    finally:
        lib.sdumc_set_split_(default_mask)
    for a, b in zip(with_ctx, with_global):
        assert torch.equal(a, b)
    for a, b in zip(before, after):
        assert torch.equal(a, b)
    if default_mask:
        assert not torch.equal(before[1], with_ctx[1])          # (the context's option really changed the arithmetic)
    np.testing.assert_allclose(with_ctx[0].cpu().numpy(), before[0].cpu().numpy(), rtol=1e-5, atol=1e-6)
    ctx.set_option("split", None)
    back = run(ctx)
    for a, b in zip(back, before):
        assert torch.equal(a, b)
    ctx.close()


@pytest.mark.parametrize("bf16", [False, "operands"])
def test_one_lane_step_equals_the_four_lane_step(E, bf16):
    """with every lane on the caller's stream (concurrency 0) the lanes share ONE scratch region; the folded dq
    slabs of the Cross_Attention pooling backward must survive the early key-projection backward that runs before the clustered
    stage sums them.  In the operand-rounding mode (bf16 = 1) that backward is a per-layer split-K GEMM with its slabs in the lane's
    scratch -- the case that overwrote the slabs.  One lane against four lanes: same kernels, same order of every sum -- gradients
    bit-identical."""
    from oracle import sdumc_oracle as O
    dims, B, Tn = (1024, 4096, 1024, 4096), 8, (375, 32, 225, 32)
    P = O.init_params(dims, seed=4)
    batch = [t.cuda() for t in O.synthetic_batch(B, Tn, dims, seed=8)]

    def run(ctx=None):
        flat, lay = flat_from(E, P, dims)
        ts = E.TrainStep(flat, B, Tn, dims, seed=9, ctx=ctx, bf16=bf16)
        ts.set_batch(*batch)
        losses = ts.run().clone()
        torch.cuda.synchronize()
        return losses, ts.grads.clone()

    four = run()
    ctx = E.ExecContext()
    ctx.set_option("concurrency", 0)
    one = run(ctx)
    ctx.close()
    assert torch.equal(four[0], one[0])
    assert torch.equal(four[1], one[1])


def test_next_step_keep_bits_are_the_same_bits(E):
    """sdumc_net_io.bits_next: a train step generates the NEXT step's keep-bits in its middle and the next step copies them when the
    {seed, call} tag matches.  Ten steps with and without the buffer: losses, gradients and parameters bit for bit; a reset of the
    call counter (tag mismatch -> generated as ever) and a repeat of the same call (tag match on stale-but-identical bits) included."""
    dims = (1024, 4096, 1024)
    lay = E.ParamLayout.get(*dims)
    g = torch.Generator().manual_seed(11)
    flat0 = (torch.randn(lay.total, generator=g) * 0.02).cuda()
    B, T = 8, (130, 32, 70, 32)
    batch = [torch.randn(B, T[0], 1024, generator=g), torch.randn(B, T[1], 4096, generator=g), torch.randn(B, T[2], 1024, generator=g),
             torch.randn(B, T[3], 4096, generator=g), torch.randn(B, generator=g)]
    runs = []
    for use in (False, True):
        flat = flat0.clone()
        ts = E.TrainStep(flat, B, T, dims, seed=5, bits_next=use)
        assert (ts._bits_next is not None) == use
        ts.set_batch(*[t.cuda() for t in batch])
        out = []
        for step in range(10):
            if step == 4:
                ts.rng.set_call(2)          # back to an earlier call: the tag names another one
            if step == 7:
                ts.rng.set_call(2 * 6)      # the same call again: the tag matches, the bits are that call's
            out.append((ts.run().clone(), ts.grads.clone()))
        if use:      # the tag left behind names the call after the last one: {seed, call + 2, magic}; the buffers behind it are filled
            half = ts._bits_next.numel() // 2
            tags = [ts._bits_next[o:o + 16].view(torch.int32).cpu() for o in (0, half)]
            calls = sorted(int(t[2]) for t in tags)
            assert calls == [2 * 6 + 2 * 2, 2 * 6 + 2 * 3], calls      # the set the last step read (its own call) and the one it filled (the next)
            assert all((int(t[3]) & 0xFFFFFFFF) == 0x5D0CB175 for t in tags)
            assert int(ts._bits_next.count_nonzero()) > ts._bits_next.numel() // 4
        runs.append((out, flat.clone()))
    for (la, ga), (lb, gb) in zip(runs[0][0], runs[1][0]):
        assert torch.equal(la, lb) and torch.equal(ga, gb)
    assert torch.equal(runs[0][1], runs[1][1])
