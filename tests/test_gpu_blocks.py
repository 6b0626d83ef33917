"""GPU parity of the stand-alone drop-in blocks (sdumc_amd.blocks.FRA2UTT_new / Cross_Attention = the UMCA block,
SURVEY §8a rows A2 / A6) against the goldens recorded from the REAL reference classes (tests/golden/blocks.npz) and,
for the gradients, against the CPU oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.asarray(a))


def close(a, b, tol=2e-5, what=""):
    a = a.detach().cpu().double().numpy()
    b = b.detach().cpu().double().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, dtype=np.float64)
    scale = max(1.0, float(np.abs(b).max()))
    err = float(np.abs(a - b.reshape(a.shape)).max())
    assert err <= tol * scale, f"{what}: max abs err {err:.3e}"


@pytest.fixture(scope="module")
def blocks():
    from sdumc_amd import blocks
    return blocks


def _load(blocks, g):
    fra = blocks.FRA2UTT_new(input_dim=256)
    ca = blocks.Cross_Attention(input_dim=256)
    fra.load_state_dict({"attention_context_vector": T(g["fra_ctx"]), "input_proj.weight": T(g["fra_w"]),
                         "input_proj.bias": T(g["fra_b"])})
    ca.load_state_dict({"query_proj.weight": T(g["ca_wq"]), "query_proj.bias": T(g["ca_bq"]),
                        "input_proj.weight": T(g["ca_wi"]), "input_proj.bias": T(g["ca_bi"])})
    return fra.cuda(), ca.cuda()


def test_state_dict_and_constructor_contract(blocks):
    ca = blocks.Cross_Attention(input_dim=256)
    assert sorted(ca.state_dict()) == ["input_proj.bias", "input_proj.weight", "query_proj.bias", "query_proj.weight"]
    fra = blocks.FRA2UTT_new(input_dim=256)
    assert sorted(fra.state_dict()) == ["attention_context_vector", "input_proj.bias", "input_proj.weight"]
    assert tuple(fra.attention_context_vector.shape) == (1, 256) and ca.softmax_scale == 0.3
    assert tuple(blocks.Cross_Attention().query_proj.weight.shape) == (1024, 1024)   # the reference default input_dim
    with pytest.raises(NotImplementedError):
        blocks.Cross_Attention(input_dim=300)     # only multiples of 256 up to 1024 are built
    from sdumc_amd._lib import SdumcError
    with pytest.raises(SdumcError):
        ca(torch.randn(2, 7, 256), torch.randn(2, 5, 256))     # CPU tensors: no fallback


def test_blocks_vs_reference_goldens_eval_and_train(blocks, golden):
    from oracle import sdumc_oracle as O
    g = golden("blocks")
    fra, ca = _load(blocks, g)
    x, q = T(g["x"]).cuda(), T(g["q"]).cuda()
    fra.eval(); ca.eval()
    with torch.no_grad():
        o, a = fra(x)
        close(o, g["fra_eval_out"], what="fra eval out"); close(a, g["fra_eval_att"], what="fra eval att")
        o, a = ca(q, x)
        close(o, g["ca_eval_out"], what="ca eval out"); close(a, g["ca_eval_att"], what="ca eval att")
    fra.train(); ca.train()
    seed, call = int(g["seed"]), int(g["call"])
    with torch.no_grad():
        blocks.manual_seed(seed, call, site=O.SITE_FRA_IN[1])
        o, a = fra(x)
        close(o, g["fra_train_out"], what="fra train out"); close(a, g["fra_train_att"], what="fra train att")
        assert torch.equal(o.cpu() == 0, T(g["fra_train_out"]) == 0)                # output-dropout mask bit-exact
        blocks.manual_seed(seed, call, site=O.SITE_CA_IN[2])
        o, a = ca(q, x)
        close(o, g["ca_train_out"], what="ca train out"); close(a, g["ca_train_att"], what="ca train att")
        assert torch.equal(o.cpu() == 0, T(g["ca_train_out"]) == 0)


@pytest.mark.parametrize("train", [False, True])
def test_blocks_backward_vs_oracle(blocks, golden, train):
    from oracle import sdumc_oracle as O
    g = golden("blocks")
    fra, ca = _load(blocks, g)
    fra.train(train); ca.train(train)
    gen = torch.Generator().manual_seed(5)
    B, Tn = 4, 37
    x = torch.randn(B, Tn, 256, generator=gen)
    q = torch.randn(B, 7, 256, generator=gen) / 4
    R1, R2 = torch.randn(B, 256, generator=gen), torch.randn(B, 7, 256, generator=gen)
    seed, call = 123, 9
    # oracle (float64)
    P = {"fra2utt_1.attention_context_vector": T(g["fra_ctx"]), "fra2utt_1.input_proj.weight": T(g["fra_w"]),
         "fra2utt_1.input_proj.bias": T(g["fra_b"]),
         "cross_att_fra2utt_2.query_proj.weight": T(g["ca_wq"]), "cross_att_fra2utt_2.query_proj.bias": T(g["ca_bq"]),
         "cross_att_fra2utt_2.input_proj.weight": T(g["ca_wi"]), "cross_att_fra2utt_2.input_proj.bias": T(g["ca_bi"])}
    P = {k: v.double().requires_grad_() for k, v in P.items()}
    xo, qo = x.double().requires_grad_(), q.double().requires_grad_()
    d = O.DropCtx("philox", seed, call) if train else O.DropCtx("eval")
    o1, _ = O.fra2utt(P, 1, xo, d)
    o2, _ = O.cross_attention(P, 2, qo, xo, d)
    ((o1 * R1.double()).sum() + (o2 * R2.double()).sum()).backward()
    # HIP modules
    xg, qg = x.cuda().requires_grad_(), q.cuda().requires_grad_()
    blocks.manual_seed(seed, call, site=O.SITE_FRA_IN[1])
    g1, _ = fra(xg)
    blocks.manual_seed(seed, call, site=O.SITE_CA_IN[2])
    g2, _ = ca(qg, xg)
    close(g1, o1, what="fra out"); close(g2, o2, what="ca out")
    ((g1 * R1.cuda()).sum() + (g2 * R2.cuda()).sum()).backward()
    close(xg.grad, xo.grad, 1e-4, "dx"); close(qg.grad, qo.grad, 1e-4, "dq")
    close(fra.attention_context_vector.grad, P["fra2utt_1.attention_context_vector"].grad, 1e-4, "d ctx")
    close(fra.input_proj.weight.grad, P["fra2utt_1.input_proj.weight"].grad, 1e-4, "fra dW")
    close(fra.input_proj.bias.grad, P["fra2utt_1.input_proj.bias"].grad, 1e-4, "fra db")
    close(ca.query_proj.weight.grad, P["cross_att_fra2utt_2.query_proj.weight"].grad, 1e-4, "ca dWq")
    close(ca.query_proj.bias.grad, P["cross_att_fra2utt_2.query_proj.bias"].grad, 1e-4, "ca dbq")
    close(ca.input_proj.weight.grad, P["cross_att_fra2utt_2.input_proj.weight"].grad, 1e-4, "ca dWi")
    close(ca.input_proj.bias.grad, P["cross_att_fra2utt_2.input_proj.bias"].grad, 1e-4, "ca dbi")


@pytest.mark.parametrize("Dm,train", [(512, True), (768, False), (1024, True)])
def test_blocks_other_widths_vs_oracle(blocks, Dm, train):
    """input_dim 512 / 768 / 1024 (the constructors' default; BASELINE configs[4]'s block-level variant with D = 1024):
    forward, dropout masks and every gradient against the oracle in float64."""
    from oracle import sdumc_oracle as O
    torch.manual_seed(Dm)
    fra = blocks.FRA2UTT_new(input_dim=Dm).cuda().train(train)
    ca = blocks.Cross_Attention(input_dim=Dm).cuda().train(train)
    gen = torch.Generator().manual_seed(Dm + 1)
    B, Tn, nq = 3, 70, 7
    x = torch.randn(B, Tn, Dm, generator=gen)
    q = torch.randn(B, nq, Dm, generator=gen) / 8
    R1, R2 = torch.randn(B, Dm, generator=gen), torch.randn(B, nq, Dm, generator=gen)
    seed, call = 321, 4
    P = {"fra2utt_0.attention_context_vector": fra.attention_context_vector, "fra2utt_0.input_proj.weight": fra.input_proj.weight,
         "fra2utt_0.input_proj.bias": fra.input_proj.bias,
         "cross_att_fra2utt_0.query_proj.weight": ca.query_proj.weight, "cross_att_fra2utt_0.query_proj.bias": ca.query_proj.bias,
         "cross_att_fra2utt_0.input_proj.weight": ca.input_proj.weight, "cross_att_fra2utt_0.input_proj.bias": ca.input_proj.bias}
    P = {k: v.detach().cpu().double().requires_grad_() for k, v in P.items()}
    xo, qo = x.double().requires_grad_(), q.double().requires_grad_()
    d = O.DropCtx("philox", seed, call) if train else O.DropCtx("eval")
    o1, a1 = O.fra2utt(P, 0, xo, d)
    o2, a2 = O.cross_attention(P, 0, qo, xo, d)
    ((o1 * R1.double()).sum() + (o2 * R2.double()).sum()).backward()
    xg, qg = x.cuda().requires_grad_(), q.cuda().requires_grad_()
    blocks.manual_seed(seed, call, site=O.SITE_FRA_IN[0])
    g1, b1 = fra(xg)
    blocks.manual_seed(seed, call, site=O.SITE_CA_IN[0])
    g2, b2 = ca(qg, xg)
    close(g1, o1, what="fra out"); close(b1, a1, what="fra att"); close(g2, o2, what="ca out"); close(b2, a2, what="ca att")
    if train:
        assert torch.equal(g2.detach().cpu() == 0, o2.detach() == 0)     # output-dropout mask bit-exact
    ((g1 * R1.cuda()).sum() + (g2 * R2.cuda()).sum()).backward()
    close(xg.grad, xo.grad, 1e-4, "dx"); close(qg.grad, qo.grad, 1e-4, "dq")
    for mod, pre in ((fra, "fra2utt_0."), (ca, "cross_att_fra2utt_0.")):
        for k, v in mod.named_parameters():
            close(v.grad, P[pre + k].grad, 1e-4, pre + k)


def test_default_width_blocks_vs_reference_goldens(blocks, golden):
    """input_dim = 1024, the reference constructors' default: outputs recorded from the REAL reference classes
    (tests/golden/blocks1024.npz).  The weights are re-drawn: same seed + same constructor order = same init stream."""
    from oracle import sdumc_oracle as O
    from tests.golden.make_goldens import digest
    g = golden("blocks1024")
    torch.manual_seed(int(g["seed_fra"]))
    fra = blocks.FRA2UTT_new()
    torch.manual_seed(int(g["seed_ca"]))
    ca = blocks.Cross_Attention()
    np.testing.assert_allclose(digest(fra.input_proj.weight, "fra.w"), g["fra_w_digest"], rtol=1e-12)
    np.testing.assert_allclose(digest(fra.attention_context_vector, "fra.ctx"), g["fra_ctx_digest"], rtol=1e-12)
    np.testing.assert_allclose(digest(ca.query_proj.weight, "ca.wq"), g["ca_wq_digest"], rtol=1e-12)
    np.testing.assert_allclose(digest(ca.input_proj.weight, "ca.wi"), g["ca_wi_digest"], rtol=1e-12)
    fra, ca = fra.cuda(), ca.cuda()
    x, q = T(g["x"]).cuda(), T(g["q"]).cuda()
    with torch.no_grad():
        fra.eval(); ca.eval()
        o, a = fra(x)
        close(o, g["fra_eval_out"], what="fra eval out"); close(a, g["fra_eval_att"], what="fra eval att")
        o, a = ca(q, x)
        close(o, g["ca_eval_out"], what="ca eval out"); close(a, g["ca_eval_att"], what="ca eval att")
        fra.train(); ca.train()
        seed, call = int(g["seed"]), int(g["call"])
        blocks.manual_seed(seed, call, site=O.SITE_FRA_IN[0])
        o, a = fra(x)
        close(o, g["fra_train_out"], what="fra train out"); close(a, g["fra_train_att"], what="fra train att")
        blocks.manual_seed(seed, call, site=O.SITE_CA_IN[0])
        o, a = ca(q, x)
        close(o, g["ca_train_out"], what="ca train out"); close(a, g["ca_train_att"], what="ca train att")
        assert torch.equal(o.cpu() == 0, T(g["ca_train_out"]) == 0)
