"""GPU parity of the generic MHA / LayerNorm / Transformer-encoder family:
HIP kernels (through the C ABI and the drop-in modules) vs the reference's golden vectors
(tests/golden/transformer.npz) and vs the CPU oracle on seeded inputs.  Tolerance: 1e-3 fp32 (north_star);
dropout masks bit-exact."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = 1e-3


@pytest.fixture(scope="module")
def te():
    from sdumc_amd import transformers_encoder
    return transformers_encoder


@pytest.fixture(scope="module")
def ops():
    from sdumc_amd import ops
    return ops


def T(a):
    return torch.from_numpy(np.asarray(a))


def G(a):
    return T(a).cuda()


def close(a, b, tol=TOL, what=""):
    a = a.detach().cpu().double().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, dtype=np.float64)
    b = b.detach().cpu().double().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, dtype=np.float64)
    scale = max(1.0, float(np.abs(b).max()))
    err = float(np.abs(a - b).max())
    assert err <= tol * scale, f"{what}: max abs err {err:.3e} (scale {scale:.3e})"


def load_params(mod, g, tag):
    sd = {k[len(tag) + 3:]: T(g[k]) for k in g.files if k.startswith(tag + "/P/")}
    mod.load_state_dict(sd, strict=True)
    return mod.cuda()


def check_grads(mod, g, tag, **inputs):
    for k, v in mod.named_parameters():
        close(v.grad, g[f"{tag}/G/{k}"], what=f"{tag} d{k}")
        v.grad = None
    for name, t in inputs.items():
        close(t.grad, g[f"{tag}/d{name}"], what=f"{tag} d{name}")


# ---------------------------------------------------------------------------------------------------
# operator level
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rows,width", [(7, 30), (15, 300), (33, 512), (64, 1024), (9, 1536), (5, 2052), (6, 4100), (3, 777)])
def test_layernorm_fwd_bwd(ops, rows, width):
    from oracle import transformer_oracle as TO
    g = torch.Generator().manual_seed(rows * 1000 + width)
    x = (1.5 * torch.randn(rows, width, generator=g) + 0.5).double().requires_grad_()
    w = (1 + 0.3 * torch.randn(width, generator=g)).double().requires_grad_()
    b = (0.3 * torch.randn(width, generator=g)).double().requires_grad_()
    R = torch.randn(rows, width, generator=g).double()
    y = TO.layer_norm(x, w, b)
    (y * R).sum().backward()
    xg, wg, bg, Rg = x.detach().float().cuda(), w.detach().float().cuda(), b.detach().float().cuda(), R.float().cuda()
    yg, mean, rstd = ops.layernorm_fwd(xg, wg, bg)
    close(yg, y, 1e-5, "y")
    dx, dw, db = ops.layernorm_bwd(Rg, xg, wg, mean, rstd)
    close(dx, x.grad, 2e-5, "dx"); close(dw, w.grad, 2e-5, "dw"); close(db, b.grad, 2e-5, "db")
    base = torch.randn(rows, width, generator=g).cuda()
    acc, _, _ = ops.layernorm_bwd(Rg, xg, wg, mean, rstd, dx_add=base, need_params=False)
    close(acc, base.double().cpu() + x.grad, 2e-5, "dx accumulate")


@pytest.mark.parametrize("B,H,tq,tk,p,masked", [(2, 3, 5, 9, 0.25, True), (2, 4, 16, 64, 0.0, False), (1, 2, 7, 512, 0.1, True),
                                                (2, 2, 3, 1000, 0.5, False), (1, 2, 4, 2500, 0.2, True), (1, 3, 5, 2051, 0.3, True),
                                                (3, 2, 6, 301, 0.0, True)])
def test_softmax_fwd_bwd_mask_dropout_headmean(ops, B, H, tq, tk, p, masked):
    from oracle import philox
    from sdumc_amd._lib import make_dropout
    g = torch.Generator().manual_seed(B * 7 + tk)
    S = (3 * torch.randn(B * H, tq, tk, generator=g))
    mask = torch.triu(torch.full((tq, tk), float("-inf")), 1 + abs(tk - tq)) if masked else None
    scale = 0.37
    seed, call, site = 1234567890123, 5, 11
    drop = make_dropout(True, site, p, tq, tk, B * H, 0, call, seed) if p > 0 else None
    Sd = S.double().requires_grad_()
    P_ref = torch.softmax(Sd * scale + (mask.double() if masked else 0), -1)
    m = torch.from_numpy(philox.dropout_mask(B * H, tq, tk, p, seed, call, site)).double() if p > 0 else 1.0
    Pd_ref = P_ref * m
    R = torch.randn(B * H, tq, tk, generator=g).double()
    (Pd_ref * R).sum().backward()
    Pg, Pd, W, desc = ops.softmax_fwd(S.cuda().clone(), B, H, scale, G(mask) if masked else None, drop)
    close(Pg, P_ref, 1e-5, "P")
    if p > 0:
        keep_gpu = (Pd != 0).cpu()
        keep_ref = (torch.as_tensor(m) != 0) & (P_ref != 0)
        assert torch.equal(keep_gpu, keep_ref), "dropout keep pattern differs from the Philox oracle"   # bit-exact
        close(Pd, Pd_ref, 1e-5, "P dropped")
    close(W, Pd_ref.reshape(B, H, tq, tk).mean(1), 1e-5, "head mean")
    dS = ops.softmax_bwd(desc, R.float().cuda().clone())
    close(dS, Sd.grad, 1e-5, "dS")


@pytest.mark.parametrize("dh,tq,tk", [(16, 40, 72), (15, 7, 4), (128, 130, 64)])
def test_strided_batched_gemm(ops, dh, tq, tk):
    """The three batched layouts sdumc_mha_* uses, straight on [T, B, H*dh] activations."""
    B, H = 3, 4
    E = H * dh
    g = torch.Generator().manual_seed(dh)
    q, k = torch.randn(tq, B, E, generator=g).cuda(), torch.randn(tk, B, E, generator=g).cuda()
    heads = lambda t: t.reshape(t.shape[0], B * H, dh).transpose(0, 1).double()    # [BH, T, dh]
    S = torch.empty(B * H, tq, tk, device="cuda")
    ops.gemm(ops.NT, q, k, tq, tk, dh, C_out=S, lda=B * E, ldb=B * E, ldc=tk, batch=B * H, stride_a=dh, stride_b=dh,
             stride_c=tq * tk, splitk=0)
    close(S, torch.bmm(heads(q), heads(k).transpose(1, 2)), 1e-5, "NT")
    O = torch.empty(tq, B, E, device="cuda")
    ops.gemm(ops.NN, S, k, tq, dh, tk, C_out=O, lda=tk, ldb=B * E, ldc=B * E, batch=B * H, stride_a=tq * tk, stride_b=dh,
             stride_c=dh, splitk=0)
    close(heads(O), torch.bmm(S.double(), heads(k)), 1e-5, "NN")
    D = torch.empty(tk, B, E, device="cuda")
    ops.gemm(ops.TN, S, q, tk, dh, tq, C_out=D, lda=tk, ldb=B * E, ldc=B * E, batch=B * H, stride_a=tq * tk, stride_b=dh,
             stride_c=dh, splitk=0)
    close(heads(D), torch.bmm(S.double().transpose(1, 2), heads(q)), 1e-5, "TN")


def test_drop_add_positions_and_residual(ops):
    from oracle import philox, transformer_oracle as TO
    from sdumc_amd._lib import make_dropout
    for (T_, B, E) in [(6, 3, 32), (5, 2, 30)]:
        g = torch.Generator().manual_seed(E)
        x = torch.randn(T_, B, E, generator=g)
        x[1, 0, 0] = 0.0
        x[4, 1, 0] = 0.0
        res = torch.randn(T_, B, E, generator=g)
        tab = TO.sinusoidal_table(T_ + 1, E)
        drop = make_dropout(True, 3, 0.2, B, E, T_, 0, 9, 77)
        m = torch.from_numpy(philox.dropout_mask(T_, B, E, 0.2, 77, 9, 3))
        pos = torch.where(x[:, :, 0] != 0, torch.arange(1, T_ + 1).unsqueeze(1).expand(T_, B), torch.zeros(T_, B, dtype=torch.long))
        want = (math.sqrt(E) * x + tab[pos]) * m + res
        got = ops.drop_add(x.cuda(), res.cuda(), drop, math.sqrt(E), tab.cuda(), x.cuda())
        close(got, want, 1e-6, "drop_add")
        assert torch.equal((got.cpu() - res) == 0, (want - res) == 0)


# ---------------------------------------------------------------------------------------------------
# modules vs the reference's goldens
# ---------------------------------------------------------------------------------------------------
def test_modules_refuse_cpu_tensors(te):
    from sdumc_amd._lib import SdumcError
    m = te.MultiheadAttention(32, 4)
    with pytest.raises(SdumcError):
        m(torch.randn(3, 2, 32), torch.randn(3, 2, 32), torch.randn(3, 2, 32))


def test_state_dict_names_match_reference(te, golden):
    g = golden("transformer")
    enc = te.TransformerEncoder(32, 4, 2, position_embedding=True)
    want = sorted(k[len("enc/P/"):] for k in g.files if k.startswith("enc/P/"))
    assert sorted(enc.state_dict()) == want
    for k, v in enc.state_dict().items():
        assert tuple(v.shape) == g["enc/P/" + k].shape, k


def test_mha_golden_self_eval(te, golden):
    g = golden("transformer")
    m = load_params(te.MultiheadAttention(32, 4), g, "mha_self").eval()
    x = G(g["mha_self/x"])
    o, w = m(x, x, x)
    close(o, g["mha_self/out"], what="out"); close(w, g["mha_self/weights"], what="weights")


def test_mha_golden_cross_train_grads(te, golden):
    g = golden("transformer")
    m = load_params(te.MultiheadAttention(32, 4, attn_dropout=0.25), g, "mha_cross").train()
    q, k, v = (G(g[f"mha_cross/{n}"]).requires_grad_() for n in "qkv")
    seed, call = (int(i) for i in g["mha_cross/seed_call"])
    te.manual_seed(seed, call)
    mask = te.buffered_future_mask(q, k)
    assert torch.equal(mask.cpu(), T(g["mha_cross/mask"]))
    o, w = m(q, k, v, attn_mask=mask)
    close(o, g["mha_cross/out"], what="out"); close(w, g["mha_cross/weights"], what="weights")
    # bit-exact mask indexing: the zero pattern of the head-averaged weights is the future mask AND'ed over heads' drops
    assert torch.equal(w.cpu() == 0, T(g["mha_cross/weights"]) == 0)
    (o * G(g["mha_cross/R"])).sum().backward()
    check_grads(m, g, "mha_cross", q=q, k=k, v=v)


def test_mha_golden_add_bias_kv_and_zero_attn(te, golden):
    """add_bias_kv / add_zero_attn (multihead_attention.py:28-38, :86-104): outputs, weights over the lengthened source and
    every gradient (incl. bias_k / bias_v) against the real module."""
    g = golden("transformer_kv")
    m = load_params(te.MultiheadAttention(32, 4, attn_dropout=0.25, add_bias_kv=True, add_zero_attn=True), g, "both").train()
    q, k, v = (G(g[f"both/{n}"]).requires_grad_() for n in "qkv")
    seed, call = (int(i) for i in g["both/seed_call"])
    te.manual_seed(seed, call)
    o, w = m(q, k, v, attn_mask=te.buffered_future_mask(q, k))
    assert tuple(w.shape) == (2, 5, 11)
    close(o, g["both/out"], what="out"); close(w, g["both/weights"], what="weights")
    assert torch.equal(w.cpu() == 0, T(g["both/weights"]) == 0)
    (o * G(g["both/R"])).sum().backward()
    check_grads(m, g, "both", q=q, k=k, v=v)
    m = load_params(te.MultiheadAttention(32, 4, add_zero_attn=True), g, "zero").eval()
    x = G(g["zero/x"]).requires_grad_()
    o, w = m(x, x, x)
    close(o, g["zero/out"], what="out"); close(w, g["zero/weights"], what="weights")
    (o * G(g["zero/R"])).sum().backward()
    check_grads(m, g, "zero", x=x)
    m = load_params(te.MultiheadAttention(30, 2, add_bias_kv=True), g, "bias").eval()
    q, kv = G(g["bias/q"]).requires_grad_(), G(g["bias/kv"]).requires_grad_()
    o, w = m(q, kv, kv)
    close(o, g["bias/out"], what="out"); close(w, g["bias/weights"], what="weights")
    (o * G(g["bias/R"])).sum().backward()
    check_grads(m, g, "bias", q=q, kv=kv)


def test_mha_golden_odd_head_dim_kv_alias(te, golden):
    g = golden("transformer")
    m = load_params(te.MultiheadAttention(30, 2), g, "mha_odd").eval()
    q, kv = G(g["mha_odd/q"]).requires_grad_(), G(g["mha_odd/kv"]).requires_grad_()
    o, w = m(q, kv, kv)
    close(o, g["mha_odd/out"], what="out"); close(w, g["mha_odd/weights"], what="weights")
    (o * G(g["mha_odd/R"])).sum().backward()
    check_grads(m, g, "mha_odd", q=q, kv=kv)


def test_layernorm_golden(te, golden):
    g = golden("transformer")
    ln = load_params(te.LayerNorm(300), g, "ln")
    x = G(g["ln/x"]).requires_grad_()
    y = ln(x)
    close(y, g["ln/y"], 1e-5, "y")
    (y * G(g["ln/R"])).sum().backward()
    check_grads(ln, g, "ln", x=x)


def test_encoder_layer_golden(te, golden):
    g = golden("transformer")
    lay = load_params(te.TransformerEncoderLayer(32, num_heads=4, attn_dropout=0.1, relu_dropout=0.2, res_dropout=0.3,
                                                 attn_mask=True), g, "layer").train()
    seed, call = (int(i) for i in g["layer/seed_call"])
    R = G(g["layer/R"])
    x = G(g["layer/x"]).requires_grad_()
    te.manual_seed(seed, call)
    y = lay(x)
    close(y, g["layer/self_out"], what="self out")
    (y * R).sum().backward()
    check_grads(lay, g, "layer/self", x=x)
    x, xk, xv = (G(g[f"layer/{n}"]).requires_grad_() for n in ("x", "xk", "xv"))
    te.manual_seed(seed, call + 1)
    y = lay(x, xk, xv)
    close(y, g["layer/cross_out"], what="cross out")
    (y * R).sum().backward()
    check_grads(lay, g, "layer/cross", x=x, xk=xk, xv=xv)
    lay.eval()
    with torch.no_grad():
        close(lay(x), g["layer/self_eval_out"], what="eval out")


def test_encoder_golden(te, golden):
    g = golden("transformer")
    enc = load_params(te.TransformerEncoder(32, 4, 2, attn_dropout=0.1, relu_dropout=0.1, res_dropout=0.2,
                                            embed_dropout=0.15, attn_mask=True, position_embedding=True), g, "enc")
    seed, call = (int(i) for i in g["enc/seed_call"])
    R = G(g["enc/R"])
    x = G(g["enc/x"]).requires_grad_()
    enc.eval()
    with torch.no_grad():
        close(enc(x), g["enc/self_eval_out"], what="eval out")
    enc.train()
    te.manual_seed(seed, call)
    y = enc(x)
    close(y, g["enc/self_out"], what="self out")
    (y * R).sum().backward()
    check_grads(enc, g, "enc/self", x=x)
    x, xk, xv = (G(g[f"enc/{n}"]).requires_grad_() for n in ("x", "xk", "xv"))
    te.manual_seed(seed, call + 1)
    y = enc(x, xk, xv)
    close(y, g["enc/cross_out"], what="cross out")
    (y * R).sum().backward()
    check_grads(enc, g, "enc/cross", x=x, xk=xk, xv=xv)
    plain = load_params(te.TransformerEncoder(32, 4, 1), g, "enc_plain").eval()
    with torch.no_grad():
        close(plain(G(g["enc_plain/x"])), g["enc_plain/out"], what="plain out")


# ---------------------------------------------------------------------------------------------------
# larger seeded shapes vs the CPU oracle (float64)
# ---------------------------------------------------------------------------------------------------
def _oracle_params(mod):
    return {k: v.detach().cpu().double().requires_grad_(v.requires_grad and v.dtype.is_floating_point and k != "version")
            for k, v in mod.state_dict(keep_vars=True).items()}


@pytest.mark.parametrize("E,H,T_,B", [(256, 8, 64, 4), (1024, 8, 512, 2)])
def test_encoder_layer_vs_oracle_large(te, E, H, T_, B):
    """(1024, 8, 512) is the per-token shape of BASELINE configs[4] (C5)."""
    from oracle import transformer_oracle as TO
    torch.manual_seed(E)
    lay = te.TransformerEncoderLayer(E, num_heads=H, attn_dropout=0.1, relu_dropout=0.1, res_dropout=0.1, attn_mask=True).cuda().train()
    x = torch.randn(T_, B, E)
    R = torch.randn(T_, B, E) / math.sqrt(T_ * B * E)
    xg = x.cuda().requires_grad_()
    te.manual_seed(4242, 8)
    y = lay(xg)
    (y * R.cuda()).sum().backward()
    P = _oracle_params(lay)
    xo = x.double().requires_grad_()
    yo = TO.encoder_layer(P, "", xo, H, TO.DropSeq(True, 4242, 8), 0.1, 0.1, 0.1, True)
    (yo * R.double()).sum().backward()
    close(y, yo, what="out")
    close(xg.grad, xo.grad, what="dx")
    for k, v in lay.named_parameters():
        close(v.grad, P[k].grad, what=f"d{k}")


def test_mha_properties_at_c5_shape(te):
    """Size-independent properties at the C5 shape (T=512, E=1024, H=8, B=32 = one rank's shard of batch 256):
    weights rows sum to 1, causal zeros are exact, and linearity of the output in `value`."""
    torch.manual_seed(0)
    E, H, T_, B = 1024, 8, 512, 32
    m = te.MultiheadAttention(E, H).cuda().eval()
    q, k, v1, v2 = (torch.randn(T_, B, E, device="cuda") for _ in range(4))
    mask = te.buffered_future_mask(q, k)
    with torch.no_grad():
        o1, w = m(q, k, v1, attn_mask=mask)
        o2, _ = m(q, k, v2, attn_mask=mask)
        o12, _ = m(q, k, v1 + 2 * v2, attn_mask=mask)
        b = m.out_proj(torch.zeros(1, 1, E, device="cuda")) - 0   # out_proj bias
        bv = m.in_proj_bias[2 * E:]
    assert torch.isfinite(o1).all()
    close(w.sum(-1), torch.ones(B, T_), 1e-5, "rows sum to one")
    assert torch.equal(w == 0, (mask == float("-inf")).unsqueeze(0).expand(B, T_, T_)), "causal zeros"
    # out is affine in value: o(v1 + 2 v2) = o(v1) + 2 o(v2) - 2 * o(0); with zero biases o(0) = 0
    assert float(b.detach().abs().max()) == 0.0 and float(bv.detach().abs().max()) == 0.0
    close(o12, o1 + 2 * o2, 1e-4, "linearity in value")


def test_encoder_layer_bf16_operand_mode(te):
    """bf16-operand mode of every product in the layer (BASELINE configs[4] dtype): outputs within 2e-2 of the fp32 oracle,
    the mode is really active, gradients norm-wise within bf16 tolerance; exact fp32 is restored afterwards."""
    from oracle import transformer_oracle as TO
    E, H, T_, B = 256, 8, 64, 4
    torch.manual_seed(E)
    lay = te.TransformerEncoderLayer(E, num_heads=H, attn_dropout=0.1, relu_dropout=0.1, res_dropout=0.1, attn_mask=True).cuda().train()
    x = torch.randn(T_, B, E)
    R = torch.randn(T_, B, E) / math.sqrt(T_ * B * E)
    P = _oracle_params(lay)
    xo = x.double().requires_grad_()
    yo = TO.encoder_layer(P, "", xo, H, TO.DropSeq(True, 99, 3), 0.1, 0.1, 0.1, True)
    (yo * R.double()).sum().backward()
    outs = {}
    try:
        for mode in (False, True):
            te.set_bf16(mode)
            xg = x.cuda().requires_grad_()
            te.manual_seed(99, 3)
            y = lay(xg)
            (y * R.cuda()).sum().backward()
            outs[mode] = (y.detach().clone(), xg.grad.clone(), {k: v.grad.clone() for k, v in lay.named_parameters()})
            for v in lay.parameters():
                v.grad = None
    finally:
        te.set_bf16(False)
    close(outs[False][0], yo, what="fp32 out")
    close(outs[True][0], yo, 2e-2, "bf16 out")
    assert float((outs[True][0] - outs[False][0]).abs().max()) > 1e-6, "bf16 mode not active"
    rel = lambda a, b: float((a.cpu().double() - b).norm() / (b.norm() + 1e-12))
    assert rel(outs[True][1], xo.grad) < 3e-2
    for k, v in outs[True][2].items():
        assert rel(v, P[k].grad) < 5e-2, k
