"""Pins the CPU oracle (oracle/) to golden vectors produced by the REAL reference
modules (tests/golden/make_goldens.py).  CPU-only."""
import numpy as np
import pytest
import torch

from oracle import philox, sdumc_oracle as O


def T(a):
    return torch.from_numpy(np.asarray(a))


def close(a, b, tol=2e-6):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, np.asarray(b), rtol=tol, atol=tol)


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32-10
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = philox.philox4x32_10(*[np.uint32(c) for c in ctr], key[0], key[1])
        assert tuple(int(g) for g in got) == want


def test_dropout_mask_statistics_and_sharding():
    m = philox.dropout_mask(8, 50, 256, 0.3, seed=9, call=1, site=6)
    assert set(np.unique(m)) == {np.float32(0), philox.drop_scale(0.3)}
    assert abs((m > 0).mean() - 0.7) < 0.01
    # a batch shard draws the same masks as the unsharded batch
    shard = philox.dropout_mask(3, 50, 256, 0.3, seed=9, call=1, site=6, sample0=5)
    np.testing.assert_array_equal(shard, m[5:8])
    assert philox.drop_threshold(0.5) == 1 << 31 and philox.drop_scale(0.5) == 2.0


def _block_params(g):
    P = {"fra2utt_1.attention_context_vector": T(g["fra_ctx"]),
         "fra2utt_1.input_proj.weight": T(g["fra_w"]), "fra2utt_1.input_proj.bias": T(g["fra_b"]),
         "cross_att_fra2utt_2.query_proj.weight": T(g["ca_wq"]), "cross_att_fra2utt_2.query_proj.bias": T(g["ca_bq"]),
         "cross_att_fra2utt_2.input_proj.weight": T(g["ca_wi"]), "cross_att_fra2utt_2.input_proj.bias": T(g["ca_bi"])}
    return P


def test_blocks_eval_and_train(golden):
    g = golden("blocks")
    P = _block_params(g)
    x, q = T(g["x"]), T(g["q"])
    o, a = O.fra2utt(P, 1, x, O.DropCtx("eval"))
    close(o, g["fra_eval_out"]); close(a, g["fra_eval_att"])
    o, a = O.cross_attention(P, 2, q, x, O.DropCtx("eval"))
    close(o, g["ca_eval_out"]); close(a, g["ca_eval_att"])
    d = O.DropCtx("philox", int(g["seed"]), int(g["call"]))
    o, a = O.fra2utt(P, 1, x, d)
    close(o, g["fra_train_out"]); close(a, g["fra_train_att"])
    o, a = O.cross_attention(P, 2, q, x, d)
    close(o, g["ca_train_out"]); close(a, g["ca_train_att"])


def _digest(t, key):
    import tests.golden.make_goldens as mg
    return mg.digest(t, key)


def test_param_init_reproducible(golden):
    g = golden("forward")
    P = O.init_params(tuple(int(v) for v in g["dims"]), seed=int(g["pseed"]))
    assert list(P) == list(O.param_shapes(tuple(g["dims"])))
    from tests.golden.make_goldens import digest
    got = np.stack([digest(P[k], k) for k in P])
    np.testing.assert_allclose(got, g["param_digest"], rtol=1e-12, atol=1e-12)
    n = sum(int(np.prod(s)) for s in O.param_shapes((1024, 4096, 1024, 4096)).values())
    assert n == 4268884
    dead = sum(int(np.prod(s)) for k, s in O.param_shapes((1024, 4096, 1024, 4096)).items() if O.is_dead(k))
    assert dead == 411593


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_forward_both_streams(golden, mode):
    g = golden("forward")
    P = O.init_params(tuple(int(v) for v in g["dims"]), seed=int(g["pseed"]))
    audio, video = T(g["audio"]), T(g["video"])
    names = ("vals", "fused", "rnc", "text_hidden", "cross_text")
    for s, tx in enumerate((T(g["text"]), T(g["feat4"]))):
        drop = O.DropCtx("eval") if mode == "eval" else O.DropCtx("philox", int(g["seed"]), 2 * int(g["step"]) + s)
        y, emb = O.forward(P, audio, tx, video, drop)
        for n, t in zip(names, [y] + list(emb)):
            close(t, g[f"{mode}{s}_{n}"], 5e-6)


def test_losses(golden):
    g = golden("losses")
    pred = T(g["mse_pred"]).requires_grad_()
    l = O.mse_loss(pred, T(g["mse_tgt"]))
    l.backward()
    close(l, g["mse"]); close(pred.grad, g["mse_dpred"])
    for tag in ("2d", "3d"):
        a, b = T(g[f"rmse{tag}_a"]).requires_grad_(), T(g[f"rmse{tag}_b"]).requires_grad_()
        l = O.rmse_loss(a, b)
        l.backward()
        close(l, g[f"rmse{tag}"]); close(a.grad, g[f"rmse{tag}_da"]); close(b.grad, g[f"rmse{tag}_db"])
    for tag in ("rnc", "rnctie"):
        f = T(g[f"{tag}_f"]).requires_grad_()
        y = T(g[f"{tag}_y"])
        l = O.rnc_loss(f, y)
        l.backward()
        close(l, g[tag]); close(f.grad, g[f"{tag}_df"], 1e-5)
        # boolean neg_mask: bit-exact
        np.testing.assert_array_equal(O.rnc_masks(y.repeat(2, 1)).numpy(), g[f"{tag}_mask"])


def test_train_step(golden):
    g = golden("step")
    from tests.golden.make_goldens import digest
    dims = tuple(int(v) for v in g["dims"])
    P = O.init_params(dims, seed=int(g["pseed"]))
    before = {k: v.clone() for k, v in P.items()}
    state = {}
    loss, terms, grads, outs = O.train_step(
        P, state, T(g["audio"]), T(g["text"]), T(g["video"]), T(g["feat4"]), T(g["vals"]),
        weights=tuple(g["weights"]), mode="philox", seed=int(g["seed"]), step=int(g["step"]))
    close(loss, g["loss"], 1e-5)
    np.testing.assert_allclose([float(t) for t in terms], g["terms"], rtol=1e-5, atol=1e-6)
    close(outs[0][0], g["y0"], 1e-5); close(outs[1][0], g["y1"], 1e-5)
    names = [str(n) for n in g["names"]]
    dead = {str(n) for n in g["dead"]}
    assert dead == {k for k in P if O.is_dead(k)}
    assert set(grads) == set(names) - dead
    for i, k in enumerate(names):
        if k in dead:
            assert torch.equal(P[k], before[k])           # Adam skips grad-None params
            continue
        got = digest(grads[k], k)
        scale = max(1e-6, abs(g["grad_digest"][i][1]))
        np.testing.assert_allclose(got, g["grad_digest"][i], rtol=2e-4, atol=2e-5 * scale, err_msg=k)
        if "grad__" + k in g.files:
            np.testing.assert_allclose(grads[k].numpy(), g["grad__" + k], rtol=1e-4, atol=1e-6, err_msg=k)
        if "delta__" + k in g.files:
            np.testing.assert_allclose(((P[k] - before[k]) * 1e4).numpy(), g["delta__" + k], rtol=2e-3, atol=2e-3, err_msg=k)
    lrs = [O.lr_lambda(e) for e in range(40)]
    np.testing.assert_allclose(lrs, g["lr_table"], rtol=1e-12)


def test_collate_contract(golden):
    """right-zero-pad per modality to the batch max; pads = max - len (read_data.py:223-248)"""
    g = golden("collate")
    lens = g["lens"]
    for i, name in enumerate(("audio", "text", "video", "feat4")):
        mx = lens[:, i].max()
        np.testing.assert_array_equal(g["pads"][i], mx - lens[:, i])
        stacked = g[name + "s"]
        for b in range(len(lens)):
            raw = g[f"raw_{name}_{b}"]
            np.testing.assert_array_equal(stacked[b, :len(raw)], raw)
            assert not stacked[b, len(raw):].any()


def test_blocks_default_width_1024(golden):
    """The oracle's block restatements at the reference constructors' default input_dim = 1024 against the real reference
    (tests/golden/blocks1024.npz).  Weights are re-drawn from the recorded seeds through the drop-in constructors (same
    layers, same order as the reference's), digests pin the stream."""
    from sdumc_amd import blocks
    from tests.golden.make_goldens import digest
    g = golden("blocks1024")
    torch.manual_seed(int(g["seed_fra"]))
    fra = blocks.FRA2UTT_new()
    torch.manual_seed(int(g["seed_ca"]))
    ca = blocks.Cross_Attention()
    np.testing.assert_allclose(digest(fra.input_proj.weight, "fra.w"), g["fra_w_digest"], rtol=1e-12)
    np.testing.assert_allclose(digest(ca.input_proj.weight, "ca.wi"), g["ca_wi_digest"], rtol=1e-12)
    P = {"fra2utt_0.attention_context_vector": fra.attention_context_vector.detach(),
         "fra2utt_0.input_proj.weight": fra.input_proj.weight.detach(), "fra2utt_0.input_proj.bias": fra.input_proj.bias.detach(),
         "cross_att_fra2utt_0.query_proj.weight": ca.query_proj.weight.detach(),
         "cross_att_fra2utt_0.query_proj.bias": ca.query_proj.bias.detach(),
         "cross_att_fra2utt_0.input_proj.weight": ca.input_proj.weight.detach(),
         "cross_att_fra2utt_0.input_proj.bias": ca.input_proj.bias.detach()}
    x, q = T(g["x"]), T(g["q"])
    o, a = O.fra2utt(P, 0, x, O.DropCtx("eval"))
    close(o, g["fra_eval_out"], 5e-6); close(a, g["fra_eval_att"], 5e-6)
    o, a = O.cross_attention(P, 0, q, x, O.DropCtx("eval"))
    close(o, g["ca_eval_out"], 5e-6); close(a, g["ca_eval_att"], 5e-6)
    d = O.DropCtx("philox", int(g["seed"]), int(g["call"]))
    o, a = O.fra2utt(P, 0, x, d)
    close(o, g["fra_train_out"], 5e-6); close(a, g["fra_train_att"], 5e-6)
    o, a = O.cross_attention(P, 0, q, x, d)
    close(o, g["ca_train_out"], 5e-6); close(a, g["ca_train_att"], 5e-6)
