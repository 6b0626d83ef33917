"""The drop-in Python surface (sdumc_amd/model.py, sdumc_amd/loss.py) against the reference's contract
(SURVEY §8b): constructor, state_dict names/shapes/order, initialisation stream, forward tuple,
autograd + torch.optim.Adam exactly as main_frame_val_text_missing.py:119-150 drives them."""
import types

import numpy as np
import pytest
import torch


def _args(dims, model="wengnet_mosei_mult_views_text_missing"):
    return types.SimpleNamespace(input_dims=dims, model=model)


def test_state_dict_and_init_stream_match_the_reference(golden):
    from sdumc_amd.model import WengnetMOSEIMultViewsTextMissing, get_models
    from tests.golden.make_goldens import digest
    g = golden("init")
    dims = tuple(int(v) for v in g["dims"])
    torch.manual_seed(0)
    net = WengnetMOSEIMultViewsTextMissing(_args(dims))
    names = [str(n) for n in g["names"]]
    assert [k for k, _ in net.named_parameters()] == names          # same registration order
    sd = net.state_dict()
    assert list(sd) == names
    for i, k in enumerate(names):
        shp = [int(v) for v in g["shapes"][i] if v]
        assert list(sd[k].shape) == shp, k
        np.testing.assert_allclose(digest(sd[k], k), g["digest"][i], rtol=1e-12, atol=1e-12, err_msg=k)
    np.testing.assert_array_equal(sd["fc_att.weight"].numpy(), g["fc_att_weight"])
    assert sum(p.numel() for p in net.parameters()) == sum(int(np.prod([v for v in s if v])) for s in g["shapes"])
    # the get_models wrapper prefixes keys with "model." (toolkit/models/__init__.py:67); the inference script
    # strips "module." and loads with strict=False (main_frame_val_text_missing_inference.py:341)
    wrapped = get_models(_args(dims))
    assert all(k.startswith("model.") for k in wrapped.state_dict())
    wrapped.load_state_dict({"model." + k: v for k, v in sd.items()}, strict=True)
    assert torch.equal(wrapped.model.state_dict()["fc_att.weight"], sd["fc_att.weight"])
    # parameters are views of one flat buffer, also after load_state_dict
    flat = wrapped.model._flat
    off = wrapped.model._layout.entries["fc_att.weight"][0]
    assert wrapped.model._get("fc_att.weight").data_ptr() == flat.data_ptr() + 4 * off
    with pytest.raises(Exception):
        get_models(_args(dims, model="tfn"))                         # not part of the hot path: fails loudly


def test_cpu_forward_fails_loudly():
    from sdumc_amd.model import WengnetMOSEIMultViewsTextMissing
    from sdumc_amd._lib import SdumcError
    net = WengnetMOSEIMultViewsTextMissing(_args((16, 8, 16, 8)))
    with pytest.raises(SdumcError):
        net([torch.zeros(2, 3, 16), torch.zeros(2, 3, 8), torch.zeros(2, 3, 16), False])


@pytest.mark.gpu
def test_reference_training_loop_with_dropin_modules(golden):
    """main :119-150 verbatim, with our model / losses / torch.optim.Adam, against the golden step
    recorded from the real reference (same Philox masks)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import sdumc_oracle as O
    from sdumc_amd.model import get_models
    from sdumc_amd.loss import MSELoss, RMSELoss, RnCLoss
    from tests.golden.make_goldens import digest
    g = golden("step")
    dims = tuple(int(v) for v in g["dims"])
    args = _args(dims)
    model = get_models(args)
    model.load_state_dict({"model." + k: v for k, v in O.init_params(dims, seed=int(g["pseed"])).items()})
    model = model.cuda()
    model.model.seed = int(g["seed"])
    model.model._calls = 2 * int(g["step"])
    losses = {'reg_loss': MSELoss().cuda(), 'rmse_loss': RMSELoss().cuda(), 'rnc_loss': RnCLoss().cuda()}
    optimizer = torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=1e-5)
    w = [float(v) for v in g["weights"]]
    T = lambda k: torch.from_numpy(g[k]).cuda()
    audio_feat, text_feat, visual_feat, feat4_feat, vals = T("audio"), T("text"), T("video"), T("feat4"), T("vals")
    before = {k: v.detach().clone() for k, v in model.model.named_parameters()}
    model.train()
    optimizer.zero_grad()
    vals_out_0, embeddings_0 = model([audio_feat, text_feat, visual_feat, False])
    features_0, rnc_feat_0, text_feat_0, text_query_feat_0 = embeddings_0
    vals_out_1, embeddings_1 = model([audio_feat, feat4_feat, visual_feat, True])
    features_1, rnc_feat_1, text_feat_1, text_query_feat_1 = embeddings_1
    n_views_feature = torch.stack((rnc_feat_0, rnc_feat_1), dim=1)
    MSEloss_0 = losses['reg_loss'](vals_out_0, vals)
    MSEloss_1 = losses['reg_loss'](vals_out_1, vals)
    rnc_loss = losses['rnc_loss'](n_views_feature, vals.unsqueeze(1))
    terms = [MSEloss_0, MSEloss_1, losses['rmse_loss'](text_feat_1, text_feat_0.detach()),
             losses['rmse_loss'](text_query_feat_1, text_query_feat_0.detach()),
             losses['rmse_loss'](features_1, features_0), rnc_loss]
    loss = sum(wi * t for wi, t in zip(w, terms))
    loss.backward()
    optimizer.step()
    np.testing.assert_allclose(float(loss), float(g["loss"]), rtol=2e-5)
    np.testing.assert_allclose([float(t) for t in terms], g["terms"], rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(vals_out_1.detach().cpu().numpy(), g["y1"], rtol=1e-4, atol=1e-5)
    dead = {str(n) for n in g["dead"]}
    for i, k in enumerate(str(n) for n in g["names"]):
        p = model.model._get(k)
        if k in dead:
            assert p.grad is None and torch.equal(p.detach(), before[k]), k
            continue
        scale = max(1e-6, abs(g["grad_digest"][i][1]))
        np.testing.assert_allclose(digest(p.grad.cpu(), k), g["grad_digest"][i], rtol=5e-4, atol=5e-5 * scale + 1e-6, err_msg=k)
        if "delta__" + k in g.files:
            ok = np.abs(g["grad__" + k].reshape(p.shape)) > 1e-5
            np.testing.assert_allclose(((p.detach() - before[k]) * 1e4).cpu().numpy()[ok], g["delta__" + k].reshape(p.shape)[ok],
                                       rtol=5e-3, atol=5e-3, err_msg=k)
    # eval mode: both streams under no_grad, as main :151-154
    model.eval()
    with torch.no_grad():
        y0, _ = model([audio_feat, text_feat, visual_feat, False])
        y0b, _ = model([audio_feat, text_feat, visual_feat, False])
    assert y0.shape == (audio_feat.shape[0], 1) and torch.equal(y0, y0b)
    # B == 1 works (the reference crashes on .squeeze(), model :308 — documented divergence)
    with torch.no_grad():
        y1, emb = model([audio_feat[:1].contiguous(), text_feat[:1].contiguous(), visual_feat[:1].contiguous(), False])
    assert y1.shape == (1, 1) and emb[3].shape == (1, 7, 128)
    np.testing.assert_allclose(y1.cpu().numpy(), y0[:1].cpu().numpy(), rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
def test_device_feature_store_equals_reference_collater(golden):
    """F1: batches assembled on the GPU from the packed device-resident store == the reference collater's output."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from sdumc_amd.data import DeviceFeatureStore, collate
    g = golden("collate")
    n = len(g["lens"])
    inst = [{k: g[f"raw_{k}_{b}"] for k in ("audio", "text", "video", "feat4")} | {"emo": 0, "val": 0.25 * b, "name": f"u{b}"}
            for b in range(n)]
    # feature widths must be multiples of 4 for the 16-byte kernel: pad the golden's odd widths consistently on both sides
    def pad4(a):
        d = (-a.shape[1]) % 4
        return np.pad(a, ((0, 0), (0, d))) if d else a
    inst = [{k: (pad4(v) if isinstance(v, np.ndarray) else v) for k, v in i.items()} for i in inst]
    store = DeviceFeatureStore(inst)
    for order in ([0, 1, 2], [2, 0], [1]):
        batch, pads, emos, vals, names = store.batch(order)
        ref, rpads, remos, rvals, rnames = collate([inst[i] for i in order])
        for k in ("audios", "texts", "videos", "feat4s"):
            assert torch.equal(batch[k].cpu(), ref[k]), k
        assert pads == rpads and names == rnames and torch.equal(vals.cpu(), rvals)
    # unpadded widths: the first golden batch reproduces the reference's stacked tensors on the original columns
    batch, *_ = store.batch([0, 1, 2])
    for k, key in (("audio", "audios"), ("text", "texts"), ("video", "videos"), ("feat4", "feat4s")):
        np.testing.assert_array_equal(batch[key].cpu().numpy()[:, :, :g[key].shape[2]], g[key])


def test_checkpoint_roundtrip_in_reference_format(tmp_path):
    """F2: {'epoch','state_dict','optimizer'} with 'model.'-prefixed keys; loads with a 'module.' prefix too;
    torch.optim.Adam accepts the optimizer state built from the fused step's flat moments."""
    from sdumc_amd.model import get_models
    from sdumc_amd import checkpoint as ck
    dims = (16, 8, 12, 8)
    torch.manual_seed(3)
    a = get_models(_args(dims))
    net = a.model
    m = torch.arange(net._layout.live, dtype=torch.float32) * 1e-6
    v = torch.arange(net._layout.live, dtype=torch.float32) * 1e-9
    opt_state = ck.adam_state_from_flat(net, m, v, step=7)
    path = str(tmp_path / "mosei_mult-view_kd_full_0.5_17.pt")
    ck.save_checkpoint(path, a, opt_state, epoch=17)
    raw = torch.load(path, weights_only=False)
    assert set(raw) == {"epoch", "state_dict", "optimizer"} and all(k.startswith("model.") for k in raw["state_dict"])
    assert sum(v.numel() for v in raw["state_dict"].values()) == sum(p.numel() for p in a.parameters())
    # DataParallel-style prefix, as the reference's inference script expects to strip
    raw["state_dict"] = {"module." + k: v for k, v in raw["state_dict"].items()}
    torch.save(raw, path)
    torch.manual_seed(4)
    b = get_models(_args(dims))
    epoch, opt, missing, unexpected = ck.load_checkpoint(path, b)
    assert epoch == 17 and not missing and not unexpected
    for (ka, pa), (kb, pb) in zip(a.named_parameters(), b.named_parameters()):
        assert ka == kb and torch.equal(pa, pb)
    optim = torch.optim.Adam(b.parameters(), lr=1e-4, weight_decay=1e-5)
    optim.load_state_dict(opt)                                  # torch accepts it
    m2, v2, step = ck.flat_from_adam_state(b.model, opt, "cpu")
    assert step == 7
    lay = b.model._layout
    for name in lay.live_names():                                # (alignment padding between tensors carries no state)
        off, shape, _ = lay.entries[name]
        n = int(np.prod(shape))
        assert torch.equal(m2[off:off + n], m[off:off + n]) and torch.equal(v2[off:off + n], v[off:off + n]), name
    dead = [i for i, n in enumerate(b.model._pnames) if not b.model._layout.entries[n][2]]
    assert dead and all(i not in opt["state"] for i in dead)     # grad-None parameters carry no Adam state


@pytest.mark.gpu
def test_inference_export_loop(golden):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import sdumc_oracle as O
    from sdumc_amd.model import get_models
    from sdumc_amd.checkpoint import run_inference
    g = golden("forward")
    dims = tuple(int(v) for v in g["dims"])
    model = get_models(_args(dims))
    model.load_state_dict({"model." + k: v for k, v in O.init_params(dims, seed=int(g["pseed"])).items()})
    model = model.cuda()
    T = lambda k: torch.from_numpy(g[k])
    data = ({"audios": T("audio"), "texts": T("text"), "videos": T("video"), "feat4s": T("feat4")}, None,
            torch.zeros(4), T("vals"), [f"u{i}" for i in range(4)])
    res = run_inference(model, [data, data])
    assert res["val_preds_full"].shape == (8, 1) and res["text_rep_full"].shape == (8, 7, 128) and len(res["names"]) == 8
    np.testing.assert_allclose(res["val_preds_full"][:4], g["eval0_vals"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(res["missing_rnc"][4:], g["eval1_rnc"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(res["text_rep_query_full"][:4], g["eval0_text_hidden"], rtol=2e-5, atol=2e-6)
