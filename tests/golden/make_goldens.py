#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REAL reference.

Runs only in the build container (needs /root/reference).  It imports the
reference's own modules by path — never copies them — and records inputs and
expected outputs as small .npz files:

  blocks.npz      FRA2UTT_new / Cross_Attention, eval and Philox-replayed train mode
  blocks1024.npz  the same two blocks at the constructors' default input_dim = 1024 (weights re-drawn from a seed)
  forward.npz     full-model forward, both streams, eval and train(Philox) mode
  losses.npz      MSELoss / RMSELoss / RnCLoss (incl. tied labels) values + input grads
  step.npz        one full two-stream train step: loss, 6 terms, per-parameter gradient
                  digests, post-Adam parameter digests, LR-lambda table
  collate.npz     pad_to_maxlen_pre_modality_tensor_4 on ragged inputs
  transformer.npz the generic transformers_encoder package (MultiheadAttention, LayerNorm,
                  TransformerEncoderLayer, TransformerEncoder): parameters, inputs, outputs, gradients

Parameters are not stored (the model has 2.7 M non-input parameters because
general_dim=256 is hard-coded, model :191): they are regenerated from
oracle.init_params(seed) and loaded into the reference model with
load_state_dict; a digest of them is stored to detect RNG drift.  Gradients are
stored as digests (sum, abs-sum, projection on a seeded random vector, float64)
plus full tensors for the small parameters.

Dropout: nn.Dropout modules of the reference are driven through a patched
torch.nn.functional.dropout that builds its mask from oracle/philox.py with the
site numbering of oracle/sdumc_oracle.py (call order of one forward).
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from oracle import philox, sdumc_oracle as O  # noqa: E402


def load_reference():
    spec = importlib.util.spec_from_file_location(
        "ref_model", os.path.join(REF, "toolkit/models/wengnet_mosei_mult_views_text_missing.py"))
    ref_model = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_model)
    spec = importlib.util.spec_from_file_location("ref_loss", os.path.join(REF, "toolkit/utils/loss.py"))
    ref_loss = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_loss)
    return ref_model, ref_loss


class PhiloxDropout:
    """Context manager: replace F.dropout by a Philox replay with a scripted
    site sequence (one entry per dropout call, in call order)."""

    def __init__(self, seed, call, sites=None, sample0=0):
        """sites = None: the k-th F.dropout call of the block uses site k (transformers_encoder convention)."""
        self.seed, self.call, self.sample0 = seed, call, sample0
        self.sites = None if sites is None else list(sites)
        self.i = 0

    def _drop(self, x, p=0.5, training=True, inplace=False):
        if not training:
            return x
        site = self.i if self.sites is None else self.sites[self.i]
        self.i += 1
        nsamp, width = x.shape[0], x.shape[-1]
        rows = x.numel() // (nsamp * width)
        m = philox.dropout_mask(nsamp, rows, width, p, self.seed, self.call, site, self.sample0)
        return x * torch.from_numpy(m).reshape(x.shape)

    def __enter__(self):
        self.orig = F.dropout
        F.dropout = self._drop
        torch.nn.functional.dropout = self._drop
        return self

    def __exit__(self, *a):
        F.dropout = self.orig
        torch.nn.functional.dropout = self.orig
        assert self.sites is None or self.i == len(self.sites), (self.i, len(self.sites))


def digest(t, key):
    """(sum, abs-sum, projection) of a tensor in float64; projection vector is
    seeded by the parameter name so that permutations are caught."""
    a = t.detach().double().reshape(-1).numpy()
    rs = np.random.RandomState(abs(hash_name(key)) % (2 ** 31))
    v = rs.standard_normal(a.shape[0])
    return np.array([a.sum(), np.abs(a).sum(), float(a @ v)], dtype=np.float64)


def hash_name(s):
    h = 2166136261
    for ch in s.encode():
        h = ((h ^ ch) * 16777619) & 0xFFFFFFFF
    return h


def np32(t):
    return t.detach().to(torch.float32).numpy()


def gen_blocks(ref_model, out):
    torch.manual_seed(11)
    B, T, Dm = 3, 11, 256
    x = torch.randn(B, T, Dm)
    q = torch.randn(B, 7, Dm)
    fra = ref_model.FRA2UTT_new(input_dim=Dm)
    ca = ref_model.Cross_Attention(input_dim=Dm)
    d = {"x": np32(x), "q": np32(q),
         "fra_ctx": np32(fra.attention_context_vector), "fra_w": np32(fra.input_proj.weight),
         "fra_b": np32(fra.input_proj.bias),
         "ca_wq": np32(ca.query_proj.weight), "ca_bq": np32(ca.query_proj.bias),
         "ca_wi": np32(ca.input_proj.weight), "ca_bi": np32(ca.input_proj.bias)}
    fra.eval(); ca.eval()
    o, a = fra(x); d["fra_eval_out"], d["fra_eval_att"] = np32(o), np32(a)
    o, a = ca(q, x); d["ca_eval_out"], d["ca_eval_att"] = np32(o), np32(a)
    fra.train(); ca.train()
    seed, call = 77, 5
    with PhiloxDropout(seed, call, [O.SITE_FRA_IN[1], O.SITE_FRA_OUT[1]]):
        o, a = fra(x)
    d["fra_train_out"], d["fra_train_att"] = np32(o), np32(a)
    with PhiloxDropout(seed, call, [O.SITE_CA_IN[2], O.SITE_CA_OUT[2]]):
        o, a = ca(q, x)
    d["ca_train_out"], d["ca_train_att"] = np32(o), np32(a)
    d["seed"], d["call"] = np.int64(seed), np.int64(call)
    # the D=1024 variant (C5 block-level shape family), eval only
    torch.manual_seed(12)
    ca2 = ref_model.Cross_Attention(input_dim=64)
    x2, q2 = torch.randn(2, 9, 64), torch.randn(2, 7, 64)
    ca2.eval()
    o, a = ca2(q2, x2)
    d.update({"ca64_x": np32(x2), "ca64_q": np32(q2), "ca64_wq": np32(ca2.query_proj.weight),
              "ca64_bq": np32(ca2.query_proj.bias), "ca64_wi": np32(ca2.input_proj.weight),
              "ca64_bi": np32(ca2.input_proj.bias), "ca64_out": np32(o), "ca64_att": np32(a)})
    np.savez_compressed(os.path.join(out, "blocks.npz"), **d)


def gen_blocks_1024(ref_model, out):
    """The two attention blocks at their constructors' DEFAULT width (input_dim = 1024; model :47, :71).  The 1024x1024
    weights are not stored: `torch.manual_seed(s); Block()` draws them, and a drop-in module constructed under the same
    seed draws the same ones (same layers in the same order); digests pin the stream."""
    d = {}
    g = torch.Generator().manual_seed(99)
    x = torch.randn(2, 9, 1024, generator=g)
    q = torch.randn(2, 7, 1024, generator=g) / 8
    torch.manual_seed(31)
    fra = ref_model.FRA2UTT_new()
    torch.manual_seed(32)
    ca = ref_model.Cross_Attention()
    d.update({"x": np32(x), "q": np32(q), "seed_fra": np.int64(31), "seed_ca": np.int64(32),
              "fra_w_digest": digest(fra.input_proj.weight, "fra.w"), "fra_ctx_digest": digest(fra.attention_context_vector, "fra.ctx"),
              "ca_wq_digest": digest(ca.query_proj.weight, "ca.wq"), "ca_wi_digest": digest(ca.input_proj.weight, "ca.wi")})
    fra.eval(); ca.eval()
    o, a = fra(x); d["fra_eval_out"], d["fra_eval_att"] = np32(o), np32(a)
    o, a = ca(q, x); d["ca_eval_out"], d["ca_eval_att"] = np32(o), np32(a)
    fra.train(); ca.train()
    seed, call = 55, 2
    with PhiloxDropout(seed, call, [O.SITE_FRA_IN[0], O.SITE_FRA_OUT[0]]):
        o, a = fra(x)
    d["fra_train_out"], d["fra_train_att"] = np32(o), np32(a)
    with PhiloxDropout(seed, call, [O.SITE_CA_IN[0], O.SITE_CA_OUT[0]]):
        o, a = ca(q, x)
    d["ca_train_out"], d["ca_train_att"] = np32(o), np32(a)
    d["seed"], d["call"] = np.int64(seed), np.int64(call)
    np.savez_compressed(os.path.join(out, "blocks1024.npz"), **d)


FWD_SITES = list(range(O.N_SITES))   # call order of one reference forward == site ids


def build_ref_net(ref_model, dims, pseed):
    args = types.SimpleNamespace(input_dims=dims)
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        net = ref_model.WengnetMOSEIMultViewsTextMissing(args)
    P = O.init_params(dims, seed=pseed)
    missing, unexpected = net.load_state_dict(P, strict=True), None
    return net, P


def gen_forward(ref_model, out):
    dims = (48, 32, 40, 32)
    B, T = 4, (13, 5, 9, 6)
    net, P = build_ref_net(ref_model, dims, pseed=3)
    audio, text, video, feat4, vals = O.synthetic_batch(B, T, dims, seed=1234)
    # ragged variant: zero the padded tail like the collater does (read_data.py:139-151)
    lens = torch.tensor([[13, 5, 9, 6], [7, 2, 9, 3], [4, 5, 3, 6], [10, 1, 6, 2]])
    for i, feat in enumerate((audio, text, video, feat4)):
        for b in range(B):
            feat[b, lens[b, i]:] = 0
    d = {"dims": np.array(dims), "T": np.array(T), "pseed": np.int64(3),
         "audio": np32(audio), "text": np32(text), "video": np32(video), "feat4": np32(feat4),
         "vals": np32(vals),
         "param_digest": np.stack([digest(P[k], k) for k in P])}
    names = ("vals", "fused", "rnc", "text_hidden", "cross_text")
    net.eval()
    with torch.no_grad():
        for s, tx in enumerate((text, feat4)):
            y, emb = net([audio, tx, video, bool(s)])
            for n, t in zip(names, [y] + list(emb)):
                d[f"eval{s}_{n}"] = np32(t)
    net.train()
    seed, step = 2024, 3
    with torch.no_grad():
        for s, tx in enumerate((text, feat4)):
            with PhiloxDropout(seed, 2 * step + s, FWD_SITES):
                y, emb = net([audio, tx, video, bool(s)])
            for n, t in zip(names, [y] + list(emb)):
                d[f"train{s}_{n}"] = np32(t)
    d["seed"], d["step"] = np.int64(seed), np.int64(step)
    np.savez_compressed(os.path.join(out, "forward.npz"), **d)


def gen_losses(ref_loss, out):
    torch.manual_seed(5)
    d = {}
    B = 6
    pred = torch.randn(B, 1, requires_grad=True)
    tgt = torch.rand(B) * 6 - 3
    l = ref_loss.MSELoss()(pred, tgt)
    l.backward()
    d.update(mse_pred=np32(pred), mse_tgt=np32(tgt), mse=np32(l), mse_dpred=np32(pred.grad))
    for tag, shp in (("2d", (B, 10)), ("3d", (B, 7, 8))):
        a = torch.randn(*shp, requires_grad=True)
        b = torch.randn(*shp, requires_grad=True)
        l = ref_loss.RMSELoss()(a, b)
        l.backward()
        d.update({f"rmse{tag}_a": np32(a), f"rmse{tag}_b": np32(b), f"rmse{tag}": np32(l),
                  f"rmse{tag}_da": np32(a.grad), f"rmse{tag}_db": np32(b.grad)})
    # RnC: distinct labels, and a tied-label case (ties + the -1e-4 slack, loss.py:303)
    for tag, labels in (("rnc", torch.rand(B, 1) * 6 - 3),
                        ("rnctie", torch.tensor([[1.0], [1.0], [-2.0], [0.5], [0.5], [1.00005]]))):
        f = torch.randn(B, 2, 16, requires_grad=True)
        l = ref_loss.RnCLoss()(f, labels)
        l.backward()
        d.update({f"{tag}_f": np32(f), f"{tag}_y": np32(labels), f"{tag}": np32(l), f"{tag}_df": np32(f.grad),
                  f"{tag}_mask": O.rnc_masks(labels.repeat(2, 1)).numpy()})
    np.savez_compressed(os.path.join(out, "losses.npz"), **d)


def gen_step(ref_model, ref_loss, out):
    dims = (48, 32, 40, 32)
    B, T = 4, (13, 5, 9, 6)
    net, P = build_ref_net(ref_model, dims, pseed=7)
    audio, text, video, feat4, vals = O.synthetic_batch(B, T, dims, seed=99)
    weights = O.DEFAULT_WEIGHTS
    reg, rmse, rnc = ref_loss.MSELoss(), ref_loss.RMSELoss(), ref_loss.RnCLoss()
    # exactly the call pattern of main_frame_val_text_missing.py:119-150
    opt = torch.optim.Adam(net.parameters(), lr=1e-4, weight_decay=1e-5)
    net.train()
    seed, step = 31337, 0
    opt.zero_grad()
    with PhiloxDropout(seed, 2 * step, FWD_SITES):
        y0, (z0, r0, t0, c0) = net([audio, text, video, False])
    with PhiloxDropout(seed, 2 * step + 1, FWD_SITES):
        y1, (z1, r1, t1, c1) = net([audio, feat4, video, True])
    nv = torch.stack((r0, r1), dim=1)
    terms = [reg(y0, vals), reg(y1, vals), rmse(t1, t0.detach()), rmse(c1, c0.detach()),
             rmse(z1, z0), rnc(nv, vals.unsqueeze(1))]
    loss = sum(w * t for w, t in zip(weights, terms))
    loss.backward()
    d = {"dims": np.array(dims), "T": np.array(T), "pseed": np.int64(7),
         "audio": np32(audio), "text": np32(text), "video": np32(video), "feat4": np32(feat4),
         "vals": np32(vals), "weights": np.array(weights), "seed": np.int64(seed), "step": np.int64(step),
         "loss": np32(loss), "terms": np.array([float(t.detach()) for t in terms], dtype=np.float64),
         "y0": np32(y0), "y1": np32(y1)}
    names, gdig, dead = [], [], []
    for k, p in net.named_parameters():
        names.append(k)
        if p.grad is None:
            dead.append(k)
            gdig.append(np.zeros(3))
        else:
            gdig.append(digest(p.grad, k))
            if p.numel() <= 2048:
                d["grad__" + k] = np32(p.grad)
    d["names"] = np.array(names)
    d["dead"] = np.array(dead)
    d["grad_digest"] = np.stack(gdig)
    before = {k: p.detach().clone() for k, p in net.named_parameters()}
    opt.step()
    d["delta_digest"] = np.stack([digest((p.detach() - before[k]) * 1e4, k) for k, p in net.named_parameters()])
    for k, p in net.named_parameters():
        if p.numel() <= 2048:
            d["delta__" + k] = np32((p.detach() - before[k]) * 1e4)
    sched = torch.optim.lr_scheduler.LambdaLR(
        torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=1.0),
        lr_lambda=lambda e: (e + 1) / 5 if e < 5 else 0.9 ** ((e + 1 - 5) // 10))
    lrs = []
    for e in range(40):
        lrs.append(sched.get_last_lr()[0])
        sched.optimizer.step()
        sched.step()
    d["lr_table"] = np.array(lrs)
    np.savez_compressed(os.path.join(out, "step.npz"), **d)


def gen_collate(out):
    # the collater imports cleanly only with stubs for absent third-party modules
    for name in ("prefetch_generator", "cv2", "torchaudio", "toolkit.utils.chatgpt"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            if name == "prefetch_generator":
                m.BackgroundGenerator = lambda it, **k: it
            if name == "toolkit.utils.chatgpt":   # absent from the reference repo
                m.get_translate_eng2chi = m.get_translate_chi2eng = None
            sys.modules[name] = m
    sys.path.insert(0, REF)
    try:
        from toolkit.utils.read_data import pad_to_maxlen_pre_modality_tensor_4
    finally:
        sys.path.remove(REF)
    rs = np.random.RandomState(8)
    lens = [(5, 2, 4, 3), (9, 1, 4, 1), (3, 6, 7, 2)]
    dims = (6, 4, 5, 4)
    raw = [[torch.from_numpy(rs.standard_normal((L[i], dims[i])).astype(np.float32)) for L in lens]
           for i in range(4)]
    a, t, v, f, pads = pad_to_maxlen_pre_modality_tensor_4(*[list(x) for x in raw])
    d = {"lens": np.array(lens), "dims": np.array(dims), "pads": np.array(pads)}
    for i, name in enumerate(("audio", "text", "video", "feat4")):
        for b in range(len(lens)):
            d[f"raw_{name}_{b}"] = raw[i][b].numpy()
    d["audios"], d["texts"] = torch.stack(a).numpy(), torch.stack(t).numpy()
    d["videos"], d["feat4s"] = torch.stack(v).numpy(), torch.stack(f).numpy()
    np.savez_compressed(os.path.join(out, "collate.npz"), **d)


def gen_init(ref_model, out):
    """state_dict of the reference right after `torch.manual_seed(0); Model(args)`: names, shapes,
    registration order and a digest per tensor (pins the drop-in module's initialisation stream)."""
    import contextlib
    import io
    dims = (48, 32, 40, 32)
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        net = ref_model.WengnetMOSEIMultViewsTextMissing(types.SimpleNamespace(input_dims=dims))
    names = [k for k, _ in net.named_parameters()]
    sd = net.state_dict()
    assert list(sd) == names
    np.savez_compressed(os.path.join(out, "init.npz"), dims=np.array(dims), names=np.array(names),
                        shapes=np.array([list(sd[k].shape) + [0] * (2 - sd[k].dim()) for k in names]),
                        digest=np.stack([digest(sd[k], k) for k in names]),
                        fc_att_weight=np32(sd["fc_att.weight"]), ctx0=np32(sd["fra2utt_0.attention_context_vector"]))


def load_reference_transformer():
    """The reference's transformers_encoder directory has no __init__.py and uses relative imports:
    import it as a synthetic package rooted at its own directory."""
    import importlib
    path = os.path.join(REF, "toolkit/models/modules/transformers_encoder")
    pkg = types.ModuleType("ref_te")
    pkg.__path__ = [path]
    sys.modules["ref_te"] = pkg
    return importlib.import_module("ref_te.transformer"), importlib.import_module("ref_te.multihead_attention")


def gen_transformer(out):
    ref_tr, ref_mha = load_reference_transformer()
    d = {}

    def put_params(tag, mod):
        for k, v in mod.state_dict().items():
            d[f"{tag}/P/{k}"] = np32(v)

    def put_grads(tag, mod, **inputs):
        for k, v in mod.named_parameters():
            d[f"{tag}/G/{k}"] = np32(v.grad)
            v.grad = None
        for k, v in inputs.items():
            d[f"{tag}/d{k}"] = np32(v.grad)
            v.grad = None

    def randomise(mod, gen):   # biases / LayerNorm affine are 0 / 1 at init: give them values
        with torch.no_grad():
            for k, v in mod.named_parameters():
                if k.endswith("bias") or "layer_norm" in k:
                    v.add_(0.2 * torch.randn(v.shape, generator=gen))

    g = torch.Generator().manual_seed(2024)
    rn = lambda *s: torch.randn(*s, generator=g)

    # A: self-attention, eval
    torch.manual_seed(1)
    m = ref_mha.MultiheadAttention(32, 4); randomise(m, g); m.eval()
    x = rn(6, 3, 32)
    o, w = m(x, x, x)
    put_params("mha_self", m)
    d.update({"mha_self/x": np32(x), "mha_self/out": np32(o), "mha_self/weights": np32(w)})

    # B: cross attention (Tk = 9: ragged quads), future mask, attention dropout, gradients
    torch.manual_seed(2)
    m = ref_mha.MultiheadAttention(32, 4, attn_dropout=0.25); randomise(m, g); m.train()
    q, k, v = (rn(5, 2, 32).requires_grad_(), rn(9, 2, 32).requires_grad_(), rn(9, 2, 32).requires_grad_())
    mask = ref_tr.buffered_future_mask(q, k)
    R = rn(5, 2, 32)
    with PhiloxDropout(91, 3):
        o, w = m(q, k, v, attn_mask=mask)
    (o * R).sum().backward()
    put_params("mha_cross", m)
    d.update({"mha_cross/q": np32(q), "mha_cross/k": np32(k), "mha_cross/v": np32(v), "mha_cross/mask": np32(mask),
              "mha_cross/R": np32(R), "mha_cross/out": np32(o), "mha_cross/weights": np32(w),
              "mha_cross/seed_call": np.array([91, 3])})
    put_grads("mha_cross", m, q=q, k=k, v=v)

    # C: head_dim 15 (unaligned head slices), key is value but not query
    torch.manual_seed(3)
    m = ref_mha.MultiheadAttention(30, 2); randomise(m, g); m.eval()
    q, kv = rn(7, 2, 30).requires_grad_(), rn(4, 2, 30).requires_grad_()
    R = rn(7, 2, 30)
    o, w = m(q, kv, kv)
    (o * R).sum().backward()
    put_params("mha_odd", m)
    d.update({"mha_odd/q": np32(q), "mha_odd/kv": np32(kv), "mha_odd/R": np32(R), "mha_odd/out": np32(o),
              "mha_odd/weights": np32(w)})
    put_grads("mha_odd", m, q=q, kv=kv)

    # D: LayerNorm(300) (the width of the reference's own __main__ example, transformer.py:206-209)
    ln = ref_tr.LayerNorm(300)
    with torch.no_grad():
        ln.weight.copy_(1 + 0.3 * rn(300)); ln.bias.copy_(0.3 * rn(300))
    x = (2.0 * rn(5, 3, 300) + 0.7).requires_grad_()
    R = rn(5, 3, 300)
    y = ln(x)
    (y * R).sum().backward()
    put_params("ln", ln)
    d.update({"ln/x": np32(x), "ln/R": np32(R), "ln/y": np32(y)})
    put_grads("ln", ln, x=x)

    # E/F: one encoder layer, self and cross, train mode with all three dropouts and the future mask
    torch.manual_seed(4)
    lay = ref_tr.TransformerEncoderLayer(32, num_heads=4, attn_dropout=0.1, relu_dropout=0.2, res_dropout=0.3,
                                         attn_mask=True)
    randomise(lay, g); lay.train()
    put_params("layer", lay)
    x = rn(8, 2, 32).requires_grad_()
    R = rn(8, 2, 32)
    with PhiloxDropout(17, 6):
        y = lay(x)
    (y * R).sum().backward()
    d.update({"layer/x": np32(x), "layer/R": np32(R), "layer/self_out": np32(y), "layer/seed_call": np.array([17, 6])})
    put_grads("layer/self", lay, x=x)
    xk, xv = rn(12, 2, 32).requires_grad_(), rn(12, 2, 32).requires_grad_()
    with PhiloxDropout(17, 7):
        y = lay(x, xk, xv)
    (y * R).sum().backward()
    d.update({"layer/xk": np32(xk), "layer/xv": np32(xv), "layer/cross_out": np32(y)})
    put_grads("layer/cross", lay, x=x, xk=xk, xv=xv)
    lay.eval()
    d["layer/self_eval_out"] = np32(lay(x))

    # G/H: two-layer encoder with sinusoidal positions (some tokens have first channel == 0 -> padding row)
    torch.manual_seed(5)
    enc = ref_tr.TransformerEncoder(32, 4, 2, attn_dropout=0.1, relu_dropout=0.1, res_dropout=0.2,
                                    embed_dropout=0.15, attn_mask=True, position_embedding=True)
    randomise(enc, g)
    put_params("enc", enc)
    x = rn(8, 3, 32)
    x[2, 1, 0] = 0.0; x[5, 0, 0] = 0.0; x[7, 2, 0] = 0.0
    x.requires_grad_()
    R = rn(8, 3, 32)
    enc.eval()
    d["enc/self_eval_out"] = np32(enc(x))
    enc.train()
    with PhiloxDropout(23, 1):
        y = enc(x)
    (y * R).sum().backward()
    d.update({"enc/x": np32(x), "enc/R": np32(R), "enc/self_out": np32(y), "enc/seed_call": np.array([23, 1])})
    put_grads("enc/self", enc, x=x)
    xk = rn(12, 3, 32); xk[0, 0, 0] = 0.0; xk[11, 2, 0] = 0.0
    xv = rn(12, 3, 32); xv[3, 1, 0] = 0.0
    xk.requires_grad_(); xv.requires_grad_()
    with PhiloxDropout(23, 2):
        y = enc(x, xk, xv)
    (y * R).sum().backward()
    d.update({"enc/xk": np32(xk), "enc/xv": np32(xv), "enc/cross_out": np32(y)})
    put_grads("enc/cross", enc, x=x, xk=xk, xv=xv)
    # no-position / no-mask variant, eval
    torch.manual_seed(6)
    enc2 = ref_tr.TransformerEncoder(32, 4, 1)
    randomise(enc2, g); enc2.eval()
    put_params("enc_plain", enc2)
    x2 = rn(5, 2, 32)
    d.update({"enc_plain/x": np32(x2), "enc_plain/out": np32(enc2(x2))})
    np.savez_compressed(os.path.join(out, "transformer.npz"), **d)


def gen_transformer_kv(out):
    """add_bias_kv / add_zero_attn of the generic MultiheadAttention (multihead_attention.py:28-38, :86-104): outputs, weights
    and every gradient of the REAL module.  A fixture of its own (transformer_kv.npz) so that transformer.npz stays as it is."""
    ref_tr, ref_mha = load_reference_transformer()
    d = {}
    g = torch.Generator().manual_seed(777)
    rn = lambda *s: torch.randn(*s, generator=g)

    def put(tag, mod, tensors, grads=None):
        for k, v in mod.state_dict().items():
            d[f"{tag}/P/{k}"] = np32(v)
        for k, v in tensors.items():
            d[f"{tag}/{k}"] = np32(v)
        if grads is not None:
            for k, v in mod.named_parameters():
                d[f"{tag}/G/{k}"] = np32(v.grad)
            for k, v in grads.items():
                d[f"{tag}/d{k}"] = np32(v.grad)

    def randomise(mod):
        with torch.no_grad():
            for k, v in mod.named_parameters():
                if k.endswith("bias"):
                    v.add_(0.2 * torch.randn(v.shape, generator=g))

    # both extras, cross attention with the future mask, attention dropout, gradients
    torch.manual_seed(11)
    m = ref_mha.MultiheadAttention(32, 4, attn_dropout=0.25, add_bias_kv=True, add_zero_attn=True); randomise(m); m.train()
    q, k, v = rn(5, 2, 32).requires_grad_(), rn(9, 2, 32).requires_grad_(), rn(9, 2, 32).requires_grad_()
    mask = ref_tr.buffered_future_mask(q, k)
    R = rn(5, 2, 32)
    with PhiloxDropout(23, 4):
        o, w = m(q, k, v, attn_mask=mask)
    (o * R).sum().backward()
    put("both", m, {"q": q, "k": k, "v": v, "mask": mask, "R": R, "out": o, "weights": w}, {"q": q, "k": k, "v": v})
    d["both/seed_call"] = np.array([23, 4])

    # add_zero_attn alone, self-attention, eval, gradients (T = 7 -> source length 8)
    torch.manual_seed(12)
    m = ref_mha.MultiheadAttention(32, 4, add_zero_attn=True); randomise(m); m.eval()
    x = rn(7, 3, 32).requires_grad_()
    R = rn(7, 3, 32)
    o, w = m(x, x, x)
    (o * R).sum().backward()
    put("zero", m, {"x": x, "R": R, "out": o, "weights": w}, {"x": x})

    # add_bias_kv alone, key is value, head_dim 15, no mask, gradients
    torch.manual_seed(13)
    m = ref_mha.MultiheadAttention(30, 2, add_bias_kv=True); randomise(m); m.eval()
    q, kv = rn(6, 2, 30).requires_grad_(), rn(4, 2, 30).requires_grad_()
    R = rn(6, 2, 30)
    o, w = m(q, kv, kv)
    (o * R).sum().backward()
    put("bias", m, {"q": q, "kv": kv, "R": R, "out": o, "weights": w}, {"q": q, "kv": kv})
    np.savez_compressed(os.path.join(out, "transformer_kv.npz"), **d)


def main():
    torch.set_num_threads(4)
    ref_model, ref_loss = load_reference()
    gen_init(ref_model, HERE)
    gen_blocks(ref_model, HERE)
    gen_blocks_1024(ref_model, HERE)
    gen_forward(ref_model, HERE)
    gen_losses(ref_loss, HERE)
    gen_step(ref_model, ref_loss, HERE)
    gen_collate(HERE)
    gen_transformer(HERE)
    gen_transformer_kv(HERE)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
