"""Drop-in replacement of the reference model module
(toolkit/models/wengnet_mosei_mult_views_text_missing.py:186-370 and the get_models wrapper,
toolkit/models/__init__.py:29-70) backed by libsdumc_hip.so.

Same constructor signature, same `forward(batch)` contract, same `state_dict()` names and shapes
(4 268 884 parameters; the published 49 MB checkpoint loads by name), same initialisation stream
(torch.manual_seed(s) followed by the constructor yields the reference's initial weights bit for bit),
`model.train()/.eval()` toggles every dropout.  The arithmetic runs in hand-written HIP kernels; there
is no PyTorch/CPU fallback: calling forward on CPU tensors raises.

Parameters are views of ONE flat fp32 buffer (live tensors first) so that the gradient bucket is a
single contiguous all-reduce payload and Adam is one kernel.
"""
import torch
import torch.nn as nn

from . import engine
from ._lib import SdumcError

GENERAL_DIM = 256


class _NetFn(torch.autograd.Function):
    """One reference forward call (one stream) = sdumc_net_forward; backward = sdumc_net_backward."""

    @staticmethod
    def forward(ctx, module, audio, text, video, call_index, lengths, *live_params):
        rng = engine.RngState(module.seed, audio.device, call=call_index)
        call = engine.NetCall(module._flat, audio, [text], video, True, rng, sample0=module.sample0,
                              p_mlp=module.dropout_p, lengths=lengths, planes=False)     # (fresh features every call: a plane copy would be split for one use)
        outs = call.forward()
        ctx.call, ctx.module = call, module
        return tuple(outs)

    @staticmethod
    def backward(ctx, d_vals, d_fused, d_rnc, d_text_hidden, d_cross_text):
        def c(t):
            return None if t is None else t.contiguous()
        lay = ctx.module._layout
        grads = ctx.call.backward(c(d_vals), c(d_fused), c(d_rnc), c(d_text_hidden), c(d_cross_text))
        out = []
        for name in ctx.module._live_names:
            off, shape, _ = lay.entries[name]
            n = 1
            for s in shape:
                n *= s
            out.append(grads[off:off + n].view(shape))
        ctx.call = None
        return (None, None, None, None, None, None) + tuple(out)


class WengnetMOSEIMultViewsTextMissing(nn.Module):
    def __init__(self, args, output_dim1=1, output_dim2=1, layers='256,128', dropout=0.3):
        super().__init__()
        if [int(x) for x in layers.split(',')] != [256, 128] or output_dim1 != 1 or output_dim2 != 1:
            raise SdumcError("the HIP path implements the shipped configuration layers='256,128', output dims 1 "
                             "(model :187); other widths are not built")
        dims = tuple(int(d) for d in args.input_dims[:3])
        self.input_dims = dims
        self.dropout_p = float(dropout)
        self.sample0 = 0                    # data-parallel shard offset for the Philox sample index
        self._layout = engine.ParamLayout.get(*dims)
        self._flat = torch.zeros(self._layout.total)
        self._live_names = self._layout.live_names()
        init = _reference_order_init(dims)
        views = self._layout.views(self._flat)
        self._pnames = []
        for name in init:                   # registration order == the reference's named_parameters() order
            views[name].copy_(init[name])
            self._register(name, nn.Parameter(views[name], requires_grad=True))
        # Philox key: drawn from torch's global RNG so that torch.manual_seed controls the dropout stream too
        self.seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        self._calls = 0

    # parameters carry the reference's dotted names ("audio_mlp.0.weight"): keep them in nested holders
    def _register(self, dotted, param):
        parts = dotted.split('.')
        mod = self
        for p in parts[:-1]:
            if p not in mod._modules:
                mod.add_module(p, nn.Module())
            mod = mod._modules[p]
        mod.register_parameter(parts[-1], param)
        self._pnames.append(dotted)

    def _get(self, dotted):
        mod = self
        parts = dotted.split('.')
        for p in parts[:-1]:
            mod = mod._modules[p]
        return mod._parameters[parts[-1]]

    def _reflatten(self):
        """Make every parameter a view of one flat buffer on the parameters' current device."""
        first = self._get(self._pnames[0])
        flat = torch.zeros(self._layout.total, device=first.device, dtype=torch.float32)
        views = self._layout.views(flat)
        with torch.no_grad():
            for name in self._pnames:
                p = self._get(name)
                views[name].copy_(p.data.to(torch.float32))
                p.data = views[name]
        self._flat = flat

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._reflatten()
        return out

    def _check_flat(self):
        base = self._flat.data_ptr()
        for name in (self._pnames[0], self._pnames[-1]):
            off = self._layout.entries[name][0]
            if self._get(name).data_ptr() != base + 4 * off:
                self._reflatten()
                return

    def forward(self, batch, lengths=None):
        """`lengths` is an extension (default None = the reference's behaviour): (audio, text, video) valid frame counts
        per sample; padded frames are then masked out of the attention poolings instead of joining the softmax."""
        audio, text, video = batch[0], batch[1], batch[2]       # batch[-1] = missing_flag: read and ignored (model :278)
        self._check_flat()
        if not audio.is_cuda:
            raise SdumcError("sdumc_amd runs on the GPU only (no CPU fallback): move the model and the batch to cuda")
        audio, text, video = (t.contiguous().float() for t in (audio, text, video))
        if self.training:
            call_index = self._calls
            self._calls += 1
            if torch.is_grad_enabled():
                live = [self._get(n) for n in self._live_names]
                vals, fused, rnc, th, ct = _NetFn.apply(self, audio, text, video, call_index, lengths, *live)
            else:
                rng = engine.RngState(self.seed, audio.device, call=call_index)
                vals, fused, rnc, th, ct = engine.NetCall(self._flat, audio, [text], video, True, rng,
                                                          sample0=self.sample0, p_mlp=self.dropout_p,
                                                          lengths=lengths, planes=False).forward()
        else:
            vals, fused, rnc, th, ct = engine.NetCall(self._flat, audio, [text], video, False, None,
                                                      p_mlp=self.dropout_p, lengths=lengths, planes=False).forward()
        return vals, [fused, rnc, th, ct]


def _reference_order_init(dims):
    """Initial values drawn in the reference constructor's order (model :193-260) from torch's global RNG,
    so `torch.manual_seed(s); Model(args)` reproduces the reference's initial state_dict."""
    out = {}

    def lin(name, i, o):
        l = nn.Linear(i, o)
        out[name + ".weight"], out[name + ".bias"] = l.weight.detach().clone(), l.bias.detach().clone()

    D, H = GENERAL_DIM, 128
    for m in range(3):
        lin(f"frame_dim_reshape_{m}", dims[m], D)
    for pre, dim, lat in (("missing_text_imagination_mlp", D, 128), ("missing_cross_text_query_imagination_mlp", 128, 64)):
        lin(pre + ".transition.0", dim * 3, dim)
        lin(pre + ".transition.2", dim, dim)
        lin(pre + ".encoder_0.0", dim, lat)
        lin(pre + ".decoder_0.0", lat, dim)
    for m in range(3):
        ctx = torch.empty(1, D)
        nn.init.xavier_normal_(ctx)
        out[f"fra2utt_{m}.attention_context_vector"] = ctx
        lin(f"fra2utt_{m}.input_proj", D, D)
    for n in ("audio", "text", "video"):
        lin(f"{n}_mlp.0", D, D)
        lin(f"{n}_mlp.3", D, D)
    lin("attention_mlp.0", 3 * D, D)
    lin("attention_mlp.3", D, D)
    lin("fc_att", D, 3)
    for n in ("fused", "at", "tv", "av", "audio", "text", "video"):
        lin(f"cross_{n}_query_mlp.0", D, D)
    for m in range(3):
        lin(f"cross_att_fra2utt_{m}.query_proj", D, D)
        lin(f"cross_att_fra2utt_{m}.input_proj", D, D)
    for n in ("audio", "text", "video"):
        lin(f"cross_{n}_mlp.0", D, D)
        lin(f"cross_{n}_mlp.3", D, H)
    lin("cross_attention_mlp.0", 7 * H, D)
    lin("cross_attention_mlp.3", D, H)
    lin("cross_fc_att", H, 7)
    lin("fc_out_e", H, 1)
    lin("fc_out_v", H, 1)
    lin("fc_out_ev", 1, 1)
    lin("orgin_linear_change.0", H, 64)
    lin("orgin_linear_change.2", 64, 64)
    out["prelu.weight"] = torch.full((6,), 0.25)
    out["layer_normali.weight"] = torch.ones(D)
    out["layer_normali.bias"] = torch.zeros(D)
    return out


class get_models(nn.Module):
    """toolkit/models/__init__.py:29-70: wraps the network as `.model` (checkpoint keys `model.<name>`)."""

    MODEL_MAP = {'wengnet_mosei_mult_views_text_missing': WengnetMOSEIMultViewsTextMissing}

    def __init__(self, args):
        super().__init__()
        args.dim = 1024
        if args.model not in self.MODEL_MAP:
            raise SdumcError(f"model '{args.model}' is not part of the SDUMC hot path; only "
                             f"{list(self.MODEL_MAP)} is built")
        self.model = self.MODEL_MAP[args.model](args)

    def forward(self, batch):
        return self.model(batch)
