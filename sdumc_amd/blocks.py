"""Stand-alone drop-in mirrors of the two attention blocks of the reference model file
(`toolkit/models/wengnet_mosei_mult_views_text_missing.py`):

  FRA2UTT_new      (model :46-68)   frame -> utterance pooling with one learned context vector
  Cross_Attention  (model :70-95)   the UMCA block: 7 (or any <= 8) view queries pool the frames of one modality

Same constructor signatures, parameter names/shapes (`attention_context_vector[1, D]`, `input_proj`,
`query_proj`, the `dropout_output` sub-module), inputs and return values; forward and backward run on the
HIP kernels the full network uses (`sdumc_gemm_f32` with the input dropout and tanh fused, `sdumc_attnpool_fwd/bwd`,
`sdumc_drop_add`).  `input_dim` may be 256 (what the model instantiates: `general_dim = 256` is hard-coded, model :191),
512, 768 or 1024 (the constructors' default); other widths raise NotImplementedError.  CPU tensors raise.

Dropout: the block's two `self.dropout_output(...)` calls draw Philox masks keyed by (seed, call, site) from the
module-level `dropout_stream` (sites in call order, exactly like `sdumc_amd.transformers_encoder`).
"""
import ctypes as C

import torch
from torch import nn

from . import _lib, ops
from .transformers_encoder import DropoutStream, _LinearFn, _c, _dev

DIMS = (256, 512, 768, 1024)

dropout_stream = DropoutStream()


def manual_seed(seed, call=0, site=0):
    dropout_stream.reset(seed, call)
    dropout_stream.site = int(site)


class _PoolFn(torch.autograd.Function):
    """(x [B,T,256], q' [B or 1, nq, 256], W_in, b_in) -> (out [B,nq,256], attn [B,T,nq])."""

    @staticmethod
    def forward(ctx, x, q, w, b, scale, x_drop, out_drop, q_shared):
        x, q, w, b = _c(x), _c(q), _c(w), _c(b)
        _dev(x, q, w, b)
        B, T, D = x.shape
        nq = q.shape[1]
        keys = ops.gemm(ops.NT, x, w, B * T, D, D, bias=b, act=ops.ACT_TANH, a_drop=x_drop, splitk=0).view(B, T, D)
        out, attn, pooled, desc = _pool_with_scale(x, keys, q, nq, B, q_shared, x_drop, out_drop, scale)
        ctx.saved = (x, keys, q, w, desc, attn, pooled, out)
        ctx.x_drop, ctx.q_shared = x_drop, q_shared
        ctx.mark_non_differentiable(attn)
        return out, attn

    @staticmethod
    def backward(ctx, dout, _dattn):
        x, keys, q, w, desc, attn, pooled, out = ctx.saved
        B, T, D = x.shape
        dz, dxd, dq = ops.attnpool_bwd(desc, _c(dout), (x, keys, q))
        dw = ops.gemm(ops.TN, dz.view(-1, D), x, D, D, B * T, b_drop=ctx.x_drop, splitk=0)
        db = ops.colsum(dz.view(-1, D))
        ops.gemm(ops.NN, dz.view(-1, D), w, B * T, D, D, C_out=dxd.view(-1, D), accumulate=True, splitk=0)
        dx = _mask_apply(dxd, ctx.x_drop, B, T)
        if ctx.q_shared:
            dq = ops.colsum(dq.view(B, -1)).view(1, -1, D)
        ctx.saved = None
        return dx, dq, dw, db, None, None, None, None


def _pool_with_scale(x, keys, q, nq, B, q_shared, x_drop, out_drop, scale):
    V, T, D = keys.shape
    dev = keys.device
    attn = torch.empty(V, T, nq, device=dev)
    pooled = torch.empty(V, nq, D, device=dev)
    out = torch.empty(V, nq, D, device=dev)
    a = ops.attnpool_desc(x, keys, q, V, T, nq, B, 0 if q_shared else nq * D, x_drop, out_drop, attn, pooled, out, scale, D)
    need = _lib.lib.sdumc_attnpool_fwd_workspace_bytes_dim(V, T, nq, D)
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    a.workspace, a.workspace_bytes = _lib.ptr(ws), need
    _lib.check(_lib.lib.sdumc_attnpool_fwd(C.byref(a), _lib.current_stream()), "sdumc_attnpool_fwd")
    return out, attn, pooled, a


def _mask_apply(g, drop, B, T):
    """g * mask of the input dropout (identity in eval mode): the backward of `dropout_output(input_tensor)`."""
    if drop is None or not drop.enabled:
        return g
    return ops.drop_add(g, None, drop, 1.0)


def _check_dim(input_dim):
    if input_dim not in DIMS:
        raise NotImplementedError(f"the HIP attention-pooling kernels are built for input_dim in {DIMS} "
                                  f"(multiples of 256, the model's general_dim); got {input_dim}")


class FRA2UTT_new(nn.Module):
    """model :46-68.  forward(input_tensor [B,T,D]) -> (output [B,D], attention [B,T,1])."""

    def __init__(self, input_dim=1024, atsize=1024, softmax_scale=0.3):
        super().__init__()
        _check_dim(input_dim)
        self.atsize = atsize
        self.softmax_scale = softmax_scale
        self.attention_context_vector = nn.Parameter(torch.empty(1, input_dim))
        nn.init.xavier_normal_(self.attention_context_vector)
        self.input_proj = nn.Linear(input_dim, input_dim)
        self.dropout_output = nn.Dropout(0.5)

    def forward(self, input_tensor):
        B, T, D = input_tensor.shape
        p = self.dropout_output.p
        x_drop = dropout_stream.draw(p, B, T, D, self.training)
        out_drop = dropout_stream.draw(p, B, 1, D, self.training)
        q = self.attention_context_vector.view(1, 1, D)
        out, attn = _PoolFn.apply(input_tensor, q, self.input_proj.weight, self.input_proj.bias, self.softmax_scale,
                                  x_drop, out_drop, True)
        return out.view(B, D), attn


class Cross_Attention(nn.Module):
    """model :70-95 (the UMCA block).  forward(query_tensor [B,nq,D], input_tensor [B,T,D]) ->
    (output [B,nq,D], attention [B,T,nq]); nq <= 8."""

    def __init__(self, input_dim=1024, atsize=1024, softmax_scale=0.3):
        super().__init__()
        _check_dim(input_dim)
        self.atsize = atsize
        self.softmax_scale = softmax_scale
        self.query_proj = nn.Linear(input_dim, input_dim)
        self.input_proj = nn.Linear(input_dim, input_dim)
        self.dropout_output = nn.Dropout(0.5)

    def forward(self, query_tensor, input_tensor):
        B, T, D = input_tensor.shape
        nq = query_tensor.shape[1]
        if nq > 8:
            raise NotImplementedError("at most 8 queries per block (the model uses 7, :332)")
        p = self.dropout_output.p
        x_drop = dropout_stream.draw(p, B, T, D, self.training)
        out_drop = dropout_stream.draw(p, B, nq, D, self.training)
        qp = _LinearFn.apply(query_tensor, self.query_proj.weight, self.query_proj.bias, False, None)
        return _PoolFn.apply(input_tensor, qp, self.input_proj.weight, self.input_proj.bias, self.softmax_scale,
                             x_drop, out_drop, False)
