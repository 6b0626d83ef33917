"""Drop-in mirror of the reference's ``toolkit/models/modules/transformers_encoder`` package
(multihead_attention.py, transformer.py, position_embedding.py) on the HIP kernels of
``csrc/transformer.hip`` + ``csrc/gemm_f32.hip``.

Same class names, constructor arguments, parameter names/shapes (so a reference ``state_dict`` loads),
Time x Batch x Channel tensors, return values and error behaviour.  The modules own ordinary
``nn.Parameter`` s; every forward and backward runs on the GPU through the C ABI (``sdumc_mha_forward`` /
``sdumc_mha_backward``, ``sdumc_layernorm_*``, ``sdumc_gemm_f32``, ``sdumc_drop_add``); autograd only
chains the four operator-level ``Function`` s below.  There is no CPU / PyTorch fallback: CPU tensors raise.

Dropout: the reference calls ``F.dropout`` (torch's bernoulli stream).  Here every dropout call draws a
Philox mask keyed by (seed, call, site) from the module-level :class:`DropoutStream`; ``site`` numbers the
``F.dropout`` calls of the reference in call order (calls with p = 0 count too), which is what the golden
fixtures replay (tests/golden/make_goldens.py: gen_transformer).

``add_bias_kv`` / ``add_zero_attn`` (multihead_attention.py:28-38, :86-104; default off and never enabled by
transformer.py) are built into the HIP path (one extra key/value row each).  Not implemented (fail loudly):
gradients flowing into the returned attention weights.
"""
import math

import torch
from torch import nn

from . import _lib, ops
from ._lib import make_dropout


class DropoutStream:
    """(seed, call, site) source of the Philox dropout masks of this module family."""

    def __init__(self, seed=0, call=0):
        self.seed, self.call, self.site = int(seed), int(call), 0

    def reset(self, seed=None, call=None):
        if seed is not None:
            self.seed = int(seed)
        if call is not None:
            self.call = int(call)
        self.site = 0

    def draw(self, p, samples, rows, width, training):
        """Descriptor of the next F.dropout call (None = identity).  Mirrors F.dropout(x, p, training)."""
        if not training:
            return None
        if not 0.0 <= p < 1.0:
            raise ValueError(f"dropout probability has to be in [0, 1), got {p}")
        site = self.site
        self.site += 1
        if p == 0.0:
            return None
        return make_dropout(True, site, p, rows, width, samples, 0, self.call, self.seed)


dropout_stream = DropoutStream()

_BF16 = False


def set_bf16(on=True):
    """bf16-operand mode for every product of this module family (fp32 accumulate, fp32 storage): Linear / in_proj /
    out_proj forward and backward, q.k^T and P.v.  Off by default (exact fp32).  Shapes whose extents are not multiples
    of 4 stay on the fp32 path."""
    global _BF16
    _BF16 = bool(on)


def manual_seed(seed, call=0):
    """Re-key the dropout stream (and restart the site numbering)."""
    dropout_stream.reset(seed, call)


def _dev(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise _lib.SdumcError("sdumc_amd.transformers_encoder runs on the GPU only (got a CPU tensor); "
                                  "there is no CPU fallback")


def _c(t):
    if t is None:
        return None
    if t.dtype != torch.float32:
        raise TypeError(f"float32 tensors only, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


# ---------------------------------------------------------------------------------------------------
# operator-level autograd Functions
# ---------------------------------------------------------------------------------------------------
class _LinearFn(torch.autograd.Function):
    """y = drop(act(x W^T + b)) over the last axis: one NT GEMM with fused epilogue; backward = one NN + one TN
    GEMM (bias gradient fused into the TN staging)."""

    @staticmethod
    def forward(ctx, x, w, b, relu, drop):
        x, w, b = _c(x), _c(w), _c(b)
        _dev(x, w, b)
        N, K = w.shape
        M = x.numel() // K
        y = torch.empty(*x.shape[:-1], N, device=x.device)
        bf16 = _BF16 and N % 4 == 0 and K % 4 == 0 and M % 4 == 0
        ops.gemm(ops.NT, x, w, M, N, K, bias=b, C_out=y, act=ops.ACT_RELU if relu else ops.ACT_NONE, c_drop=drop,
                 splitk=0, bf16=bf16)
        ctx.save_for_backward(x, w, y if relu else None)
        ctx.has_bias, ctx.relu, ctx.scale = b is not None, relu, (drop.scale if drop is not None else 1.0)
        ctx.bf16 = bf16
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        dy = _c(dy)
        N, K = w.shape
        M = x.numel() // K
        if ctx.relu:   # dz = dy * [y > 0] / (1 - p) from the saved post-dropout output
            dz = torch.empty_like(dy)
            _lib.check(_lib.lib.sdumc_relu_drop_bwd(dy.data_ptr(), y.data_ptr(), ctx.scale, dz.data_ptr(), dy.numel(),
                                                    _lib.current_stream()), "sdumc_relu_drop_bwd")
            dy = dz
        dx = torch.empty_like(x)
        ops.gemm(ops.NN, dy, w, M, K, N, C_out=dx, splitk=0, bf16=ctx.bf16)
        dw = torch.empty_like(w)
        db = torch.empty(N, device=x.device) if ctx.has_bias else None
        ops.gemm(ops.TN, dy, x, N, K, M, C_out=dw, colsum_a=db, splitk=0, bf16=ctx.bf16)
        return dx, dw, db, None, None


class _LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        x, gamma, beta = _c(x), _c(gamma), _c(beta)
        _dev(x, gamma, beta)
        y, mean, rstd = ops.layernorm_fwd(x, gamma, beta, eps)
        ctx.save_for_backward(x, gamma, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mean, rstd = ctx.saved_tensors
        dx, dg, db = ops.layernorm_bwd(_c(dy), x, gamma, mean, rstd)
        return dx, dg, db, None


class _NormResidualFn(torch.autograd.Function):
    """(LayerNorm(x), x): the normalised branch input and the residual passthrough of a pre-LN block as ONE node, so
    that the two gradients meeting at x are summed by the LayerNorm backward kernel (accumulate_dx) instead of a
    separate element-wise add."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        x, gamma, beta = _c(x), _c(gamma), _c(beta)
        _dev(x, gamma, beta)
        y, mean, rstd = ops.layernorm_fwd(x, gamma, beta, eps)
        ctx.save_for_backward(x, gamma, mean, rstd)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dres):
        x, gamma, mean, rstd = ctx.saved_tensors
        dx, dg, db = ops.layernorm_bwd(_c(dy), x, gamma, mean, rstd, dx_add=_c(dres))
        return dx, dg, db, None


class _DropAddFn(torch.autograd.Function):
    """y = drop(alpha * x + pos) + residual."""

    @staticmethod
    def forward(ctx, x, residual, drop, alpha, pos_table):
        x, residual = _c(x), _c(residual)
        _dev(x, residual)
        y = ops.drop_add(x, residual, drop, alpha, pos_table, x if pos_table is not None else None)
        ctx.drop, ctx.alpha, ctx.has_res = drop, alpha, residual is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        if ctx.drop is None and ctx.alpha == 1.0:
            dx = dy
        else:
            dx = ops.drop_add(dy, None, ctx.drop, ctx.alpha)
        return dx, (dy if ctx.has_res else None), None, None, None


class _MhaFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, query, key, value, w_in, b_in, w_out, b_out, attn_mask, heads, drop, bias_k=None, bias_v=None,
                add_zero_attn=False):
        # aliasing is decided the way the reference decides it (data_ptr() equality, multihead_attention.py:61-62)
        def alias(a, b):
            return a.data_ptr() == b.data_ptr() and a.shape == b.shape and a.stride() == b.stride()
        same_kq, same_vk, same_vq = alias(key, query), alias(value, key), alias(value, query)
        query = _c(query)
        key = query if same_kq else _c(key)
        value = query if same_vq else (key if same_vk else _c(value))
        w_in, b_in, w_out, b_out, attn_mask = _c(w_in), _c(b_in), _c(w_out), _c(b_out), _c(attn_mask)
        bk = _c(bias_k.reshape(-1)) if bias_k is not None else None       # [1, 1, E] parameters -> [E]
        bv = _c(bias_v.reshape(-1)) if bias_v is not None else None
        _dev(query, key, value, w_in, b_in, w_out, b_out, attn_mask, bk, bv)
        out, weights, saved = ops.mha_forward(query, key, value, w_in, b_in, w_out, b_out, heads, attn_mask, drop,
                                              bf16=_BF16, bias_k=bk, bias_v=bv, add_zero_attn=add_zero_attn)
        ctx.saved = saved
        ctx.bias_shape = tuple(bias_k.shape) if bias_k is not None else None
        ctx.mark_non_differentiable(weights)
        return out, weights

    @staticmethod
    def backward(ctx, dout, _dweights):
        res = ops.mha_backward(ctx.saved, _c(dout))
        ctx.saved = None
        dq, dk, dv, dw_in, db_in, dw_out, db_out = res[:7]
        dbk = res[7].view(ctx.bias_shape) if ctx.bias_shape is not None else None
        dbv = res[8].view(ctx.bias_shape) if ctx.bias_shape is not None else None
        return dq, dk, dv, dw_in, db_in, dw_out, db_out, None, None, None, dbk, dbv, None


def _linear(x, weight, bias, relu=False, drop=None):
    return _LinearFn.apply(x, weight, bias, relu, drop)


# ---------------------------------------------------------------------------------------------------
# multihead_attention.py
# ---------------------------------------------------------------------------------------------------
class MultiheadAttention(nn.Module):
    """Multi-headed attention (multihead_attention.py:9-154)."""

    def __init__(self, embed_dim, num_heads, attn_dropout=0., bias=True, add_bias_kv=False, add_zero_attn=False):
        super().__init__()
        self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.attn_dropout = attn_dropout
        self.head_dim = embed_dim // num_heads
        assert self.head_dim * num_heads == self.embed_dim, "embed_dim must be divisible by num_heads"
        self.scaling = self.head_dim ** -0.5
        self.in_proj_weight = nn.Parameter(torch.Tensor(3 * embed_dim, embed_dim))
        self.register_parameter('in_proj_bias', None)
        if bias:
            self.in_proj_bias = nn.Parameter(torch.Tensor(3 * embed_dim))
        self.out_proj = nn.Linear(embed_dim, embed_dim, bias=bias)
        if add_bias_kv:   # multihead_attention.py:28-32
            self.bias_k = nn.Parameter(torch.Tensor(1, 1, embed_dim))
            self.bias_v = nn.Parameter(torch.Tensor(1, 1, embed_dim))
        else:
            self.bias_k = self.bias_v = None
        self.add_zero_attn = add_zero_attn
        self.reset_parameters()

    def reset_parameters(self):   # multihead_attention.py:40-48
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.xavier_uniform_(self.out_proj.weight)
        if self.in_proj_bias is not None:
            nn.init.constant_(self.in_proj_bias, 0.)
            nn.init.constant_(self.out_proj.bias, 0.)
        if self.bias_k is not None:
            nn.init.xavier_normal_(self.bias_k)
        if self.bias_v is not None:
            nn.init.xavier_normal_(self.bias_v)

    def forward(self, query, key, value, attn_mask=None):
        """Time x Batch x Channel in; returns (attn [T_q, B, E], head-averaged weights [B, T_q, T_k])."""
        tgt_len, bsz, embed_dim = query.size()
        assert embed_dim == self.embed_dim
        assert list(query.size()) == [tgt_len, bsz, embed_dim]
        assert key.size() == value.size()
        src_len = key.size(0)
        if attn_mask is not None and tuple(attn_mask.shape) != (tgt_len, src_len):
            raise RuntimeError(f"attn_mask must be [{tgt_len}, {src_len}], got {tuple(attn_mask.shape)}")
        ext = src_len + (1 if self.bias_k is not None else 0) + (1 if self.add_zero_attn else 0)   # :86-104
        drop = dropout_stream.draw(self.attn_dropout, bsz * self.num_heads, tgt_len, ext, self.training)
        return _MhaFn.apply(query, key, value, self.in_proj_weight, self.in_proj_bias, self.out_proj.weight,
                            self.out_proj.bias, attn_mask, self.num_heads, drop, self.bias_k, self.bias_v, self.add_zero_attn)

    # the projection helpers of multihead_attention.py:133-154
    def in_proj_qkv(self, query):
        return self._in_proj(query).chunk(3, dim=-1)

    def in_proj_kv(self, key):
        return self._in_proj(key, start=self.embed_dim).chunk(2, dim=-1)

    def in_proj_q(self, query, **kwargs):
        return self._in_proj(query, end=self.embed_dim, **kwargs)

    def in_proj_k(self, key):
        return self._in_proj(key, start=self.embed_dim, end=2 * self.embed_dim)

    def in_proj_v(self, value):
        return self._in_proj(value, start=2 * self.embed_dim)

    def _in_proj(self, input, start=0, end=None, **kwargs):
        weight = kwargs.get('weight', self.in_proj_weight)
        bias = kwargs.get('bias', self.in_proj_bias)
        weight = weight[start:end, :]
        if bias is not None:
            bias = bias[start:end]
        return _linear(input, weight, bias)


# ---------------------------------------------------------------------------------------------------
# position_embedding.py
# ---------------------------------------------------------------------------------------------------
def make_positions(tensor, padding_idx, left_pad):
    """Non-padding symbols -> their position numbers (from padding_idx + 1); padding symbols keep padding_idx
    (position_embedding.py:8-26)."""
    seq = tensor.size(1)
    mask = tensor.ne(padding_idx)
    positions = torch.arange(padding_idx + 1, padding_idx + 1 + seq, device=tensor.device).expand_as(tensor)
    if left_pad:
        positions = positions - seq + mask.long().sum(dim=1).unsqueeze(1)
    return torch.where(mask, positions, torch.full_like(positions, padding_idx)).long()


class SinusoidalPositionalEmbedding(nn.Module):
    """Sinusoidal positional embeddings of any length (position_embedding.py:29-76)."""

    def __init__(self, embedding_dim, padding_idx=0, left_pad=0, init_size=128):
        super().__init__()
        self.embedding_dim = embedding_dim
        self.padding_idx = padding_idx
        self.left_pad = left_pad
        self.weights = dict()
        self.register_buffer('_float_tensor', torch.FloatTensor(1))

    @staticmethod
    def get_embedding(num_embeddings, embedding_dim, padding_idx=None):
        half_dim = embedding_dim // 2
        step = math.log(10000) / (half_dim - 1)
        freq = torch.exp(torch.arange(half_dim, dtype=torch.float) * -step)
        ang = torch.arange(num_embeddings, dtype=torch.float).unsqueeze(1) * freq.unsqueeze(0)
        emb = torch.cat([torch.sin(ang), torch.cos(ang)], dim=1).view(num_embeddings, -1)
        if embedding_dim % 2 == 1:
            emb = torch.cat([emb, torch.zeros(num_embeddings, 1)], dim=1)
        if padding_idx is not None:
            emb[padding_idx, :] = 0
        return emb

    def table(self, max_pos, device):
        """[max_pos, E] table on `device` (built on the host with the reference's formula, cached per device)."""
        key = str(device)
        if key not in self.weights or max_pos > self.weights[key].size(0):
            self.weights[key] = self.get_embedding(max_pos, self.embedding_dim, self.padding_idx).to(device)
        return self.weights[key]

    def forward(self, input):
        """input [bsz, seqlen] -> [bsz, seqlen, E] (detached)."""
        bsz, seq_len = input.size()
        max_pos = self.padding_idx + 1 + seq_len
        w = self.table(max_pos, input.device)
        positions = make_positions(input, self.padding_idx, self.left_pad)
        return w.index_select(0, positions.contiguous().view(-1)).view(bsz, seq_len, -1).detach()

    def max_positions(self):
        return int(1e5)


# ---------------------------------------------------------------------------------------------------
# transformer.py
# ---------------------------------------------------------------------------------------------------
class _HipLinear(nn.Linear):
    def forward(self, input):
        return _linear(input, self.weight, self.bias)


class _HipLayerNorm(nn.LayerNorm):
    def forward(self, input):
        return _LayerNormFn.apply(input, self.weight, self.bias, self.eps)


def Linear(in_features, out_features, bias=True):   # transformer.py:193-198
    m = _HipLinear(in_features, out_features, bias)
    nn.init.xavier_uniform_(m.weight)
    if bias:
        nn.init.constant_(m.bias, 0.)
    return m


def LayerNorm(embedding_dim):   # transformer.py:201-203
    return _HipLayerNorm(embedding_dim)


def fill_with_neg_inf(t):
    return t.float().fill_(float('-inf')).type_as(t)


_future_masks = {}


def buffered_future_mask(tensor, tensor2=None):
    """[T_q, T_k] additive mask, -inf strictly above diagonal 1 + |T_k - T_q| (transformer.py:183-190).
    Built once per (T_q, T_k, device) ON the device: the reference rebuilds it on the host and copies it
    over on every layer call, which on this stack stalls the stream for tens of milliseconds."""
    dim1 = dim2 = tensor.size(0)
    if tensor2 is not None:
        dim2 = tensor2.size(0)
    key = (dim1, dim2, str(tensor.device))
    m = _future_masks.get(key)
    if m is None:
        m = torch.triu(torch.full((dim1, dim2), float('-inf'), device=tensor.device), 1 + abs(dim2 - dim1))
        _future_masks[key] = m
    return m


def _dropout_add(x, residual, p, training):
    """residual + F.dropout(x, p, training) (transformer.py:161-162, :171-172)."""
    T, B, E = x.shape
    drop = dropout_stream.draw(p, T, B, E, training)
    return _DropAddFn.apply(x, residual, drop, 1.0, None)


class TransformerEncoderLayer(nn.Module):
    """Pre-LN encoder block (transformer.py:106-181)."""

    def __init__(self, embed_dim, num_heads=4, attn_dropout=0.1, relu_dropout=0.1, res_dropout=0.1, attn_mask=False):
        super().__init__()
        self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.self_attn = MultiheadAttention(embed_dim=self.embed_dim, num_heads=self.num_heads,
                                            attn_dropout=attn_dropout)
        self.attn_mask = attn_mask
        self.relu_dropout = relu_dropout
        self.res_dropout = res_dropout
        self.normalize_before = True
        self.fc1 = Linear(self.embed_dim, 4 * self.embed_dim)
        self.fc2 = Linear(4 * self.embed_dim, self.embed_dim)
        self.layer_norms = nn.ModuleList([LayerNorm(self.embed_dim) for _ in range(2)])

    def _norm(self, i, x, pre):
        """LayerNorm i on the pre- or post- side of its sub-block, whichever `normalize_before` selects."""
        return self.layer_norms[i](x) if pre == self.normalize_before else x

    def maybe_layer_norm(self, i, x, before=False, after=False):   # the reference's public helper, same contract
        assert before ^ after
        return self._norm(i, x, pre=before)

    def _norm_split(self, i, x):
        """(branch input, residual) of sub-block i."""
        if self.normalize_before:
            ln = self.layer_norms[i]
            return _NormResidualFn.apply(x, ln.weight, ln.bias, ln.eps)
        return x, x

    def _attention_block(self, x, x_k, x_v):
        h, x = self._norm_split(0, x)
        mask = buffered_future_mask(h, x_k) if self.attn_mask else None
        if x_k is None and x_v is None:
            k = v = h          # one tensor: the fused self-attention projection path
        else:
            k, v = self._norm(0, x_k, pre=True), self._norm(0, x_v, pre=True)
        h, _ = self.self_attn(query=h, key=k, value=v, attn_mask=mask)
        return self._norm(0, _dropout_add(h, x, self.res_dropout, self.training), pre=False)

    def _ffn_block(self, x):
        h, x = self._norm_split(1, x)
        T, B, _ = h.shape
        drop = dropout_stream.draw(self.relu_dropout, T, B, 4 * self.embed_dim, self.training)
        h = _linear(h, self.fc1.weight, self.fc1.bias, relu=True, drop=drop)   # ReLU + dropout in the GEMM epilogue
        h = self.fc2(h)
        return self._norm(1, _dropout_add(h, x, self.res_dropout, self.training), pre=False)

    def forward(self, x, x_k=None, x_v=None):
        """x [T, B, E] (x_k / x_v [T_k, B, E] for cross-modal attention) -> [T, B, E]
        (transformer.py:137-176: LN -> MHA -> dropout -> +x ; LN -> fc1 -> ReLU -> dropout -> fc2 -> dropout -> +x)."""
        return self._ffn_block(self._attention_block(x, x_k, x_v))


class TransformerEncoder(nn.Module):
    """Stack of TransformerEncoderLayer (transformer.py:10-103)."""

    def __init__(self, embed_dim, num_heads, layers, attn_dropout=0.0, relu_dropout=0.0, res_dropout=0.0,
                 embed_dropout=0.0, attn_mask=False, position_embedding=False):
        super().__init__()
        self.dropout = embed_dropout
        self.attn_dropout = attn_dropout
        self.embed_dim = embed_dim
        self.embed_scale = math.sqrt(embed_dim)
        self.embed_positions = SinusoidalPositionalEmbedding(embed_dim) if position_embedding else None
        self.attn_mask = attn_mask
        self.layers = nn.ModuleList([
            TransformerEncoderLayer(embed_dim, num_heads=num_heads, attn_dropout=attn_dropout,
                                    relu_dropout=relu_dropout, res_dropout=res_dropout, attn_mask=attn_mask)
            for _ in range(layers)])
        self.register_buffer('version', torch.Tensor([2]))
        self.normalize = True
        if self.normalize:
            self.layer_norm = LayerNorm(embed_dim)

    def _embed(self, x_in):
        """F.dropout(embed_scale * x_in + positions) (transformer.py:68-71) in one kernel."""
        T, B, E = x_in.shape
        table = self.embed_positions.table(T + 1, x_in.device) if self.embed_positions is not None else None
        drop = dropout_stream.draw(self.dropout, T, B, E, self.training)
        return _DropAddFn.apply(x_in, None, drop, self.embed_scale, table)

    def forward(self, x_in, x_in_k=None, x_in_v=None):
        x = self._embed(x_in)
        cross = x_in_k is not None and x_in_v is not None
        if cross:
            x_k = self._embed(x_in_k)
            x_v = self._embed(x_in_v)
        for layer in self.layers:
            x = layer(x, x_k, x_v) if cross else layer(x)
        if self.normalize:
            x = self.layer_norm(x)
        return x

    def max_positions(self):
        if self.embed_positions is None:
            return self.max_source_positions
        return min(self.max_source_positions, self.embed_positions.max_positions())
