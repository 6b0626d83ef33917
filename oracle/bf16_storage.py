"""TEST INFRASTRUCTURE — not product code (see oracle/sdumc_oracle.py's header for who may import it).

Emulation, on the oracle's graph, of WHERE the engine's bf16-storage mode (sdumc_net_dims.bf16 = 2, BASELINE configs[2] / [4])
rounds: the frame-level tensors it keeps in HBM as bf16.  Everything else is evaluated in the dtype of the parameters handed in
(the tests use fp64), so a comparison against this separates ROUNDING (reproduced here) from ERRORS (not reproduced) -- the
fp32 oracle alone cannot: bf16 storage moves gradients by several per cent, and a bound that admits that admits bugs.

Rounding points ("bf16 storage"; sdumc_amd/csrc/engine.hip):
  features                      bf16 (TrainStep.set_batch rounds the fp32 batch)
  frame_dim_reshape / input_proj weights   bf16 copies (biases stay fp32)
  x  = Lin(feature)             stored bf16; its gradient dx (sum over sites and streams) stored bf16
  xd = dropout(x)               exact in bf16 (the scale 1/(1-0.5) is a power of two)
  keys = tanh(Lin(xd))          stored bf16; dz = dk (1 - keys^2) with the STORED keys, stored bf16
  dxd                           the pooling path's part is stored bf16, then dz W is added and the sum stored bf16 again
Audio / video frames are projected ONCE per step and shared by the two streams (their dx is one sum): `frames` caches them.
"""
import torch
import torch.nn.functional as F


def bf(t):
    return t.to(torch.bfloat16).to(t.dtype)


class _RoundBoth(torch.autograd.Function):
    """value and incoming gradient both rounded to bf16 (a tensor stored as bf16 whose gradient is stored as bf16)"""

    @staticmethod
    def forward(ctx, x):
        return bf(x)

    @staticmethod
    def backward(ctx, g):
        return bf(g)


class _RoundGrad(torch.autograd.Function):
    """identity whose incoming gradient is rounded to bf16"""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return bf(g)


class _TanhKeys(torch.autograd.Function):
    """keys = bf16(tanh(z)); dz = bf16(dk * (1 - keys^2)) with the stored keys (attnpool_bwd: attn_pool.hip)"""

    @staticmethod
    def forward(ctx, z):
        k = bf(torch.tanh(z))
        ctx.save_for_backward(k)
        return k

    @staticmethod
    def backward(ctx, g):
        (k,) = ctx.saved_tensors
        return bf(g * (1 - k * k))


def _w(P, name):
    """bf16 copy of a weight, gradient passed straight through to the fp32 master"""
    W = P[name]
    return W + (bf(W) - W).detach()


class Bf16Storage:
    def __init__(self):
        self._x = {}

    def frames(self, P, m, feat):
        """x_m = bf16(Lin(bf16 features)); one autograd node per distinct input tensor (audio / video: shared by both streams)"""
        key = (m, id(feat))
        if key not in self._x:
            y = F.linear(bf(feat), _w(P, f"frame_dim_reshape_{m}.weight"), P[f"frame_dim_reshape_{m}.bias"])
            self._x[key] = _RoundBoth.apply(y)
        return self._x[key]

    def site(self, P, lin_name, xd):
        """(xd as the key projection reads it, xd as the pooling reads it, keys) of one attention site"""
        xd = _RoundGrad.apply(xd)                    # dxd total: bf16(bf16(pool path) + dz W)
        xv = _RoundGrad.apply(xd)                    # the pooling path's contribution, stored before dz W is added
        z = F.linear(xd, _w(P, lin_name + ".weight"), P[lin_name + ".bias"])
        return xd, xv, _TanhKeys.apply(z)
