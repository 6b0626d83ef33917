"""Drop-in replacements of the loss callables the training step uses (toolkit/utils/loss.py):
MSELoss (:19-33), RMSELoss (:37-51), RnCLoss (:271-315).  Same call signatures, 0-dim results with
grad; value and gradient come from the HIP kernels (sdumc_amd/csrc/loss.hip)."""
import torch
import torch.nn as nn

from . import ops
from ._lib import SdumcError


def _flat2(pred, target):
    # the reference's view logic (loss.py:26-31 / :44-49)
    if pred.dim() == 1 or target.dim() == 1:
        return pred.reshape(-1, 1), target.reshape(-1, 1)
    if pred.dim() == 3 and target.dim() == 3:
        return pred.reshape(pred.shape[0], -1), target.reshape(target.shape[0], -1)
    return pred, target


def _dev(*ts):
    for t in ts:
        if not t.is_cuda:
            raise SdumcError("sdumc_amd losses run on the GPU only (no CPU fallback)")


class _MSE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target):
        p, t = pred.contiguous().float(), target.contiguous().float()
        loss, dp = ops.mse_fwd_bwd(p.view(-1), t.view(-1), 1.0, denom=pred.shape[0])
        ctx.save_for_backward(dp)
        ctx.shape = pred.shape
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (dp,) = ctx.saved_tensors
        d = (dp * g).view(ctx.shape)
        return d, (-d if ctx.needs_input_grad[1] else None)


class MSELoss(nn.Module):
    def forward(self, pred, target):
        _dev(pred, target)
        p, t = _flat2(pred, target)
        if p.shape != t.shape:
            raise SdumcError(f"MSELoss: shapes {tuple(p.shape)} vs {tuple(t.shape)}")
        return _MSE.apply(p, t)


class _RMSE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a_, b_ = a.contiguous().float(), b.contiguous().float()
        loss, da, _ = ops.rmse_fwd_bwd(a_, b_, need_db=False)
        ctx.save_for_backward(da)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (da,) = ctx.saved_tensors
        d = da * g
        return (d if ctx.needs_input_grad[0] else None), (-d if ctx.needs_input_grad[1] else None)


class RMSELoss(nn.Module):
    def forward(self, pred, target):
        _dev(pred, target)
        p, t = _flat2(pred, target)
        return _RMSE.apply(p, t)


class _RnC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, labels, temperature):
        B = features.shape[0]
        feats = torch.cat([features[:, 0], features[:, 1]], dim=0).contiguous().float()   # loss.py:282
        y = labels.reshape(B, -1)
        if y.shape[1] != 1:
            raise SdumcError("RnCLoss: label_dim must be 1 on this path")
        y2 = y.repeat(2, 1).reshape(-1).contiguous().float()                                # loss.py:283
        loss, df, _ = ops.rnc_fwd_bwd(feats, y2, temperature=temperature)
        ctx.save_for_backward(df)
        ctx.B = B
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (df,) = ctx.saved_tensors
        B = ctx.B
        d = torch.stack((df[:B], df[B:]), dim=1) * g
        return d, None, None


class RnCLoss(nn.Module):
    def __init__(self, temperature=2, label_diff='l1', feature_sim='l2'):
        super().__init__()
        if label_diff != 'l1' or feature_sim != 'l2':
            raise SdumcError("only label_diff='l1', feature_sim='l2' (the reference defaults) are built")
        self.t = float(temperature)

    def forward(self, features, labels):
        _dev(features, labels)
        return _RnC.apply(features, labels, self.t)
