"""Losses — drop-in for reference loss.py (NSBCELoss :7-16, BPRLoss :19-27).

``NSBCELoss()(input, target, negative_mask)``: binary cross-entropy over the positions where
``target + negative_mask != 0`` (mean), forward and backward in HIP (``yr_nsbce_fwd/bwd``).

``BPRLoss()(pos, neg) = mean(-logsigmoid(pos - neg))`` with the reduction and its
backward in HIP (``yr_bpr_loss_fwd`` / ``yr_bpr_loss_bwd``); the mean is over the
actual batch length.  The fused trainer path (MatrixFactorization.bpr_loss_backward)
never materialises ``pos``/``neg`` and does not come through here.
"""
import torch
import torch.nn as nn

from . import engine


class _BPRLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pos, neg):
        pos, neg = pos.contiguous(), neg.contiguous()
        ctx.save_for_backward(pos, neg)
        partials = torch.empty(engine.LOSS_PARTIALS, dtype=torch.float32, device=pos.device)
        engine.bpr_loss_fwd(pos, neg, partials)
        B = pos.numel()
        return engine.loss_finalize(partials, 1.0 / B if B else 0.0).reshape(())

    @staticmethod
    def backward(ctx, gout):
        pos, neg = ctx.saved_tensors
        gpos, gneg = torch.empty_like(pos), torch.empty_like(neg)
        engine.bpr_loss_bwd(pos, neg, gout.reshape(1).contiguous(), gpos, gneg)
        return gpos, gneg


class BPRLoss(nn.Module):

    def __init__(self):
        super().__init__()

    def forward(self, positive_preds, negative_preds):
        return _BPRLossFn.apply(positive_preds, negative_preds)


class _NSBCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, negative_mask):
        pred, target = pred.contiguous(), target.contiguous()
        negative_mask = None if negative_mask is None else negative_mask.contiguous()
        stats = engine.nsbce_fwd(pred.detach(), target, negative_mask)
        ctx.save_for_backward(pred.detach(), target, stats)
        ctx.negative_mask = negative_mask
        return stats[0].clone()

    @staticmethod
    def backward(ctx, gout):
        pred, target, stats = ctx.saved_tensors
        return engine.nsbce_bwd(pred, target, ctx.negative_mask, stats, gout.reshape(1).contiguous()), None, None


class NSBCELoss(nn.Module):
    """reference loss.py:7-16 (weight / reduction options of nn.BCELoss are not used by the
    reference's trainers; mean reduction, no weight)."""

    def __init__(self, weight=None, size_average=None, reduce=None, reduction: str = 'mean') -> None:
        super().__init__()
        if weight is not None or reduction != 'mean':
            raise NotImplementedError("NSBCELoss: only the unweighted mean used by the reference's trainers")

    def forward(self, input, target, negative_mask):
        return _NSBCEFn.apply(input, target, negative_mask)


class BCELoss(nn.Module):
    """nn.BCELoss() as the reference's CDAE trainer uses it when negative_sampling is off
    (trainers/cdae_trainer.py:29-30): mean BCE over ALL positions."""

    def forward(self, input, target):
        return _NSBCEFn.apply(input, target, None)
