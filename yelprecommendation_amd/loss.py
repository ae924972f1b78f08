"""Losses — drop-in for reference loss.py (BPRLoss :19-27).

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
