"""BPR matrix factorisation — drop-in for reference models/mf.py:7-23.

Same constructor ``MatrixFactorization(cfg, num_users, num_items)``, same parameter
names (``user_embedding.weight``, ``item_embedding.weight``: state_dict-compatible
with the reference's ``best_model.pt``), same xavier-uniform init in the same order
(so a seeded reference run and a seeded run of this class start from identical
tables), same ``forward(user_id, item_id) -> [B]``.

What differs is underneath: ``forward`` is one gather+dot HIP kernel
(``yr_mf_score``) with a hand-written backward (``yr_mf_score_backward``, dense
scatter-add like nn.Embedding(sparse=False)), and ``bpr_loss_backward`` fuses the two
``forward`` calls, ``BPRLoss`` and ``loss.backward()`` of the reference's train loop
(trainers/mf_trainer.py:106-111) into a single kernel.  Tensors must be on the GPU.
"""
import torch
import torch.nn as nn

from .. import engine
from .base_model import BaseModel


class _MFScore(torch.autograd.Function):
    """score = sum(U[u] * I[i], dim=1) with dense embedding gradients."""

    @staticmethod
    def forward(ctx, U, I, user_id, item_id, err_flag):
        user_id = user_id.contiguous()
        item_id = item_id.contiguous()
        ctx.save_for_backward(U, I, user_id, item_id)
        ctx.err_flag = err_flag
        return engine.mf_score(U.detach(), I.detach(), user_id, item_id, err_flag=err_flag)

    @staticmethod
    def backward(ctx, gout):
        U, I, user_id, item_id = ctx.saved_tensors
        gU = torch.zeros_like(U)          # embedding_dense_backward: zero-filled [rows, D]
        gI = torch.zeros_like(I)
        engine.mf_score_backward(U.detach(), I.detach(), user_id, item_id, gout.contiguous(), gU, gI,
                                 err_flag=ctx.err_flag)
        return gU, gI, None, None, None


class MatrixFactorization(BaseModel):

    def __init__(self, cfg, num_users, num_items):
        super().__init__()
        self.user_embedding = nn.Embedding(num_users, cfg.embed_size, dtype=torch.float32)
        self.item_embedding = nn.Embedding(num_items, cfg.embed_size, dtype=torch.float32)
        self._init_weights()
        self._err_flag = None
        self._loss_partials = None

    def _init_weights(self):
        # reference models/mf.py:15-18 — user table first, then item table
        for child in self.children():
            if isinstance(child, nn.Embedding):
                nn.init.xavier_uniform_(child.weight)

    # -- buffers that live next to the weights ---------------------------------------
    def _flag(self):
        dev = self.user_embedding.weight.device
        if self._err_flag is None or self._err_flag.device != dev:
            self._err_flag = engine.new_error_flag(dev)
        return self._err_flag

    def _partials(self):
        dev = self.user_embedding.weight.device
        if self._loss_partials is None or self._loss_partials.device != dev:
            self._loss_partials = torch.zeros(engine.LOSS_PARTIALS, dtype=torch.float32, device=dev)
        return self._loss_partials

    def check_indices(self):
        """Raise IndexError if any kernel since the last call met an out-of-range id
        (nn.Embedding would have raised at the call; the kernels skip and flag)."""
        if self._err_flag is not None:
            engine.raise_on_flag(self._err_flag, "MatrixFactorization")

    # -- reference surface -------------------------------------------------------------
    def forward(self, user_id, item_id):
        # reference models/mf.py:20-23
        return _MFScore.apply(self.user_embedding.weight, self.item_embedding.weight,
                              user_id, item_id, self._flag())

    # -- fused training op ---------------------------------------------------------------
    def bpr_loss_backward(self, user_id, pos_item, neg_item, loss_out=None, loss_accum=None,
                          inv_batch=None, backward=True):
        """model(u,p), model(u,n), BPRLoss, loss.backward() in ONE kernel
        (reference trainers/mf_trainer.py:106-111).

        Gradients are ACCUMULATED into the dense ``.grad`` of both tables (allocated
        zero-filled on first use, exactly what autograd would leave there); the caller
        clears them via ``optimizer.zero_grad(set_to_none=False)`` or the optimizer's
        fused ``step(zero_grad=True)``.  Returns the batch-mean loss as a 1-element
        device tensor (no host sync); ``loss_accum`` (float64[1]) also receives it.
        ``backward=False`` is the validate path (forward + loss only).
        """
        U, I = self.user_embedding.weight, self.item_embedding.weight
        gU = gI = None
        if backward:
            if U.grad is None:
                U.grad = torch.zeros_like(U)
            if I.grad is None:
                I.grad = torch.zeros_like(I)
            gU, gI = U.grad, I.grad
        B = user_id.numel()
        partials = self._partials()
        engine.bpr_mf_fwd_bwd(U.detach(), I.detach(), user_id.contiguous(), pos_item.contiguous(),
                              neg_item.contiguous(), gU, gI, partials, inv_batch=inv_batch,
                              err_flag=self._flag())
        scale = inv_batch if inv_batch is not None else (1.0 / B if B else 0.0)
        return engine.loss_finalize(partials, scale, loss_out, loss_accum)
