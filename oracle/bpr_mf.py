"""Oracle: BPR matrix factorisation step (test infrastructure — see oracle/__init__.py).

Restates, in float32 NumPy with explicit formulas:
  * reference models/mf.py:20-23       MatrixFactorization.forward
  * reference loss.py:25-27            BPRLoss.forward  = mean(-logsigmoid(pos - neg))
  * the autograd of both (triggered at trainers/mf_trainer.py:111), i.e. torch's
    embedding_dense_backward (zero-filled [rows, D] + index_add_)
  * reference trainers/mf_trainer.py:100-116 / :118-132   train / validate loops
"""
from __future__ import annotations

import numpy as np

from .adam import Optimizer

F32 = np.float32


def xavier_uniform_bound(rows: int, dim: int) -> float:
    """models/mf.py:15-18 -> nn.init.xavier_uniform_ on a [rows, dim] weight:
    bound = sqrt(6 / (fan_in + fan_out)) with fan_in = dim, fan_out = rows."""
    return float(np.sqrt(6.0 / (rows + dim)))


def forward(U: np.ndarray, I: np.ndarray, user_id: np.ndarray, item_id: np.ndarray) -> np.ndarray:
    """models/mf.py:20-23: sum(user_emb * item_emb, dim=1) -> [B] float32."""
    return np.sum(U[user_id] * I[item_id], axis=1, dtype=F32)


def log_sigmoid(x: np.ndarray) -> np.ndarray:
    """ATen log_sigmoid_forward: min(x, 0) - log1p(exp(-|x|)) (float32)."""
    x = x.astype(F32, copy=False)
    return (np.minimum(x, F32(0)) - np.log1p(np.exp(-np.abs(x)))).astype(F32)


def bpr_loss(pos: np.ndarray, neg: np.ndarray) -> np.float32:
    """loss.py:25-27: mean over the ACTUAL batch length (last batch is short)."""
    return F32(np.mean(-log_sigmoid(pos - neg), dtype=F32))


def bpr_coeff(pos: np.ndarray, neg: np.ndarray) -> np.ndarray:
    """d loss / d (pos - neg) per triplet = -sigmoid(-(pos-neg)) / B  (float32)."""
    x = (pos - neg).astype(F32)
    z = np.exp(-np.abs(x))
    sig_neg = np.where(x < 0, F32(1) / (F32(1) + z), z / (F32(1) + z)).astype(F32)
    return (-sig_neg / F32(x.shape[0])).astype(F32)


def loss_and_grads(U, I, u, p, n):
    """One fwd+bwd of mf_trainer.py:106-111.  Returns (loss, gradU, gradI) with the
    gradients DENSE ([U,D], [I,D], zero where untouched) as nn.Embedding(sparse=False)
    produces them; duplicate rows inside a batch accumulate (index_add_)."""
    pos = forward(U, I, u, p)
    neg = forward(U, I, u, n)
    loss = bpr_loss(pos, neg)
    g = bpr_coeff(pos, neg)[:, None]
    gU = np.zeros_like(U)
    gI = np.zeros_like(I)
    # user rows receive two separate index_adds (one per model() call)
    np.add.at(gU, u, g * I[p])
    np.add.at(gU, u, -g * I[n])
    np.add.at(gI, p, g * U[u])
    np.add.at(gI, n, -g * U[u])
    return loss, gU, gI


class MFState:
    """Weights + optimizer state of one BPR-MF model, advanced by ``train_step``."""

    def __init__(self, U0, I0, optimizer="adam", lr=1e-4, weight_decay=0.0):
        self.U = np.array(U0, dtype=F32, copy=True)
        self.I = np.array(I0, dtype=F32, copy=True)
        # base_trainer.py:34-43 builds ONE optimizer over model.parameters():
        # user table first, item table second (models/mf.py:11-12)
        self.opt = Optimizer(optimizer, [self.U, self.I], lr=lr, weight_decay=weight_decay)

    def train_step(self, u, p, n) -> np.float32:
        """mf_trainer.py:104-114 for one batch; returns loss.item()."""
        loss, gU, gI = loss_and_grads(self.U, self.I, u, p, n)
        self.opt.step([gU, gI])
        return loss

    def valid_step(self, u, p, n) -> np.float32:
        """mf_trainer.py:123-130 (forward + loss only)."""
        return bpr_loss(forward(self.U, self.I, u, p), forward(self.U, self.I, u, n))

    def train_epoch(self, u, p, n, batch_sizes):
        """mf_trainer.py:100-116: returns (SUM of per-batch mean losses, per-step losses)."""
        total, steps, pos = 0.0, [], 0
        for b in batch_sizes:
            s = slice(pos, pos + int(b))
            l = self.train_step(u[s], p[s], n[s])
            steps.append(float(l))
            total += float(l)
            pos += int(b)
        return total, np.asarray(steps)

    def valid_epoch(self, u, p, n, batch_sizes):
        total, steps, pos = 0.0, [], 0
        for b in batch_sizes:
            s = slice(pos, pos + int(b))
            l = self.valid_step(u[s], p[s], n[s])
            steps.append(float(l))
            total += float(l)
            pos += int(b)
        return total, np.asarray(steps)
