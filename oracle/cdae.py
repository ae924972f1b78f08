"""Oracle: CDAE forward, NS-BCE loss and their gradients (test infrastructure — see
oracle/__init__.py).  float32 NumPy, explicit formulas, no autograd.

Restates
  * reference models/cdae.py:46-52   forward:  z = act_h(W_h . dropout(x) + b_h + V[u]),
                                               y = act_o(W_o z + b_o)
      (``if self.train:`` at :47 tests a bound method and is always true, but nn.Dropout itself
      follows .train()/.eval(): train = inverted dropout, scale 1/(1-p); eval = identity)
  * reference loss.py:12-16          NSBCELoss: idx = nonzero(target + negative_mask);
                                               binary_cross_entropy(input[idx], target[idx]), mean
      (torch clamps each log term at -100 and the gradient's denominator at 1e-12)
  * reference trainers/cdae_trainer.py:123-144  mask by MULTIPLYING scores with
      logical_not(input_mask) (-> 0), then row-wise argpartition / argsort
  * torch.optim.Adam over (hidden_layer.weight, hidden_layer.bias, user_nodes.weight,
    output_layer.weight, output_layer.bias) — module order of models/cdae.py:17-29.
"""
from __future__ import annotations

import numpy as np

from .adam import Optimizer

F32 = np.float32


def sigmoid(x):
    return (F32(1) / (F32(1) + np.exp(-x.astype(F32)))).astype(F32)


def act(name, x):
    return sigmoid(x) if name == "sigmoid" else x.astype(F32)


def act_grad(name, y):
    """d act / d pre expressed through the OUTPUT y."""
    return (y * (F32(1) - y)).astype(F32) if name == "sigmoid" else np.ones_like(y)


def forward(params, user_id, x_in, hidden_act="sigmoid", output_act="sigmoid"):
    """x_in is the (already corrupted, if training) input [B, I].  Returns (y, z)."""
    Wh, bh, V, Wo, bo = params
    z = act(hidden_act, (x_in.astype(F32) @ Wh.T + bh + V[user_id]).astype(F32))
    y = act(output_act, (z @ Wo.T + bo).astype(F32))
    return y, z


def nsbce_loss(pred, target, negative_mask):
    sel = (target + negative_mask) != 0
    p, t = pred[sel].astype(F32), target[sel].astype(F32)
    with np.errstate(divide="ignore"):
        lp = np.maximum(np.log(p), F32(-100))
        lq = np.maximum(np.log(F32(1) - p), F32(-100))
    return F32(np.mean(-(t * lp + (F32(1) - t) * lq), dtype=F32)), sel


def loss_and_grads(params, user_id, x_in, target, negative_mask, hidden_act="sigmoid", output_act="sigmoid"):
    Wh, bh, V, Wo, bo = params
    y, z = forward(params, user_id, x_in, hidden_act, output_act)
    loss, sel = nsbce_loss(y, target.astype(F32), negative_mask.astype(F32))
    cnt = F32(sel.sum())
    dy = np.zeros_like(y)
    dy[sel] = ((y[sel] - target.astype(F32)[sel]) / np.maximum((F32(1) - y[sel]) * y[sel], F32(1e-12)) / cnt)
    dpre_o = (dy * act_grad(output_act, y)).astype(F32)
    dWo = (dpre_o.T @ z).astype(F32)
    dbo = dpre_o.sum(axis=0, dtype=F32)
    dz = (dpre_o @ Wo).astype(F32)
    dpre_h = (dz * act_grad(hidden_act, z)).astype(F32)
    dWh = (dpre_h.T @ x_in.astype(F32)).astype(F32)
    dbh = dpre_h.sum(axis=0, dtype=F32)
    dV = np.zeros_like(V)
    np.add.at(dV, user_id, dpre_h)
    return loss, [dWh, dbh, dV, dWo, dbo]


def top_k_multiply_mask(pred, pred_mask, k):
    """cdae_trainer.py:123-144: scores * logical_not(mask), top-k per row (score desc, id asc)."""
    s = pred * np.logical_not(pred_mask)
    order = np.lexsort((np.broadcast_to(np.arange(s.shape[1]), s.shape), -s), axis=1)
    return order[:, :k]


class CDAEState:
    def __init__(self, params, lr=1e-4, optimizer="adam", hidden_act="sigmoid", output_act="sigmoid"):
        self.params = [np.array(p, dtype=F32, copy=True) for p in params]
        self.hidden_act, self.output_act = hidden_act, output_act
        self.opt = Optimizer(optimizer, self.params, lr=lr)

    def train_step(self, user_id, x_corrupted, target, negative_mask):
        loss, grads = loss_and_grads(self.params, user_id, x_corrupted, target, negative_mask,
                                     self.hidden_act, self.output_act)
        self.opt.step(grads)
        return loss

    def predict(self, user_id, x):
        return forward(self.params, user_id, x, self.hidden_act, self.output_act)[0]
