"""Oracle: NGCF propagation, scoring and their gradients (test infrastructure — see
oracle/__init__.py).  float32 NumPy + scipy.sparse, explicit formulas, no autograd.

Restates
  * reference data/datasets/ngcf_data_pipeline.py:19-44   _set_laplacian_matrix
      R = pivot_table(user x item, values=rating)  (MEAN over duplicate reviews, not binarised),
      A = [[0, R], [R^T, 0]],  L = D^-1/2 A D^-1/2,  D = diag(column sums of A)
  * reference models/ngcf.py:60-72   embedding_propagation
      E' = leaky_relu( W1((L + I) E) + W2(E * (L E)) ),  Linear(x) = x @ W^T, slope 0.01
      (the reference materialises eye(N, N) per layer per batch; (L + I)E = LE + E)
  * reference models/ngcf.py:30-58   bpr_forward / forward: concat of the layer-0..K rows, dot
  * autograd of all of it + Adam (trainers/ngcf_trainer.py:102-117)
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

from . import bpr_mf
from .adam import Optimizer

F32 = np.float32
SLOPE = F32(0.01)


def laplacian_csr(user_id, item_id, rating, num_users, num_items):
    """CSR (float32) of the rating-weighted normalised adjacency, N = num_users + num_items."""
    user_id = np.asarray(user_id, dtype=np.int64)
    item_id = np.asarray(item_id, dtype=np.int64)
    rating = np.asarray(rating, dtype=np.float64)
    # pivot_table(values='rating') default aggfunc = mean over duplicate (user, item) rows
    key = user_id * num_items + item_id
    uniq, inv = np.unique(key, return_inverse=True)
    mean = (np.bincount(inv, weights=rating) / np.bincount(inv)).astype(F32)
    u, i = uniq // num_items, uniq % num_items
    n = num_users + num_items
    rows = np.concatenate([u, num_users + i])
    cols = np.concatenate([num_users + i, u])
    vals = np.concatenate([mean, mean]).astype(F32)
    keep = vals != 0                                    # fillna(0) / to_sparse() drop zero entries
    A = sp.csr_matrix((vals[keep], (rows[keep], cols[keep])), shape=(n, n), dtype=F32)
    deg = np.asarray(A.sum(axis=0)).ravel().astype(F32)
    with np.errstate(divide="ignore"):
        d = (F32(1) / np.sqrt(deg)).astype(F32)         # inf for isolated nodes, as in the reference
    A = A.tocoo()
    val = ((d[A.row] * A.data).astype(F32) * d[A.col]).astype(F32)
    L = sp.csr_matrix((val, (A.row, A.col)), shape=(n, n), dtype=F32)
    L.sort_indices()
    return L


def leaky_relu(x):
    return np.where(x > 0, x, SLOPE * x).astype(F32)


def propagate(E, W1, W2, L):
    """models/ngcf.py:60-72.  Returns (E_next, cache for backward)."""
    Z = (L @ E).astype(F32)                             # neighbor_embeddings = L E
    A = (Z + E).astype(F32)                             # (L + I) E
    H = (E * Z).astype(F32)
    P = (A @ W1.T + H @ W2.T).astype(F32)
    return leaky_relu(P), (E, Z, A, H, P)


def propagate_backward(dE_next, cache, W1, W2, L):
    """Gradients of propagate(): returns (dE, dW1, dW2).  L is symmetric, so L^T = L."""
    E, Z, A, H, P = cache
    dP = (dE_next * np.where(P > 0, F32(1), SLOPE)).astype(F32)
    dW1 = (dP.T @ A).astype(F32)
    dW2 = (dP.T @ H).astype(F32)
    dA = (dP @ W1).astype(F32)
    dH = (dP @ W2).astype(F32)
    dZ = (dA + dH * E).astype(F32)
    dE = (dA + dH * Z + (L.T @ dZ)).astype(F32)
    return dE, dW1, dW2


def layer_outputs(E0, W1s, W2s, L):
    """[E_0, E_1, ..., E_K] and the per-layer caches."""
    outs, caches = [E0], []
    for W1, W2 in zip(W1s, W2s):
        nxt, c = propagate(outs[-1], W1, W2, L)
        outs.append(nxt)
        caches.append(c)
    return outs, caches


def bpr_forward(E0, W1s, W2s, L, num_users, u, p, n):
    """models/ngcf.py:30-45 -> (pos, neg)."""
    outs, _ = layer_outputs(E0, W1s, W2s, L)
    cat = np.concatenate(outs, axis=1)
    return (np.sum(cat[u] * cat[num_users + p], axis=1, dtype=F32),
            np.sum(cat[u] * cat[num_users + n], axis=1, dtype=F32))


def forward(E0, W1s, W2s, L, num_users, u, i):
    """models/ngcf.py:47-58."""
    outs, _ = layer_outputs(E0, W1s, W2s, L)
    cat = np.concatenate(outs, axis=1)
    return np.sum(cat[u] * cat[num_users + i], axis=1, dtype=F32)


def loss_and_grads(E0, W1s, W2s, L, num_users, u, p, n):
    """One forward + backward of ngcf_trainer.py:106-112: (loss, dE0, [dW1_k], [dW2_k])."""
    outs, caches = layer_outputs(E0, W1s, W2s, L)
    pu, pp, pn = u, num_users + p, num_users + n
    pos = sum(np.sum(o[pu] * o[pp], axis=1, dtype=F32) for o in outs).astype(F32)
    neg = sum(np.sum(o[pu] * o[pn], axis=1, dtype=F32) for o in outs).astype(F32)
    loss = bpr_mf.bpr_loss(pos, neg)
    g = bpr_mf.bpr_coeff(pos, neg)[:, None]
    d_outs = []
    for o in outs:
        d = np.zeros_like(o)
        np.add.at(d, pu, g * (o[pp] - o[pn]))
        np.add.at(d, pp, g * o[pu])
        np.add.at(d, pn, -g * o[pu])
        d_outs.append(d)
    dW1s, dW2s = [None] * len(W1s), [None] * len(W2s)
    carry = d_outs[-1]
    for k in range(len(W1s) - 1, -1, -1):
        dE, dW1s[k], dW2s[k] = propagate_backward(carry, caches[k], W1s[k], W2s[k], L)
        carry = (d_outs[k] + dE).astype(F32)
    return loss, carry, dW1s, dW2s


class NGCFState:
    """Parameters + one Adam over them in the reference's parameter order
    (embedding.weight, W1.0.., W2.0..: models/ngcf.py:15-23)."""

    def __init__(self, E0, W1s, W2s, L, num_users, lr=1e-4, optimizer="adam", weight_decay=0.0):
        self.E = np.array(E0, dtype=F32, copy=True)
        self.W1 = [np.array(w, dtype=F32, copy=True) for w in W1s]
        self.W2 = [np.array(w, dtype=F32, copy=True) for w in W2s]
        self.L, self.num_users = L, num_users
        self.opt = Optimizer(optimizer, [self.E] + self.W1 + self.W2, lr=lr, weight_decay=weight_decay)

    def train_step(self, u, p, n):
        loss, dE, dW1, dW2 = loss_and_grads(self.E, self.W1, self.W2, self.L, self.num_users, u, p, n)
        self.opt.step([dE] + dW1 + dW2)
        return loss

    def valid_step(self, u, p, n):
        pos, neg = bpr_forward(self.E, self.W1, self.W2, self.L, self.num_users, u, p, n)
        return bpr_mf.bpr_loss(pos, neg)
