"""Oracle: the optimizers the reference builds (test infrastructure — see oracle/__init__.py).

reference trainers/base_trainer.py:34-43 ->
    Adam(model.parameters(), lr, weight_decay)   (L2-style weight decay)
    AdamW(...)                                   (decoupled weight decay)
    SGD(..., lr, weight_decay)                   (no momentum)
all with torch defaults (betas=(0.9, 0.999), eps=1e-8, amsgrad=False) and DENSE
gradients, so every row of every table moves on every step.

The arithmetic is torch's (third-party, pinned torch==2.2.2 in poetry.lock:2187-2188,
not under /root/reference); this restates the CPU single-tensor path
(torch/optim/adam.py::_single_tensor_adam, the branch taken for CPU parameters):

    grad   = grad + wd * param                        (Adam, wd != 0)
    param *= 1 - lr * wd                              (AdamW, wd != 0)
    m      = m + (1 - b1) * (grad - m)                (lerp_)
    v      = v * b2 + (1 - b2) * grad * grad          (mul_, addcmul_)
    denom  = sqrt(v) / sqrt(1 - b2^t) + eps
    param  = param + (-(lr / (1 - b1^t)) * m) / denom (addcdiv_)

Scalars (bias corrections, step size) are Python doubles rounded to float32 at the
tensor op, as in torch.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32


def adam_scalars(step: int, lr: float, beta1: float = 0.9, beta2: float = 0.999):
    """Host-side doubles of one Adam step: (step_size, bias_correction2_sqrt)."""
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    return lr / bc1, bc2 ** 0.5


def adam_update(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8,
                weight_decay=0.0, decoupled=False):
    """In-place Adam/AdamW update of one float32 tensor (all arrays same shape)."""
    if weight_decay != 0:
        if decoupled:
            p *= F32(1.0 - lr * weight_decay)
        else:
            g = g + F32(weight_decay) * p
    step_size, bc2_sqrt = adam_scalars(step, lr, beta1, beta2)
    m += F32(1.0 - beta1) * (g - m)
    v *= F32(beta2)
    v += (F32(1.0 - beta2) * g) * g
    denom = np.sqrt(v) / F32(bc2_sqrt) + F32(eps)
    p += (F32(-step_size) * m) / denom


def sgd_update(p, g, lr, weight_decay=0.0):
    """torch.optim.SGD without momentum: p -= lr * (g + wd * p)."""
    if weight_decay != 0:
        g = g + F32(weight_decay) * p
    p += F32(-lr) * g


class Optimizer:
    """One optimizer over a list of float32 arrays (updated IN PLACE), mirroring
    base_trainer.py:34-43.  ``step(grads)`` takes dense grads in parameter order."""

    def __init__(self, name, params, lr, weight_decay=0.0, betas=(0.9, 0.999), eps=1e-8):
        self.name = name.lower()
        if self.name not in ("adam", "adamw", "sgd"):
            raise NotImplementedError(f"Optimizer Not Exists: {name}")   # base_trainer.py:41-43
        self.params = params
        self.lr, self.wd, self.betas, self.eps = lr, weight_decay, betas, eps
        self.t = 0
        self.m = [np.zeros_like(p) for p in params]
        self.v = [np.zeros_like(p) for p in params]

    def step(self, grads):
        self.t += 1
        for p, g, m, v in zip(self.params, grads, self.m, self.v):
            if self.name == "sgd":
                sgd_update(p, g, self.lr, self.wd)
            else:
                adam_update(p, g, m, v, self.t, self.lr, self.betas[0], self.betas[1], self.eps,
                            self.wd, decoupled=(self.name == "adamw"))
