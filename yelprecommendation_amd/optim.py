"""Dense optimizers backed by the HIP kernels — what reference
trainers/base_trainer.py:34-43 builds (torch.optim.Adam / AdamW / SGD over
``model.parameters()`` with torch's default betas/eps), as ``torch.optim.Optimizer``
subclasses so ``zero_grad()`` / ``step()`` / ``state_dict()`` keep their meaning.

Dense on purpose: the reference's embeddings are ``sparse=False``, so Adam decays
m/v of EVERY row each step and rows touched earlier keep moving; a lazy update
would not be the same algorithm.
"""
import torch
from torch.optim import Optimizer

from . import engine


class _DenseBase(Optimizer):
    def _grad(self, p):
        g = p.grad
        if g is None:
            return None
        if g.is_sparse:
            raise RuntimeError("dense optimizers only (nn.Embedding(sparse=False) as in the reference)")
        return g


class Adam(_DenseBase):
    _decoupled = False
    SMALL = 1 << 18            # tensors up to this many elements share one multi-tensor launch

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None, zero_grad=False):
        """One update of every parameter that has a gradient.  ``zero_grad=True``
        also clears the gradients in the same pass (saves the separate memset that
        ``optimizer.zero_grad()`` costs at mf_trainer.py:109)."""
        if closure is not None:
            raise NotImplementedError("closure is not used by the reference trainers")
        for group in self.param_groups:
            b1, b2 = group["betas"]
            small = {}                                   # step count -> [(p, g, m, v)] of the small tensors
            for p in group["params"]:
                g = self._grad(p)
                if g is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] += 1
                if p.numel() <= self.SMALL:
                    small.setdefault(st["step"], []).append((p.data, g, st["exp_avg"], st["exp_avg_sq"]))
                    continue
                engine.adam_dense(p.data, g, st["exp_avg"], st["exp_avg_sq"], st["step"], group["lr"],
                                  b1, b2, group["eps"], group["weight_decay"],
                                  decoupled=self._decoupled, zero_grad=zero_grad)
            # weight matrices and biases: one launch for all of them instead of ~5 us each
            for step, tensors in small.items():
                engine.adam_dense_multi(tensors, step, group["lr"], b1, b2, group["eps"], group["weight_decay"],
                                        decoupled=self._decoupled, zero_grad=zero_grad)


class AdamW(Adam):
    _decoupled = True

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, lr, betas, eps, weight_decay)


class SGD(_DenseBase):
    def __init__(self, params, lr=1e-3, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None, zero_grad=False):
        if closure is not None:
            raise NotImplementedError("closure is not used by the reference trainers")
        for group in self.param_groups:
            for p in group["params"]:
                g = self._grad(p)
                if g is None:
                    continue
                engine.sgd_dense(p.data, g, group["lr"], group["weight_decay"], zero_grad=zero_grad)
