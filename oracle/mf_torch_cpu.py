"""Oracle / CPU baseline: the reference's BPR-MF step as the SAME torch-CPU op sequence
(test infrastructure and bench.py's ``cpu_baseline`` leg only — see oracle/__init__.py).

Where oracle/bpr_mf.py spells the arithmetic out in NumPy, this module issues exactly
the ATen ops the reference issues on its CPU path, so timing it on the GPU box's host
cores is a fair "reference CPU path" number (kind = "port"):

  models/mf.py:9-23            two nn.Embedding(sparse=False) tables, forward = sum(u*i, 1)
  loss.py:25-27                mean(-logsigmoid(pos - neg))
  trainers/mf_trainer.py:104-114   2 x forward, zero_grad, loss, backward, Adam.step, loss.item()
  trainers/base_trainer.py:34-36   torch.optim.Adam(model.parameters(), lr, weight_decay)
"""
import time

import torch
import torch.nn as nn


class MFTorchCPU(nn.Module):
    def __init__(self, U0, I0, lr=1e-4, weight_decay=0.0, optimizer="adam"):
        super().__init__()
        self.user_embedding = nn.Embedding.from_pretrained(torch.as_tensor(U0).clone().float(), freeze=False)
        self.item_embedding = nn.Embedding.from_pretrained(torch.as_tensor(I0).clone().float(), freeze=False)
        self.logsigmoid = nn.LogSigmoid()
        opt = {"adam": torch.optim.Adam, "adamw": torch.optim.AdamW, "sgd": torch.optim.SGD}[optimizer.lower()]
        self.optimizer = opt(self.parameters(), lr=lr, weight_decay=weight_decay)

    def forward(self, user_id, item_id):
        return torch.sum(self.user_embedding(user_id) * self.item_embedding(item_id), dim=1)

    def train_step(self, u, p, n) -> float:
        pos_pred = self(u, p)
        neg_pred = self(u, n)
        self.optimizer.zero_grad()
        loss = torch.mean(-self.logsigmoid(pos_pred - neg_pred))
        loss.backward()
        self.optimizer.step()
        return loss.item()

    def valid_step(self, u, p, n) -> float:
        return torch.mean(-self.logsigmoid(self(u, p) - self(u, n))).item()


def time_steps(num_users, num_items, dim, batches, budget_s=15.0, min_steps=3, lr=1e-4, seed=0):
    """Run train_step over ``batches`` (list of (u,p,n) int64 CPU tensors, cycled) until
    ``budget_s`` seconds of timed work have elapsed.  Returns (triplets/s, steps, seconds)."""
    g = torch.Generator().manual_seed(seed)
    bound_u = (6.0 / (num_users + dim)) ** 0.5
    bound_i = (6.0 / (num_items + dim)) ** 0.5
    U0 = (torch.rand(num_users, dim, generator=g) * 2 - 1) * bound_u
    I0 = (torch.rand(num_items, dim, generator=g) * 2 - 1) * bound_i
    model = MFTorchCPU(U0, I0, lr=lr)
    model.train_step(*batches[0])                     # warm-up (allocations, Adam state)
    done, triplets, t0 = 0, 0, time.perf_counter()
    while True:
        u, p, n = batches[done % len(batches)]
        model.train_step(u, p, n)
        done += 1
        triplets += u.numel()
        el = time.perf_counter() - t0
        if done >= min_steps and el >= budget_s:
            break
        if el >= 2 * budget_s:
            break
    return triplets / el, done, el
