"""Trainer base — drop-in for reference trainers/base_trainer.py:14-157.

Same constructor, same helper names and semantics (`_device`, `_model`,
`_optimizer`, `_loss`, `run`, `_is_surpass_best_metric`, `load_best_model`); the
optimizers it hands out are the dense HIP-backed ones of :mod:`..optim`.
"""
import os
from abc import ABC, abstractmethod

import torch
from torch.nn import Module

from .. import optim
from ..utils import logger


# position of every selectable metric in the (valid_loss, precision, recall, map, ndcg) tuples and
# whether smaller is better (reference base_trainer.py:117-141)
_BEST_METRIC = {"loss": (0, True), "precision": (1, False), "recall": (2, False), "map": (3, False), "ndcg": (4, False)}
_OPTIMIZERS = {"adam": optim.Adam, "adamw": optim.AdamW, "sgd": optim.SGD}


class BaseTrainer(ABC):

    def __init__(self, cfg) -> None:
        self.cfg = cfg
        self.device: torch.device = self._device(cfg.device)
        os.makedirs(cfg.model_dir, exist_ok=True)

    def _device(self, device_name: str) -> torch.device:
        """'cuda' is the ROCm device on MI355X.  'cpu' stays a legal name (the reference's default,
        base_trainer.py:20-25) but the HIP ops refuse CPU tensors, so such a run fails loudly at the
        first model call instead of silently falling back; any other name logs and yields cpu."""
        name = device_name.lower()
        if name not in ("cpu", "cuda"):
            logger.error(f"Not supported device: {device_name}")
            name = "cpu"
        return torch.device(name)

    def _model(self, model_name: str) -> Module:
        # reference base_trainer.py:27-32: only the placeholder name 'test' is known at this level
        if model_name.lower() != "test":
            logger.error(f"Not implemented model: {model_name}")
            raise NotImplementedError(f"Not implemented model: {model_name}")
        return type("TestModel", (Module,), {"forward": (lambda self, x: x)})()

    def _optimizer(self, optimizer_name: str, model: Module, learning_rate: float, weight_decay: float = 0):
        """adam / adamw / sgd over all parameters with torch's default betas / eps (reference
        base_trainer.py:34-43), as the dense HIP-backed optimizers of :mod:`..optim`."""
        make = _OPTIMIZERS.get(optimizer_name.lower())
        if make is None:
            logger.error(f"Optimizer Not Exists: {optimizer_name}")
            raise NotImplementedError(f"Optimizer Not Exists: {optimizer_name}")
        return make(model.parameters(), lr=learning_rate, weight_decay=weight_decay)

    def _loss(self, loss_name: str):
        # reference base_trainer.py:45-50
        if loss_name.lower() != "bce":
            logger.error(f"Loss Not Exists: {loss_name}")
            raise NotImplementedError(f"Loss Not Exists: {loss_name}")
        return torch.nn.BCELoss()

    def run(self, train_dataloader, valid_dataloader):
        # reference base_trainer.py:52-115 (validate() returns loss + the four metrics)
        logger.info("[Trainer] run...")
        best = (1e+6, .0, .0, .0, .0)
        endurance = 0
        for epoch in range(self.cfg.epochs):
            train_loss = self.train(train_dataloader)
            current = self.validate(valid_dataloader)
            self._log_epoch(epoch, train_loss, *current)
            if self._is_surpass_best_metric(current=current, best=best):
                logger.info("[Trainer] update best model...")
                best = tuple(current)
                endurance = 0
                torch.save(self.model.state_dict(), f'{self.cfg.model_dir}/best_model.pt')
            else:
                endurance += 1
                if endurance > self.cfg.patience:
                    logger.info("[Trainer] ealry stopping...")
                    break

    def _log_epoch(self, epoch, train_loss, valid_loss, p, r, m, n):
        logger.info(f"[Trainer] epoch: {epoch} > train loss: {train_loss:.4f} / valid loss: {valid_loss:.4f} / "
                    f"precision@K : {p:.4f} / Recall@K: {r:.4f} / MAP@K: {m:.4f} / NDCG@K: {n:.4f}")

    def _is_surpass_best_metric(self, **metric) -> bool:
        """``current`` beats ``best`` on ``cfg.best_metric`` (both are (valid_loss, precision, recall,
        map, ndcg) tuples; an unknown metric name never improves — reference base_trainer.py:117-141)."""
        choice = _BEST_METRIC.get(self.cfg.best_metric)
        if choice is None:
            return False
        position, smaller_is_better = choice
        current, best = metric['current'][position], metric['best'][position]
        return current < best if smaller_is_better else current > best

    @abstractmethod
    def train(self, train_dataloader) -> float:
        pass

    @abstractmethod
    def validate(self, valid_dataloader):
        pass

    @abstractmethod
    def evaluate(self, test_dataloader):
        pass

    # -- resume (not in the reference, which saves weights only: SURVEY.md §8 f4) ---------------
    def save_checkpoint(self, path, **extra):
        """Weights (the reference's ``state_dict`` keys) + optimizer state (Adam moments and step
        counts) + anything the caller wants to find again (epoch, best metric, ...)."""
        torch.save({"model": self.model.state_dict(), "optimizer": self.optimizer.state_dict(), "extra": extra}, path)

    def load_checkpoint(self, path):
        """Restore what :meth:`save_checkpoint` wrote; training continues as if it had never stopped (to
        float rounding) when fed the same batches.  Returns the ``extra`` dict."""
        ck = torch.load(path, map_location=self.device, weights_only=True)
        self.model.load_state_dict(ck["model"])
        self.optimizer.load_state_dict(ck["optimizer"])
        return ck.get("extra", {})

    def load_best_model(self):
        # reference base_trainer.py:155-157 (weights_only: the file holds tensors only)
        logger.info("[Trainer] Load best model...")
        state = torch.load(f'{self.cfg.model_dir}/best_model.pt', map_location=self.device, weights_only=True)
        self.model.load_state_dict(state)
