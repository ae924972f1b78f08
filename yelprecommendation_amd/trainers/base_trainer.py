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


class BaseTrainer(ABC):
    def __init__(self, cfg) -> None:
        self.cfg = cfg
        self.device: torch.device = self._device(self.cfg.device)
        os.makedirs(self.cfg.model_dir, exist_ok=True)

    def _device(self, device_name: str) -> torch.device:
        # reference base_trainer.py:20-25.  'cuda' is the ROCm device on MI355X.  'cpu' is
        # still a legal name (the reference's default) but the HIP ops refuse CPU tensors,
        # so a cpu run fails loudly at the first model call instead of silently falling back.
        if device_name.lower() in ('cpu', 'cuda',):
            return torch.device(device_name.lower())
        else:
            logger.error(f"Not supported device: {device_name}")
            return torch.device('cpu')

    def _model(self, model_name: str) -> Module:
        # reference base_trainer.py:27-32
        if model_name.lower() in ('test',):
            return type("TestModel", (Module,), {"forward": (lambda self, x: x)})()
        else:
            logger.error(f"Not implemented model: {model_name}")
            raise NotImplementedError(f"Not implemented model: {model_name}")

    def _optimizer(self, optimizer_name: str, model: Module, learning_rate: float, weight_decay: float = 0):
        # reference base_trainer.py:34-43
        if optimizer_name.lower() == 'adam':
            return optim.Adam(model.parameters(), lr=learning_rate, weight_decay=weight_decay)
        elif optimizer_name.lower() == 'adamw':
            return optim.AdamW(model.parameters(), lr=learning_rate, weight_decay=weight_decay)
        elif optimizer_name.lower() == 'sgd':
            return optim.SGD(model.parameters(), lr=learning_rate, weight_decay=weight_decay)
        else:
            logger.error(f"Optimizer Not Exists: {optimizer_name}")
            raise NotImplementedError(f"Optimizer Not Exists: {optimizer_name}")

    def _loss(self, loss_name: str):
        # reference base_trainer.py:45-50
        if loss_name.lower() == 'bce':
            return torch.nn.BCELoss()
        else:
            logger.error(f"Loss Not Exists: {loss_name}")
            raise NotImplementedError(f"Loss Not Exists: {loss_name}")

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
        # reference base_trainer.py:117-141
        (valid_loss, valid_precision_at_k, valid_recall_at_k, valid_map_at_k, valid_ndcg_at_k) = metric['current']
        (best_valid_loss, best_valid_precision_at_k, best_valid_recall_at_k, best_valid_map_at_k,
         best_valid_ndcg_at_k) = metric['best']
        if self.cfg.best_metric == 'loss':
            return valid_loss < best_valid_loss
        elif self.cfg.best_metric == 'precision':
            return valid_precision_at_k > best_valid_precision_at_k
        elif self.cfg.best_metric == 'recall':
            return valid_recall_at_k > best_valid_recall_at_k
        elif self.cfg.best_metric == 'map':
            return valid_map_at_k > best_valid_map_at_k
        elif self.cfg.best_metric == 'ndcg':
            return valid_ndcg_at_k > best_valid_ndcg_at_k
        else:
            return False

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
