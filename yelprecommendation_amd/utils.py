"""Run utilities — counterpart of reference utils.py and of the config plumbing of
reference train.py:117-128 (hydra/omegaconf are not required).
"""
import logging
import os
import random
from functools import wraps
from typing import Callable

import numpy as np
import torch

logger = logging.getLogger("yelprecommendation_amd")


def set_seed(seed: int):
    """reference utils.py:14-24 — same calls in the same order, so a seeded run
    consumes the Python / NumPy / torch RNG streams exactly as the reference does."""
    logger.info(f"[utils] set seed as {seed}...")
    random.seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    torch.cuda.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)
    torch.backends.cudnn.deterministic = True


def log_metric(func: Callable):
    """reference utils.py:27-41 logs the four test metrics to wandb when a run is
    active; wandb is out of scope here, so the decorator logs and passes through."""
    @wraps(func)
    def wrapper(*args, **kwargs):
        precision_at_k, recall_at_k, map_at_k, ndcg_at_k = func(*args, **kwargs)
        logger.info(f"[Trainer] test P/R/MAP/NDCG = {precision_at_k:.4f} / {recall_at_k:.4f} / "
                    f"{map_at_k:.4f} / {ndcg_at_k:.4f}")
        return (precision_at_k, recall_at_k, map_at_k, ndcg_at_k)
    return wrapper


class Config(dict):
    """Attribute-style config with the keys of reference configs/train_config.yaml
    (what hydra's DictConfig gives the reference's trainers)."""
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__

    def copy(self):
        return Config(super().copy())


# reference configs/train_config.yaml:1-56 (run keys + the per-model blocks in scope)
DEFAULTS = Config(
    seed=42, shuffle=True, model_dir="outputs/models", submit_dir="outputs/submissions",
    data_dir="data/", log_dir="logs/", sweep=False, wandb=False,
    device="cuda", epochs=5, batch_size=32, lr=1e-4, optimizer="adam", loss_name="bpr",
    patience=5, top_n=10, weight_decay=0, best_metric="loss", model_name="MF",
    model=dict(
        CDAE=dict(negative_sampling=True, neg_times=5, hidden_size=64, corruption_level=0.6,
                  hidden_activation="sigmoid", output_activation="sigmoid"),
        MF=dict(embed_size=64),
        NGCF=dict(embed_size=64, num_orders=2),
    ),
)


def unpack_model(cfg) -> Config:
    """reference train.py:117-128: merge cfg.model[cfg.model_name] into the root."""
    if cfg["model_name"] not in cfg["model"]:
        raise ValueError(f"model '{cfg['model_name']}' is not defined in train_config.yaml")
    merged = Config({k: v for k, v in cfg.items() if k != "model"})
    merged.update(cfg["model"][cfg["model_name"]])
    return merged


def make_config(model_name="MF", **overrides) -> Config:
    cfg = Config(DEFAULTS)
    cfg["model_name"] = model_name
    merged = unpack_model(cfg)
    merged.update(overrides)
    return merged


def load_config(path: str, **overrides) -> Config:
    """Read a YAML file with the layout of reference configs/train_config.yaml."""
    import yaml
    with open(path) as f:
        raw = yaml.safe_load(f)
    cfg = Config(DEFAULTS)
    cfg.update(raw or {})
    if "model_name" in overrides:
        cfg["model_name"] = overrides.pop("model_name")
    merged = unpack_model(cfg)
    merged.update(overrides)
    return merged
