from .base_trainer import BaseTrainer
from .mf_trainer import MFTrainer

__all__ = ["BaseTrainer", "MFTrainer"]
