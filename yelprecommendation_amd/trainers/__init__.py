from .base_trainer import BaseTrainer
from .mf_trainer import MFTrainer
from .ngcf_trainer import NGCFTrainer

__all__ = ["BaseTrainer", "MFTrainer", "NGCFTrainer"]
