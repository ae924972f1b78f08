from .base_trainer import BaseTrainer
from .cdae_trainer import CDAETrainer
from .mf_trainer import MFTrainer
from .ngcf_trainer import NGCFTrainer

__all__ = ["BaseTrainer", "CDAETrainer", "MFTrainer", "NGCFTrainer"]
