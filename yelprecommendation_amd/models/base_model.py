"""Drop-in for reference models/base_model.py:6-18 (same class, same hooks)."""
from abc import ABC, abstractmethod

import torch.nn as nn


class BaseModel(nn.Module, ABC):
    def __init__(self, *args, **kwargs) -> None:
        super().__init__(*args, **kwargs)

    def _activation_module(self, function_name: str) -> nn.Module:
        # reference models/base_model.py:10-14 (returns None for any other name)
        if function_name == 'sigmoid':
            return nn.Sigmoid()
        elif function_name == 'identity':
            return nn.Identity()

    @abstractmethod
    def _init_weights(self):
        pass
