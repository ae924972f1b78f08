"""Common base of the three models (the reference's ``BaseModel``, models/base_model.py:6-18).

Kept as an ``nn.Module`` so ``.to()``, ``.train()/.eval()``, ``.parameters()`` and
``state_dict()/load_state_dict()`` behave as the reference's trainers expect
(trainers/base_trainer.py:110,157).  The one helper the subclasses use, ``_activation_module``,
maps a config string to a module; an unknown name yields ``None`` exactly as in the reference
(models/base_model.py:10-14), and subclasses must provide ``_init_weights``.
"""
import abc

from torch import nn

_ACTIVATIONS = {"sigmoid": nn.Sigmoid, "identity": nn.Identity}


class BaseModel(nn.Module, abc.ABC):

    def _activation_module(self, function_name: str):
        make = _ACTIVATIONS.get(function_name)
        return make() if make is not None else None

    @abc.abstractmethod
    def _init_weights(self):
        """Initialise the parameters (called by the subclass constructor where the reference does)."""
