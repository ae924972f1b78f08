from .base_model import BaseModel
from .mf import MatrixFactorization
from .ngcf import NGCF

__all__ = ["BaseModel", "MatrixFactorization", "NGCF"]
