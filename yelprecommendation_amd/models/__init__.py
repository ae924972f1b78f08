from .base_model import BaseModel
from .mf import MatrixFactorization

__all__ = ["BaseModel", "MatrixFactorization"]
