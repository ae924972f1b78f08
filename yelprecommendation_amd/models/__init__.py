from .base_model import BaseModel
from .cdae import CDAE
from .mf import MatrixFactorization
from .ngcf import NGCF

__all__ = ["BaseModel", "CDAE", "MatrixFactorization", "NGCF"]
