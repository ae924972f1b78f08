"""reference data/datasets/data_pipeline.py:4-23 — abstract base of the data pipelines."""
from abc import ABC, abstractmethod


class DataPipeline(ABC):
    def __init__(self, cfg):
        self.cfg = cfg

    @abstractmethod
    def split(self):
        pass

    @abstractmethod
    def preprocess(self):
        pass
