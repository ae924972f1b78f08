"""Interface of the three data pipelines (the reference's ``DataPipeline``,
data/datasets/data_pipeline.py:4-23): they are built from the run config, ``preprocess()`` yields
the frame a model family works on and ``split()`` cuts it into the train / valid / test parts."""
import abc


class DataPipeline(abc.ABC):

    def __init__(self, cfg):
        self.cfg = cfg

    @abc.abstractmethod
    def preprocess(self):
        """Load the interactions and bring them into the family's input form."""

    @abc.abstractmethod
    def split(self):
        """Cut the preprocessed data into train / valid / test parts."""
