"""NGCF data pipeline — drop-in for reference data/datasets/ngcf_data_pipeline.py:13-49.

``preprocess()`` additionally builds ``self.laplacian_matrix`` (torch sparse COO, float32, on the
configured device) from the FULL interaction frame, as the reference does (:46-49) — but sparse
all the way: the reference goes through a dense (N, N) float32 array (19.4 GB at Yelp2018 size)
and hard-codes ``.to('cuda')``.
"""
import torch

from ...graph import LaplacianCSR
from ...utils import logger
from .mf_data_pipeline import MFDataPipeline


class NGCFDataPipeline(MFDataPipeline):

    def __init__(self, cfg):
        super().__init__(cfg)
        self.laplacian_matrix = None
        self.laplacian_csr = None

    def _set_laplacian_matrix(self, df):
        logger.info('set laplacian matrix...')
        device = torch.device(self.cfg.device if str(self.cfg.device) != 'cpu' else 'cpu')
        self.laplacian_csr = LaplacianCSR.from_interactions(
            df['user_id'].values, df['business_id'].values, df['rating'].values,
            self.num_users, self.num_items, device)
        self.laplacian_matrix = self.laplacian_csr.to_torch_sparse().to(device)
        logger.info('done...')

    def preprocess(self):
        df = super().preprocess()
        self._set_laplacian_matrix(df)
        return df
