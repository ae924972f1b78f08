"""CDAE data pipeline — drop-in for reference data/datasets/cdae_data_pipeline.py:9-90.

``preprocess()`` pivots the interactions into the dense binary user x item frame the reference
builds (:78-90); ``split(df)`` shuffles each user's history with the global NumPy RNG (unseeded in
the reference: train.py:154 runs before set_seed at :57) and cuts it 60/20/20 (:30-32)."""
import os

import numpy as np
import pandas as pd

from ...utils import logger
from .data_pipeline import DataPipeline


class CDAEDataPipeline(DataPipeline):

    def __init__(self, cfg):
        super().__init__(cfg)

    def split(self, df):
        logger.info('start random user split...')
        train_data, valid_data, test_data = {}, {}, {}
        values = df.values
        for row in values:
            user_id = int(row[0])
            hist = row[1:]
            user_history = np.argwhere(hist).reshape(-1)
            np.random.shuffle(user_history)
            train_samples, test_samples = np.split(user_history, [int(0.8 * len(user_history))])
            train_samples, valid_samples = np.split(train_samples, [int(0.75 * len(train_samples))])
            masks = []
            for samples in (train_samples, valid_samples, test_samples, np.union1d(train_samples, valid_samples)):
                m = np.zeros(hist.shape[0], dtype=np.int32)
                m[samples] = 1
                masks.append(m)
            train_mask, valid_mask, test_mask, train_valid_mask = masks
            train_data[user_id] = {'input_mask': train_mask}
            valid_data[user_id] = {'input_mask': train_mask, 'valid_mask': valid_mask}
            test_data[user_id] = {'input_mask': train_valid_mask, 'test_mask': test_mask}
        logger.info("done")
        return train_data, valid_data, test_data

    def preprocess(self) -> pd.DataFrame:
        logger.info("start preprocessing...")
        df = self._load_df()
        training_set = self._transform_into_training_set(df)
        logger.info("done")
        return training_set

    def _load_df(self):
        logger.info("load df...")
        return pd.read_csv(os.path.join(self.cfg.data_dir, 'yelp_interactions.tsv'), sep='\t', index_col=False)

    def _transform_into_training_set(self, df):
        # reference cdae_data_pipeline.py:78-90
        logger.info("transform df into training set...")
        item_inputs = df.pivot_table(index='user_id', columns=['business_id'], values=['rating'])
        training_set = item_inputs.droplevel(0, 1)
        training_set = training_set.fillna(0)
        training_set = training_set.mask(training_set > 0, 1)
        training_set = training_set.reset_index()
        training_set.index.name = None
        return training_set
