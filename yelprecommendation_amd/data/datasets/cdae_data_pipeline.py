"""CDAE data pipeline — drop-in for reference data/datasets/cdae_data_pipeline.py:9-90.

``preprocess()`` pivots the interactions into the dense binary user x item frame the reference
builds (:78-90); ``split(df)`` shuffles each user's history with the global NumPy RNG (unseeded in
the reference: train.py:154 runs before set_seed at :57) and cuts it 60/20/20 (:30-32)."""
import os

import numpy as np
import pandas as pd

from ...utils import logger
from .data_pipeline import DataPipeline


def _cut(history):
    """train / valid / test parts of one (already shuffled) history: int(0.8 n) go to train+valid, of
    which int(0.75 .) to train (cdae_data_pipeline.py:30-32)."""
    n_train_valid = int(0.8 * len(history))
    n_train = int(0.75 * n_train_valid)
    return history[:n_train], history[n_train:n_train_valid], history[n_train_valid:]


def _mask(num_items, ids):
    m = np.zeros(num_items, dtype=np.int32)
    m[ids] = 1
    return m


class CDAEDataPipeline(DataPipeline):

    def split(self, df):
        """``df``: the frame of :meth:`preprocess` (column 0 = user id, then one 0/1 column per item).
        Returns three dicts keyed by user id (train / valid / test), each value holding the dense
        masks the matching :class:`CDAEDataset` mode reads."""
        logger.info('start random user split...')
        table = df.values
        num_items = table.shape[1] - 1
        train_data, valid_data, test_data = {}, {}, {}
        for row in table:
            user = int(row[0])
            history = np.flatnonzero(row[1:])
            np.random.shuffle(history)               # one global-RNG shuffle per user, in frame order
            train_ids, valid_ids, test_ids = _cut(history)
            seen_in_training = _mask(num_items, train_ids)
            train_data[user] = {'input_mask': seen_in_training}
            valid_data[user] = {'input_mask': seen_in_training, 'valid_mask': _mask(num_items, valid_ids)}
            test_data[user] = {'input_mask': _mask(num_items, np.concatenate([train_ids, valid_ids])),
                               'test_mask': _mask(num_items, test_ids)}
        logger.info("done")
        return train_data, valid_data, test_data

    def preprocess(self) -> pd.DataFrame:
        logger.info("start preprocessing...")
        wide = self._transform_into_training_set(self._load_df())
        logger.info("done")
        return wide

    def _load_df(self):
        logger.info("load df...")
        path = os.path.join(self.cfg.data_dir, 'yelp_interactions.tsv')
        return pd.read_csv(path, sep='\t', index_col=False)

    def _transform_into_training_set(self, df):
        """users x items 0/1 frame with the user id as first column (cdae_data_pipeline.py:78-90):
        pivot on the rating, missing -> 0, any rating -> 1."""
        logger.info("transform df into training set...")
        wide = df.pivot_table(index='user_id', columns=['business_id'], values=['rating']).droplevel(0, axis=1).fillna(0)
        wide = wide.mask(wide > 0, 1).reset_index()
        wide.index.name = None
        return wide
