"""MF data pipeline — drop-in for reference data/datasets/mf_data_pipeline.py:12-98.

``preprocess()`` reads ``<data_dir>/yelp_interactions.tsv`` (columns user_id, business_id, rating)
and sets ``num_users`` / ``num_items``; ``split(df)`` returns the reference's four frames
``(train_data, valid_data, valid_eval_data, test_eval_data)`` with the same columns, row order and
list contents.

The reference calls sklearn's ``train_test_split(user_df, test_size=.2, random_state=seed)`` and then
``(.., test_size=.25, random_state=seed)`` once per user (mf_data_pipeline.py:25-36, ~2.3 s per 30 k
rows).  With a fixed ``random_state`` the shuffle depends only on the group LENGTH:
``perm = RandomState(seed).permutation(n)``, test = perm[:ceil(test_size * n)], train = the rest — so
the split is restated here with one cached permutation per distinct length and NumPy indexing
(identical output, checked against a golden capture of the reference's split).
"""
import os

import numpy as np
import pandas as pd

from ...utils import logger
from .data_pipeline import DataPipeline


def _shuffle_split(n: int, test_size: float, seed: int):
    """sklearn ShuffleSplit(n_splits=1, test_size, random_state=seed) on n samples ->
    (train positions, test positions), as train_test_split uses it."""
    n_test = int(np.ceil(test_size * n))
    n_train = n - n_test
    if n_train <= 0:
        raise ValueError(f"With n_samples={n}, test_size={test_size} the resulting train set will be empty")
    perm = np.random.RandomState(seed).permutation(n)
    return perm[n_test:n_test + n_train], perm[:n_test]


class MFDataPipeline(DataPipeline):

    def __init__(self, cfg):
        super().__init__(cfg)
        self.num_items = None
        self.num_users = None

    def split(self, df):
        """df columns: user_id, business_id, rating (any row order)."""
        logger.info('start random user split...')
        if self.cfg.loss_name == 'pointwise':
            raise NotImplementedError("pointwise (stratified) split is outside the BPR path")
        seed = self.cfg.seed
        order = np.argsort(df['user_id'].values, kind='stable')          # groupby('user_id') order
        users_sorted = df['user_id'].values[order]
        bounds = np.flatnonzero(np.r_[True, users_sorted[1:] != users_sorted[:-1], True])
        cache = {}
        tr, va, te = [], [], []
        for a, b in zip(bounds[:-1], bounds[1:]):
            n = b - a
            if n not in cache:
                train1, test = _shuffle_split(n, .2, seed)
                train2, valid = _shuffle_split(len(train1), .25, seed)
                cache[n] = (train1[train2], train1[valid], test)
            t, v, s = cache[n]
            tr.append(order[a + t]); va.append(order[a + v]); te.append(order[a + s])
        frames = []
        for rows in (tr, va, te):
            rows = np.concatenate(rows) if rows else np.zeros(0, np.int64)
            part = df.iloc[rows].reset_index()                            # keeps the original index as 'index'
            frames.append(part)
        train_df, valid_df, test_df = frames

        def pos_lists(frame):
            # = frame.groupby('user_id').agg({'business_id': [('pos_items', list)]}).droplevel(0, 1)
            # (reference mf_data_pipeline.py:38-41) without pandas' per-group Python loop: users
            # ascending, row order kept inside each user's list, plain Python ints
            u, b = frame['user_id'].values, frame['business_id'].values
            by_user = np.argsort(u, kind='stable')
            us, bs = u[by_user], b[by_user]
            cut = np.flatnonzero(np.r_[True, us[1:] != us[:-1], True])
            lists = pd.Series([bs[i:j].tolist() for i, j in zip(cut[:-1], cut[1:])], dtype=object,
                              index=pd.Index(us[cut[:-1]], name='user_id'))
            return pd.DataFrame({'pos_items': lists})

        train_pos_df = pos_lists(train_df)
        valid_pos_df = pos_lists(valid_df)
        train_valid_pos_df = pos_lists(pd.concat([train_df, valid_df], axis=0))
        test_pos_df = pos_lists(test_df)

        train_data = pd.merge(train_df, train_pos_df, left_on='user_id', right_on='user_id', how='left')
        valid_data = pd.merge(valid_df, train_valid_pos_df, left_on='user_id', right_on='user_id', how='left')
        valid_eval_data = pd.merge(valid_pos_df, train_pos_df.rename(columns={'pos_items': 'mask_items'}),
                                   left_on='user_id', right_on='user_id', how='left')
        test_eval_data = pd.merge(test_pos_df, train_valid_pos_df.rename(columns={'pos_items': 'mask_items'}),
                                  left_on='user_id', right_on='user_id', how='left')
        return train_data, valid_data, valid_eval_data, test_eval_data

    def preprocess(self) -> pd.DataFrame:
        logger.info("start preprocessing...")
        df = self._load_df()
        self._set_num_items_and_num_users(df)
        if self.cfg.loss_name == 'pointwise':
            raise NotImplementedError("pointwise negative sampling is outside the BPR path")
        logger.info("done")
        return df

    def _load_df(self):
        logger.info("load df...")
        return pd.read_csv(os.path.join(self.cfg.data_dir, 'yelp_interactions.tsv'), sep='\t', index_col=False)

    def _set_num_items_and_num_users(self, df):
        self.num_items = df.business_id.nunique()
        self.num_users = df.user_id.nunique()
