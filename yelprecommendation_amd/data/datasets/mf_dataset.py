"""MF dataset — drop-in for reference data/datasets/mf_dataset.py:8-32, plus the fast path.

``MFDataset(data, num_items)[index]`` returns the reference's dict ``{'user_id', 'pos_item',
'neg_item'}`` with a negative drawn by the same rejection loop from the global NumPy RNG
(``np.random.randint(num_items)`` until it is not one of the row's ``pos_items``), so a
``DataLoader(MFDataset(...), shuffle=True)`` consumes the RNG streams as the reference does.

``to_sampler(device)`` hands the same rows to :class:`..triplets.TripletSampler`, which draws a whole
epoch on the device at once (the per-row path costs ~65 us/row in pandas).
"""
import numpy as np
from torch.utils.data import Dataset


class MFDataset(Dataset):

    def __init__(self, data, num_items=None):
        super().__init__()
        self.data = data
        self.num_items = num_items
        # column arrays once, instead of a pandas .iloc per row
        self._user = data['user_id'].values.astype('int64')
        self._item = data['business_id'].values.astype('int64')
        self._pos = data['pos_items'].values

    def __len__(self):
        return self.data.shape[0]

    def _negative_sampling(self, user_positives):
        # reference mf_dataset.py:18-22
        neg_item = np.random.randint(self.num_items)
        while neg_item in user_positives:
            neg_item = np.random.randint(self.num_items)
        return neg_item

    def __getitem__(self, index):
        # reference mf_dataset.py:24-32
        return {
            'user_id': self._user[index],
            'pos_item': self._item[index],
            'neg_item': self._negative_sampling(self._pos[index]),
        }

    def to_sampler(self, device, num_users, seed=0):
        """Device-side epoch sampler over the same rows; negatives avoid the same ``pos_items``."""
        import torch
        from ..triplets import TripletSampler
        u = torch.from_numpy(self._user).to(device)
        i = torch.from_numpy(self._item).to(device)
        # the avoid-set is the union of the rows' pos_items lists (train rows: train positives;
        # valid rows: train+valid positives — mf_data_pipeline.py:47-48)
        first = np.flatnonzero(np.r_[True, self._user[1:] != self._user[:-1]])
        pu = np.concatenate([np.full(len(self._pos[k]), self._user[k], dtype=np.int64) for k in first])
        pi = np.concatenate([np.asarray(self._pos[k], dtype=np.int64) for k in first])
        return TripletSampler(u, i, num_users, self.num_items, torch.from_numpy(pu).to(device),
                              torch.from_numpy(pi).to(device), seed=seed)
