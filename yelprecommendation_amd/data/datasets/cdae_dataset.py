"""Per-user CDAE samples on the host — the counterpart of the reference's ``CDAEDataset``
(data/datasets/cdae_dataset.py:8-59): ``dataset[user_id]`` is a dict with the user's dense 0/1
``input_mask`` over the catalogue and, by mode, ``valid_mask`` / ``test_mask`` and a
``negative_mask`` of ``neg_times`` x positives items the user has not interacted with, drawn
WITHOUT replacement from the global NumPy RNG (one ``np.random.choice`` per fetched user, over the
ascending list of non-positive ids — the draw sequence of cdae_dataset.py:20-34).

For full-size runs use :mod:`..cdae_batches` instead: it keeps the interactions sparse on the device.
"""
import numpy as np
from torch.utils.data import Dataset

_HELD_OUT_KEY = {"valid": "valid_mask", "test": "test_mask"}


class CDAEDataset(Dataset):

    def __init__(self, data, mode='train', neg_times: int = 5):
        super().__init__()
        self.data, self.mode = data, mode
        if mode != 'test':                         # the test split draws no negatives
            self.neg_times = neg_times

    def __len__(self):
        return len(self.data)

    def _negative_sampling(self, seen):
        """0/1 vector marking ``neg_times * seen.sum()`` items out of those with ``seen == 0``."""
        candidates = np.flatnonzero(1 - seen)
        chosen = np.random.choice(candidates, int(seen.sum()) * self.neg_times, replace=False)
        mask = np.zeros_like(seen)
        mask[chosen] = 1.
        return mask

    def __getitem__(self, user_id):
        record = self.data[user_id]
        sample = {'user_id': user_id, 'input_mask': record['input_mask'].astype('float32')}
        held_out = _HELD_OUT_KEY.get(self.mode)
        if held_out is not None:
            sample[held_out] = record[held_out].astype('float32')
        if self.mode == 'train':
            sample['negative_mask'] = self._negative_sampling(sample['input_mask'])
        elif self.mode == 'valid':                 # negatives avoid the train AND the valid items
            sample['negative_mask'] = self._negative_sampling(sample['input_mask'] + sample['valid_mask'])
        return sample
