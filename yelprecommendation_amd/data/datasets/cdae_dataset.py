"""CDAE dataset — drop-in for reference data/datasets/cdae_dataset.py:8-59 (dense 0/1 masks per
user, negative mask = ``neg_times`` x positives drawn without replacement from the global NumPy RNG)."""
import numpy as np
from torch.utils.data import Dataset


class CDAEDataset(Dataset):

    def __init__(self, data, mode='train', neg_times: int = 5):
        super().__init__()
        self.data = data
        self.mode = mode
        if self.mode != 'test':
            self.neg_times = neg_times

    def __len__(self):
        return len(self.data.keys())

    def _negative_sampling(self, input_mask):
        # reference cdae_dataset.py:20-34
        num_pos = int(input_mask.sum())
        negative_indexes = (1 - input_mask).nonzero()[0]
        negative_samples = np.random.choice(negative_indexes, num_pos * self.neg_times, replace=False)
        negative_mask = np.zeros_like(input_mask)
        negative_mask[negative_samples] = 1.
        return negative_mask

    def __getitem__(self, user_id):
        # reference cdae_dataset.py:36-59
        input_mask = self.data[user_id]['input_mask'].astype('float32')
        if self.mode == 'train':
            return {'user_id': user_id, 'input_mask': input_mask,
                    'negative_mask': self._negative_sampling(input_mask)}
        elif self.mode == 'valid':
            valid_mask = self.data[user_id]['valid_mask'].astype('float32')
            return {'user_id': user_id, 'input_mask': input_mask, 'valid_mask': valid_mask,
                    'negative_mask': self._negative_sampling(input_mask + valid_mask)}
        else:
            test_mask = self.data[user_id]['test_mask'].astype('float32')
            return {'user_id': user_id, 'input_mask': input_mask, 'test_mask': test_mask}
