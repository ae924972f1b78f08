"""Device-side BPR triplet stream.

Counterpart of the reference's ``DataLoader(MFDataset, shuffle=True)``
(train.py:76, data/datasets/mf_dataset.py:18-32): one random permutation of the
train rows per epoch, and for each row a negative item drawn uniformly from
``[0, num_items)`` and redrawn while it is one of that user's positives
(``_negative_sampling``, mf_dataset.py:18-22).  The reference does this per row on the
host (~65 us/row); here a whole epoch is drawn at once with tensor ops on
the device the tables live on.  The RNG differs from the reference's NumPy stream by
construction, so parity tests REPLAY recorded streams instead (tests/replay.py);
this sampler feeds training runs and the benchmark.
"""
import torch


class TripletSampler:
    def __init__(self, user_id, item_id, num_users, num_items, pos_user=None, pos_item=None,
                 seed=0):
        """``user_id, item_id``: the rows to draw positives from (train or valid rows).
        ``pos_user, pos_item``: the (user, item) pairs negatives must avoid (defaults to
        the rows themselves: train positives for the train set; the reference passes
        train+valid positives for the valid set, mf_data_pipeline.py:47-48)."""
        self.device = user_id.device
        self.user, self.item = user_id.long(), item_id.long()
        self.num_users, self.num_items = int(num_users), int(num_items)
        pu = self.user if pos_user is None else pos_user.long().to(self.device)
        pi = self.item if pos_item is None else pos_item.long().to(self.device)
        self._keys = torch.unique(pu * self.num_items + pi)
        self._gen = torch.Generator(device=self.device).manual_seed(seed)

    def __len__(self):
        return self.user.numel()

    def _is_positive(self, u, i):
        k = u * self.num_items + i
        pos = torch.searchsorted(self._keys, k).clamp_(max=self._keys.numel() - 1)
        return self._keys[pos] == k

    def negatives(self, u):
        neg = torch.randint(0, self.num_items, u.shape, generator=self._gen, device=self.device)
        bad = self._is_positive(u, neg)
        while bool(bad.any()):
            idx = bad.nonzero(as_tuple=True)[0]
            redraw = torch.randint(0, self.num_items, idx.shape, generator=self._gen, device=self.device)
            neg[idx] = redraw
            bad = torch.zeros_like(bad)
            bad[idx] = self._is_positive(u[idx], redraw)
        return neg

    def epoch(self, shuffle=True):
        """(user, pos, neg) int64 tensors for one pass over the rows."""
        if shuffle:
            perm = torch.randperm(len(self), generator=self._gen, device=self.device)
            u, p = self.user[perm], self.item[perm]
        else:
            u, p = self.user, self.item
        return u, p, self.negatives(u)

    def stream(self, total, shuffle=True):
        """``total`` triplets: successive epochs concatenated and cut to length."""
        us, ps, ns, have = [], [], [], 0
        while have < total:
            u, p, n = self.epoch(shuffle)
            us.append(u); ps.append(p); ns.append(n)
            have += u.numel()
        return torch.cat(us)[:total], torch.cat(ps)[:total], torch.cat(ns)[:total]


class EpochLoader:
    """Iterable of batch dicts (the shape MFTrainer.train consumes) over one freshly
    sampled epoch per ``iter()`` — the fast path replacing DataLoader(MFDataset)."""

    def __init__(self, sampler: TripletSampler, batch_size: int, shuffle: bool = True):
        self.sampler, self.batch_size, self.shuffle = sampler, int(batch_size), shuffle

    def __len__(self):
        return (len(self.sampler) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        u, p, n = self.sampler.epoch(self.shuffle)
        for lo in range(0, u.numel(), self.batch_size):
            s = slice(lo, lo + self.batch_size)
            yield {"user_id": u[s], "pos_item": p[s], "neg_item": n[s]}


def split_train_rows(user_id, item_id, generator=None):
    """Per-user 60/20/20 random split with the reference's counts
    (mf_data_pipeline.py:32-33: test_size=.2 then .25 of the rest; sklearn rounds the
    test part UP): n_test = ceil(.2 n), n_valid = ceil(.25 (n - n_test)).
    Returns a label tensor (0 train / 1 valid / 2 test) aligned with the inputs, which
    must be sorted by user.  Device-side and vectorised; the exact sklearn permutation
    (needed only for reference parity) is restated in data/datasets/mf_data_pipeline.py."""
    dev = user_id.device
    n = user_id.numel()
    counts = torch.bincount(user_id)
    start = torch.cumsum(counts, 0) - counts
    # random order inside each user segment: sort by (user, random key)
    r = torch.rand(n, generator=generator, device=dev)
    order = torch.argsort(user_id.double() + r.double() * 0.999999)
    rank = torch.empty(n, dtype=torch.long, device=dev)
    rank[order] = torch.arange(n, device=dev) - start[user_id[order]]
    c = counts[user_id].double()
    n_test = torch.ceil(0.2 * c)
    n_valid = torch.ceil(0.25 * (c - n_test))
    label = torch.zeros(n, dtype=torch.long, device=dev)
    label[rank.double() < n_test] = 2
    label[(rank.double() >= n_test) & (rank.double() < n_test + n_valid)] = 1
    return label
