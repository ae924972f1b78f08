"""Device-side BPR triplet stream — csrc/triplets.hip through the C ABI.

Counterpart of the reference's ``DataLoader(MFDataset, shuffle=True)``
(train.py:76, data/datasets/mf_dataset.py:18-32): one random permutation of the
train rows per epoch, and for each row a negative item drawn uniformly from
``[0, num_items)`` and redrawn while it is one of that user's positives
(``_negative_sampling``, mf_dataset.py:18-22).  The reference does this per row on the
host (~65 us/row); here ONE kernel launch produces any slice of an epoch's stream on the
device the tables live on: position t of epoch e is a pure function of (seed, e, t) (keyed Feistel
permutation + Philox rejection sampling against the per-user avoid CSR), so there is no state, no
host synchronisation and a rank can produce exactly its share.  The RNG differs from the reference's
NumPy stream by construction: parity runs REPLAY recorded streams (:class:`RecordedStream`); this
sampler feeds training runs and the benchmark.  oracle/triplet_sampler.py states the same words on the CPU.
"""
import torch

from .. import engine


class TripletSampler:
    def __init__(self, user_id, item_id, num_users, num_items, pos_user=None, pos_item=None,
                 seed=0):
        """``user_id, item_id``: the rows to draw positives from (train or valid rows).
        ``pos_user, pos_item``: the (user, item) pairs negatives must avoid (defaults to
        the rows themselves: train positives for the train set; the reference passes
        train+valid positives for the valid set, mf_data_pipeline.py:47-48)."""
        self.device = user_id.device
        if not user_id.is_cuda:
            raise engine.EngineError("TripletSampler runs on the GPU only (csrc/triplets.hip); parity runs "
                                     "replay a recorded stream with RecordedStream")
        self.user, self.item = user_id.long().contiguous(), item_id.long().contiguous()
        self.num_users, self.num_items = int(num_users), int(num_items)
        pu = self.user if pos_user is None else pos_user.long().to(self.device)
        pi = self.item if pos_item is None else pos_item.long().to(self.device)
        # per-user avoid lists as CSR, ascending and duplicate-free inside a user (one-off set-up)
        keys = torch.unique(pu * self.num_items + pi)
        owner = torch.div(keys, self.num_items, rounding_mode="floor")
        self.avoid_idx = (keys - owner * self.num_items).contiguous()
        self.avoid_ptr = torch.zeros(self.num_users + 1, dtype=torch.int64, device=self.device)
        self.avoid_ptr[1:] = torch.cumsum(torch.bincount(owner, minlength=self.num_users), 0)
        self.seed, self.epoch_no = int(seed), 0
        self.flag = engine.new_error_flag(self.device)

    def __len__(self):
        return self.user.numel()

    def draw(self, epoch, first=0, count=None, shuffle=True):
        """(user, pos, neg) int64 tensors for stream positions [first, first + count) of epoch ``epoch``."""
        count = len(self) - first if count is None else count
        return engine.triplet_sample(self.user, self.item, self.avoid_ptr, self.avoid_idx, self.num_users,
                                     self.num_items, self.seed, epoch, shuffle, first, count, err_flag=self.flag)

    def epoch(self, shuffle=True):
        """(user, pos, neg) for one pass over the rows; successive calls are successive epochs."""
        out = self.draw(self.epoch_no, shuffle=shuffle)
        self.epoch_no += 1
        return out

    def stream(self, total, shuffle=True):
        """``total`` triplets: successive epochs concatenated and cut to length."""
        us, ps, ns, have = [], [], [], 0
        while have < total:
            take = min(len(self), total - have)
            u, p, n = self.draw(self.epoch_no, 0, take, shuffle)
            self.epoch_no += 1
            us.append(u); ps.append(p); ns.append(n)
            have += take
        return torch.cat(us), torch.cat(ps), torch.cat(ns)

    def check(self):
        engine.raise_on_flag(self.flag, "TripletSampler")


class RecordedStream:
    """Replay mode: yields the recorded batches of ONE epoch as the reference's DataLoader would (dict of
    int64 tensors), so that a trainer can be driven by a golden triplet stream ("identical (seeded)
    negative samples" of the parity protocol)."""

    def __init__(self, u, p, n, batch_sizes, device=None):
        import numpy as np
        conv = lambda a: torch.from_numpy(np.ascontiguousarray(a).astype(np.int64)) if not torch.is_tensor(a) else a.long()
        self.u, self.p, self.n = (conv(a) if device is None else conv(a).to(device) for a in (u, p, n))
        self.sizes = [int(b) for b in batch_sizes]

    def __iter__(self):
        pos = 0
        for b in self.sizes:
            s = slice(pos, pos + b)
            yield {"user_id": self.u[s], "pos_item": self.p[s], "neg_item": self.n[s]}
            pos += b

    def __len__(self):
        return len(self.sizes)


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
