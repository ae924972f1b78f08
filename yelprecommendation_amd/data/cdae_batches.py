"""Device-side CDAE batches (the CDAE counterpart of data/triplets.py).

The reference keeps, per user, dense 0/1 vectors over the whole catalogue on the host
(data/datasets/cdae_data_pipeline.py:22-48: four int32 masks per user — 19 GB at Yelp2018 size) and
builds every batch from them in ``CDAEDataset.__getitem__`` (cdae_dataset.py:36-59), negative mask
included (``np.random.choice(non-positives, neg_times * positives, replace=False)``, :20-34).  Each
[B, I] float batch then crosses PCIe (39 MB per 256 users) — two orders of magnitude more time than
the training step on MI355X.

Here the interactions stay sparse (per-user train / valid / test item lists as CSR on the device) and
the dense rows of a batch are materialised on the device right before the step, with the same keys
the reference's DataLoader yields (``user_id, input_mask, negative_mask | valid_mask | test_mask``).
Negative masks have the reference's law — exactly ``neg_times * positives`` distinct non-positive
items per user, every subset equally likely (random keys + per-row order statistic) — but from the
device RNG, so parity tests replay recorded masks instead (tests/test_gpu_cdae.py).

Both pieces are HIP kernels (``yr_csr_rows_to_dense``, ``yr_negative_mask``: csrc/cdae_batches.hip);
the store may be built on any device (index bookkeeping in torch), but batches exist on the GPU only —
on CPU tensors the engine raises.  The CPU statement of the two kernels lives in oracle/cdae_batches.py.
"""
import torch

from .. import engine


def _csr_from_pairs(rows, cols, num_rows):
    order = torch.argsort(rows * (int(cols.max()) + 1 if cols.numel() else 1) + cols)
    rows, cols = rows[order], cols[order]
    ptr = torch.zeros(num_rows + 1, dtype=torch.int64, device=rows.device)
    ptr[1:] = torch.cumsum(torch.bincount(rows, minlength=num_rows), 0)
    return ptr, cols.contiguous()


class CDAEInteractions:
    """Per-user train / valid / test item lists (CSR, item ids ascending inside a user)."""

    PARTS = ("train", "valid", "test")

    def __init__(self, num_users, num_items, parts, device):
        self.num_users, self.num_items, self.device = int(num_users), int(num_items), torch.device(device)
        self._csr = {k: (p.to(self.device), i.to(self.device)) for k, (p, i) in parts.items()}

    # -- constructors -----------------------------------------------------------------------
    @classmethod
    def from_split(cls, train_data, valid_data, test_data, device="cpu"):
        """From the dicts ``CDAEDataPipeline.split`` returns (dense masks per user): the same
        split, stored sparsely.  Users are the dict keys, assumed dense 0..U-1 like the
        reference's ``nn.Embedding(num_users)`` lookup (models/cdae.py:49)."""
        import numpy as np
        users = sorted(train_data.keys())
        num_items = len(next(iter(train_data.values()))["input_mask"])
        parts = {}
        for name, get in (("train", lambda u: train_data[u]["input_mask"]),
                          ("valid", lambda u: valid_data[u]["valid_mask"]),
                          ("test", lambda u: test_data[u]["test_mask"])):
            ptr = np.zeros(len(users) + 1, dtype=np.int64)
            idx = []
            for k, u in enumerate(users):
                nz = np.flatnonzero(get(u))
                idx.append(nz)
                ptr[k + 1] = ptr[k] + len(nz)
            parts[name] = (torch.from_numpy(ptr), torch.from_numpy(np.concatenate(idx).astype(np.int64)))
        return cls(len(users), num_items, parts, device)

    @classmethod
    def from_interactions(cls, user, item, num_users, num_items, seed=0, device=None):
        """Straight from (user, item) pairs: each user's history is shuffled and cut 60/20/20 with
        the reference's arithmetic (cdae_data_pipeline.py:30-32: ``int(0.8 n)`` train+valid,
        of which ``int(0.75 .)`` train)."""
        device = torch.device(device if device is not None else user.device)
        user, item = user.to(device).long(), item.to(device).long()
        key = torch.unique(user * num_items + item)                       # the pivot de-duplicates pairs
        user, item = torch.div(key, num_items, rounding_mode="floor"), key % num_items
        gen = torch.Generator(device=device).manual_seed(seed)
        r = torch.rand(user.numel(), generator=gen, device=device)
        order = torch.argsort(user.double() + r.double() * 0.999999)      # by user, random inside a user
        user, item = user[order], item[order]
        counts = torch.bincount(user, minlength=num_users)
        start = torch.cumsum(counts, 0) - counts
        pos = torch.arange(user.numel(), device=device) - start[user]     # rank inside the shuffled history
        n = counts[user]
        n_tv = torch.floor(0.8 * n.double()).long()                       # int(0.8 * len)
        n_tr = torch.floor(0.75 * n_tv.double()).long()                   # int(0.75 * len(train_samples))
        label = torch.where(pos < n_tr, 0, torch.where(pos < n_tv, 1, 2))
        parts = {}
        for k, name in enumerate(cls.PARTS):
            m = label == k
            parts[name] = _csr_from_pairs(user[m], item[m], num_users)
        return cls(num_users, num_items, parts, device)

    # -- access -------------------------------------------------------------------------------
    def csr(self, part):
        """(ptr, idx) of ``part``; 'train_valid' (train | valid per user, ids ascending) is merged on
        first use."""
        if part == "train_valid" and part not in self._csr:
            pt, it = self._csr["train"]
            pv, iv = self._csr["valid"]
            users = torch.arange(self.num_users, device=self.device)
            rows = torch.cat([torch.repeat_interleave(users, pt[1:] - pt[:-1]),
                              torch.repeat_interleave(users, pv[1:] - pv[:-1])])
            self._csr[part] = _csr_from_pairs(rows, torch.cat([it, iv]), self.num_users)
        return self._csr[part]

    def counts(self, part):
        ptr, _ = self._csr[part]
        return ptr[1:] - ptr[:-1]

    def dense(self, part, users):
        """[len(users), num_items] float32 0/1 rows of ``part`` ('train', 'valid', 'test' or
        'train_valid' = train | valid, the test-time input of cdae_data_pipeline.py:38)."""
        out = None
        for p in (("train", "valid") if part == "train_valid" else (part,)):
            ptr, idx = self._csr[p]
            out = engine.csr_rows_to_dense(ptr, idx, users.contiguous(), self.num_items, out=out,
                                           accumulate=out is not None)
        return out


class CDAEBatchLoader:
    """Iterable of batches with the reference DataLoader's keys, tensors already on the device.

    mode 'train': user_id, input_mask (train items), negative_mask
    mode 'valid': user_id, input_mask (train items), valid_mask, negative_mask (avoids train + valid)
    mode 'test' : user_id, input_mask (train + valid items), test_mask
    """

    def __init__(self, data: CDAEInteractions, mode="train", batch_size=32, neg_times=5, shuffle=False, seed=0,
                 lists=False, dropout=0.0):
        """``lists=True``: a batch is ``{"user_id", "lists"[, "item_lists"]}`` with ``lists`` an engine.TrainLists —
        the encoder's input and the NS-BCE positions (positives + ``neg_times`` x as many sampled negatives) as
        per-row lists made straight from the CSR (yr_cdae_train_lists); no dense row or mask exists.
        'train': input = train items after nn.Dropout(``dropout``), positives = train items (CDAETrainer.train feeds
        the fused step); 'valid': input = train items, positives = train + held-out items; 'test': input = train +
        valid items, no loss list.  CDAETrainer.validate / evaluate then score all users at once with the fused
        evaluation kernel."""
        if mode not in ("train", "valid", "test"):
            raise ValueError(f"mode {mode!r}")
        self.data, self.mode, self.batch_size, self.neg_times, self.shuffle = data, mode, int(batch_size), neg_times, shuffle
        self.lists, self.dropout = bool(lists), float(dropout)
        self._gen = torch.Generator(device=data.device).manual_seed(seed)
        self._seeds = torch.Generator().manual_seed(seed)          # host generator: per-batch kernel seeds
        self._flag = engine.new_error_flag(data.device) if data.device.type == "cuda" else None   # cpu store: iterating raises
        self._list_pool = {}                                       # this loader's TrainLists storage (one batch alive at a time)

    def __len__(self):
        return (self.data.num_users + self.batch_size - 1) // self.batch_size

    def negative_mask(self, positives):
        """Exactly ``neg_times * positives`` distinct non-positive items per row, uniformly: the
        items with the smallest of I i.i.d. random keys among the non-positives."""
        seed = int(torch.randint(0, 1 << 62, (1,), generator=self._seeds).item())
        return engine.negative_mask(positives.contiguous(), self.neg_times, seed, err_flag=self._flag)

    def super_batches(self, group=8):
        """lists=True, mode 'valid' / 'test': the same batches, ``group`` at a time — ``{"user_id" (all rows of the
        group), "lists" (one engine.TrainLists over them, made by ONE launch with the seeds of the single batches),
        "batch_rows", "item_lists"}``.  The host generators are consumed exactly as by ``__iter__`` (one permutation,
        two seeds per batch), so the lists — and every number derived from them — are those of the per-batch
        iteration; CDAETrainer.validate / evaluate then need a handful of launches per group instead of ~10 per
        batch."""
        if not self.lists or self.mode == "train":
            raise ValueError("super_batches: list batches of the 'valid' / 'test' loaders")
        d = self.data
        order = (torch.randperm(d.num_users, generator=self._gen, device=d.device) if self.shuffle
                 else torch.arange(d.num_users, device=d.device))
        bs = self.batch_size
        lists = {"users": None, "actual": d.csr("test" if self.mode == "test" else "valid"),
                 "seen": d.csr("train_valid" if self.mode == "test" else "train")}
        for s in range(0, d.num_users, bs * group):
            users = order[s:s + bs * group].contiguous()
            nb = -(-users.numel() // bs)
            seeds = torch.randint(0, 1 << 62, (nb, 2), generator=self._seeds)      # the draws of nb single batches
            seeds = seeds.to(d.device)
            ptr, idx = d.csr("train_valid" if self.mode == "test" else "train")
            made = engine.TrainLists(ptr, idx, users, d.num_users, d.num_items, 0 if self.mode == "test" else self.neg_times,
                                     seeds[:, 0].contiguous(), seeds[:, 1].contiguous(), 0.0, err_flag=self._flag,
                                     extra=d.csr("valid") if self.mode == "valid" else None, pool=self._list_pool,
                                     batch_rows=bs)
            yield {"user_id": users, "lists": made, "batch_rows": bs, "item_lists": dict(lists, users=users)}
        if self._flag is not None and int(self._flag.item()):
            self._flag.zero_()
            raise ValueError("Cannot take a larger sample than population when 'replace=False'")

    def __iter__(self):
        d = self.data
        order = (torch.randperm(d.num_users, generator=self._gen, device=d.device) if self.shuffle
                 else torch.arange(d.num_users, device=d.device))
        for s in range(0, d.num_users, self.batch_size):
            users = order[s:s + self.batch_size]
            # beside the reference's keys: where the held-out and the seen items of these users live
            # in the per-user CSR, so that evaluation needs no nonzero() over the dense masks
            lists = {"users": users}
            if self.mode == "valid":
                lists.update(actual=d.csr("valid"), seen=d.csr("train"))
            elif self.mode == "test":
                lists.update(actual=d.csr("test"), seen=d.csr("train_valid"))
            if self.lists:
                seeds = torch.randint(0, 1 << 62, (2,), generator=self._seeds).tolist()
                ptr, idx = d.csr("train_valid" if self.mode == "test" else "train")
                made = engine.TrainLists(ptr, idx, users.contiguous(), d.num_users, d.num_items,
                                         0 if self.mode == "test" else self.neg_times, seeds[0], seeds[1],
                                         self.dropout if self.mode == "train" else 0.0, err_flag=self._flag,
                                         extra=d.csr("valid") if self.mode == "valid" else None, pool=self._list_pool)
                batch = {"user_id": users, "lists": made}
                if self.mode != "train":
                    batch["item_lists"] = lists
                yield batch
            elif self.mode == "train":
                x = d.dense("train", users)
                yield {"user_id": users, "input_mask": x, "negative_mask": self.negative_mask(x)}
            elif self.mode == "valid":
                x, v = d.dense("train", users), d.dense("valid", users)
                yield {"user_id": users, "input_mask": x, "valid_mask": v, "negative_mask": self.negative_mask(x + v),
                       "item_lists": lists}
            else:
                yield {"user_id": users, "input_mask": d.dense("train_valid", users), "test_mask": d.dense("test", users),
                       "item_lists": lists}
        if self._flag is not None and int(self._flag.item()):
            # a row asked for more negatives than it has non-positives: np.random.choice(replace=False)
            # raises in the reference (cdae_dataset.py:27); here the check is one read per epoch
            self._flag.zero_()
            raise ValueError("Cannot take a larger sample than population when 'replace=False'")
