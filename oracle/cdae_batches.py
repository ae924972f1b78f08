"""CPU statement of what the device-side CDAE batch kernels compute (test infrastructure only:
imported by tests/, never by the product).

* :func:`dense_rows`     — 0/1 rows of a per-user item CSR: the dense masks the reference keeps per user
  (data/datasets/cdae_data_pipeline.py:33-37) for a batch of users.  Checked against the reference-style
  pipeline's masks in tests/test_cdae_batches.py.
* :func:`negative_mask`  — the law of ``CDAEDataset._negative_sampling`` (data/datasets/cdae_dataset.py:20-34):
  exactly ``neg_times * positives`` distinct non-positive items per row, every subset equally likely
  (``np.random.choice(non_positives, k, replace=False)``), as i.i.d. float64 keys + a per-row order
  statistic.  The RNG stream is not the reference's (nor the kernel's): the tests check the law —
  count, disjointness, uniformity — on both.
"""
import torch


def dense_rows(ptr, idx, users, num_items):
    out = torch.zeros((users.numel(), num_items), dtype=torch.float32)
    lo, cnt = ptr[users], ptr[users + 1] - ptr[users]
    rows = torch.repeat_interleave(torch.arange(users.numel()), cnt)
    offs = torch.arange(rows.numel()) - torch.repeat_interleave(torch.cumsum(cnt, 0) - cnt, cnt)
    out[rows, idx[lo[rows] + offs]] = 1.0
    return out


def negative_mask(positives, neg_times, generator=None):
    n = (positives.sum(dim=1) * neg_times).long()
    room = positives.shape[1] - positives.sum(dim=1).long()
    if bool((n > room).any()):
        # np.random.choice(..., replace=False) raises in the reference (cdae_dataset.py:27)
        raise ValueError("Cannot take a larger sample than population when 'replace=False'")
    kmax = int(n.max()) if n.numel() else 0
    if kmax == 0:
        return torch.zeros_like(positives)
    # float64 keys: with float32 a row of 38k keys holds dozens of ties, and a tie AT the threshold
    # would break the exact count
    keys = torch.rand(positives.shape, generator=generator, dtype=torch.float64)
    keys = torch.where(positives > 0, torch.full_like(keys, 2.0), keys)
    smallest = torch.topk(keys, kmax, dim=1, largest=False, sorted=True).values
    thr = smallest.gather(1, (n - 1).clamp_(min=0)[:, None])
    return ((keys <= thr) & (n[:, None] > 0)).float()
