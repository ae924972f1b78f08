"""Host logic of the device-side CDAE batches: the sparse per-user store (index bookkeeping in torch, any
device) reproduces the dense masks of the reference-style pipeline, its split arithmetic is the
reference's, and the CPU statement of the negative-mask kernel (oracle/cdae_batches.py) has the
reference's law.  The HIP kernels themselves are compared with these definitions in test_gpu_cdae.py;
without a GPU the product loader refuses to build batches."""
import numpy as np
import pandas as pd
import torch

import pytest

from oracle import cdae_batches as ocb
from yelprecommendation_amd.data.cdae_batches import CDAEBatchLoader, CDAEInteractions
from yelprecommendation_amd.data.datasets.cdae_data_pipeline import CDAEDataPipeline
from yelprecommendation_amd.data.datasets.cdae_dataset import CDAEDataset
from yelprecommendation_amd.data.synthetic import make_frame
from yelprecommendation_amd.utils import make_config


def _pipeline_split(seed=3):
    cfg = make_config("CDAE", device="cpu", model_dir="/tmp/yr_cdae_batches")
    pipe = CDAEDataPipeline(cfg)
    df = make_frame(60, 90, 12.0)
    pipe._load_df = lambda: df
    frame = pipe.preprocess()
    np.random.seed(seed)
    return df, pipe.split(frame)


def _dense(data, part, users):
    if part == "train_valid":
        return torch.clamp(_dense(data, "train", users) + _dense(data, "valid", users), max=1.0)
    ptr, idx = data.csr(part)
    return ocb.dense_rows(ptr, idx, users, data.num_items)


def test_sparse_store_reproduces_dense_pipeline_masks():
    df, (train_data, valid_data, test_data) = _pipeline_split()
    data = CDAEInteractions.from_split(train_data, valid_data, test_data)
    assert data.num_users == len(train_data) and data.num_items == len(train_data[0]["input_mask"])
    users = torch.arange(data.num_users)
    dense = {p: _dense(data, p, users).numpy() for p in ("train", "valid", "test", "train_valid")}
    for u in range(data.num_users):
        np.testing.assert_array_equal(dense["train"][u], train_data[u]["input_mask"])
        np.testing.assert_array_equal(dense["valid"][u], valid_data[u]["valid_mask"])
        np.testing.assert_array_equal(dense["train_valid"][u], test_data[u]["input_mask"])
        np.testing.assert_array_equal(dense["test"][u], test_data[u]["test_mask"])
    # the merged train | valid CSR the test-time batches index (ids ascending inside a user)
    ptr, idx = data.csr("train_valid")
    np.testing.assert_array_equal(ocb.dense_rows(ptr, idx, users, data.num_items).numpy(), dense["train_valid"])
    assert all(bool((idx[ptr[u]:ptr[u + 1]][1:] > idx[ptr[u]:ptr[u + 1]][:-1]).all()) for u in range(data.num_users))
    # batches are built by HIP kernels only: a CPU store cannot be iterated
    from yelprecommendation_amd._lib import EngineError
    with pytest.raises(EngineError):
        next(iter(CDAEBatchLoader(data, "train", batch_size=16, neg_times=1)))


def test_negative_mask_definition_has_the_reference_law():
    df, (train_data, valid_data, test_data) = _pipeline_split()
    data = CDAEInteractions.from_split(train_data, valid_data, test_data)
    users = torch.arange(data.num_users)
    for pos in (_dense(data, "train", users), _dense(data, "train", users) + _dense(data, "valid", users)):
        a = ocb.negative_mask(pos, 3, torch.Generator().manual_seed(1))
        b = ocb.negative_mask(pos, 3, torch.Generator().manual_seed(2))
        assert set(a.unique().tolist()) <= {0.0, 1.0}
        assert float((a * pos).sum()) == 0.0                              # never a positive
        torch.testing.assert_close(a.sum(1), 3 * pos.sum(1))               # exact count, no replacement
        assert not torch.equal(a, b)                                       # the seed matters
    # uniformity: over many draws every non-positive item of a user is picked equally often
    pos = torch.zeros(1, 40); pos[0, :4] = 1
    gen = torch.Generator().manual_seed(7)
    hits = sum(ocb.negative_mask(pos, 2, gen) for _ in range(3000))[0]
    assert float(hits[:4].sum()) == 0.0
    freq = hits[4:] / 3000.0                                               # expected 8 / 36
    assert float((freq - 8 / 36).abs().max()) < 0.04
    # asking for more negatives than exist fails like np.random.choice(replace=False)
    crowded = torch.ones(1, 10); crowded[0, 0] = 0
    with pytest.raises(ValueError):
        ocb.negative_mask(crowded, 2)


def test_split_from_interactions_follows_reference_arithmetic():
    df, _ = _pipeline_split()
    u = torch.from_numpy(df.user_id.values.astype(np.int64)); i = torch.from_numpy(df.business_id.values.astype(np.int64))
    nu, ni = int(u.max()) + 1, int(i.max()) + 1
    data = CDAEInteractions.from_interactions(torch.cat([u, u[:50]]), torch.cat([i, i[:50]]), nu, ni, seed=5)  # duplicates collapse
    hist = pd.DataFrame({"u": u.numpy(), "i": i.numpy()}).drop_duplicates().groupby("u")["i"].apply(set)
    tr, va, te = (data.counts(p).numpy() for p in CDAEInteractions.PARTS)
    for user in range(nu):
        n = len(hist[user])
        n_tv = int(0.8 * n); n_tr = int(0.75 * n_tv)                       # cdae_data_pipeline.py:30-32
        assert (tr[user], va[user], te[user]) == (n_tr, n_tv - n_tr, n - n_tv)
        rows = torch.tensor([user])
        parts = [set(_dense(data, p, rows)[0].nonzero().flatten().tolist()) for p in CDAEInteractions.PARTS]
        assert parts[0] | parts[1] | parts[2] == hist[user] and not (parts[0] & parts[1]) and not (parts[1] & parts[2])
        assert set(_dense(data, "train_valid", rows)[0].nonzero().flatten().tolist()) == parts[0] | parts[1]
    other = CDAEInteractions.from_interactions(u, i, nu, ni, seed=6)
    assert not torch.equal(_dense(other, "train", torch.arange(nu)), _dense(data, "train", torch.arange(nu)))
