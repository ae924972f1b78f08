"""Host logic of the device-side CDAE batches (pure torch ops: runs on CPU tensors here, on the GPU in
training): the sparse store reproduces the dense masks of the reference-style pipeline, the split
arithmetic is the reference's, and negative masks have the reference's law."""
import numpy as np
import pandas as pd
import torch

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


def test_sparse_store_reproduces_dense_pipeline_batches():
    df, (train_data, valid_data, test_data) = _pipeline_split()
    data = CDAEInteractions.from_split(train_data, valid_data, test_data)
    assert data.num_users == len(train_data) and data.num_items == len(train_data[0]["input_mask"])
    for mode, ref_ds in (("train", CDAEDataset(train_data, "train", 1)), ("valid", CDAEDataset(valid_data, "valid", 1)),
                         ("test", CDAEDataset(test_data, "test"))):
        loader = CDAEBatchLoader(data, mode, batch_size=16, neg_times=1)
        assert len(loader) == 4
        seen = 0
        for batch in loader:
            for r, u in enumerate(batch["user_id"].tolist()):
                want = ref_ds[u]
                np.testing.assert_array_equal(batch["input_mask"][r].numpy(), want["input_mask"])
                for key in ("valid_mask", "test_mask"):
                    if key in want:
                        np.testing.assert_array_equal(batch[key][r].numpy(), want[key])
                seen += 1
        assert seen == data.num_users


def test_negative_masks_have_the_reference_law():
    df, (train_data, valid_data, test_data) = _pipeline_split()
    data = CDAEInteractions.from_split(train_data, valid_data, test_data)
    for mode in ("train", "valid"):
        a = list(CDAEBatchLoader(data, mode, batch_size=25, neg_times=3, seed=1))
        b = list(CDAEBatchLoader(data, mode, batch_size=25, neg_times=3, seed=2))
        differ = False
        for x, y in zip(a, b):
            pos = x["input_mask"] + (x["valid_mask"] if mode == "valid" else 0)
            neg = x["negative_mask"]
            assert set(neg.unique().tolist()) <= {0.0, 1.0}
            assert float((neg * pos).sum()) == 0.0                        # never a positive (train + valid in valid mode)
            torch.testing.assert_close(neg.sum(1), 3 * pos.sum(1))         # exact count, no replacement
            differ |= not torch.equal(neg, y["negative_mask"])
        assert differ                                                      # the seed matters
    # uniformity: over many draws every non-positive item of a user is picked equally often
    pos = torch.zeros(1, 40); pos[0, :4] = 1
    loader = CDAEBatchLoader(data, "train", neg_times=2, seed=7)
    hits = sum(loader.negative_mask(pos) for _ in range(3000))[0]
    assert float(hits[:4].sum()) == 0.0
    freq = hits[4:] / 3000.0                                               # expected 8 / 36
    assert float((freq - 8 / 36).abs().max()) < 0.04
    # asking for more negatives than exist fails like np.random.choice(replace=False)
    crowded = torch.ones(1, 10); crowded[0, 0] = 0
    try:
        loader.negative_mask(crowded)
        raise AssertionError("expected ValueError")
    except ValueError:
        pass


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
        parts = [set(data.dense(p, rows)[0].nonzero().flatten().tolist()) for p in CDAEInteractions.PARTS]
        assert parts[0] | parts[1] | parts[2] == hist[user] and not (parts[0] & parts[1]) and not (parts[1] & parts[2])
        assert set(data.dense("train_valid", rows)[0].nonzero().flatten().tolist()) == parts[0] | parts[1]
    other = CDAEInteractions.from_interactions(u, i, nu, ni, seed=6)
    assert not torch.equal(other.dense("train", torch.arange(nu)), data.dense("train", torch.arange(nu)))
