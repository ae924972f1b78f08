"""CPU: the MF/NGCF data pipelines against the reference's own split() output (golden capture)."""
import os

import numpy as np
import pandas as pd
import pytest


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "mf_small.npz"))


def _cfg(seed=42):
    from yelprecommendation_amd.utils import make_config
    return make_config("MF", seed=seed, device="cpu", loss_name="bpr")


def test_split_reproduces_reference(g):
    from yelprecommendation_amd.data.datasets.mf_data_pipeline import MFDataPipeline
    df = pd.DataFrame({"user_id": g["tsv_user"].astype(np.int64), "business_id": g["tsv_item"].astype(np.int64),
                       "rating": g["tsv_rating"].astype(np.int64)})
    pipe = MFDataPipeline(_cfg())
    pipe._set_num_items_and_num_users(df)
    assert (pipe.num_users, pipe.num_items) == (int(g["num_users"]), int(g["num_items"]))
    train, valid, valid_eval, test_eval = pipe.split(df)
    assert list(train.columns) == ["index", "user_id", "business_id", "rating", "pos_items"]
    np.testing.assert_array_equal(train["user_id"].values, g["train_user"])
    np.testing.assert_array_equal(train["business_id"].values, g["train_item"])
    np.testing.assert_array_equal(train["index"].values, g["train_index"])
    np.testing.assert_array_equal(valid["user_id"].values, g["valid_user"])
    np.testing.assert_array_equal(valid["business_id"].values, g["valid_item"])
    np.testing.assert_array_equal(valid["index"].values, g["valid_index"])

    def csr(col):
        ptr = np.zeros(len(col) + 1, np.int64)
        ptr[1:] = np.cumsum([len(x) for x in col])
        return ptr, np.concatenate([np.asarray(x) for x in col])

    for frame, name in ((valid_eval, "valid"), (test_eval, "test")):
        np.testing.assert_array_equal(frame.index.values, g[f"{name}_eval_users"])
        for colname, key in (("pos_items", "pos"), ("mask_items", "mask")):
            ptr, idx = csr(frame[colname])
            np.testing.assert_array_equal(ptr, g[f"{name}_{key}_ptr"])
            np.testing.assert_array_equal(idx, g[f"{name}_{key}_idx"])
    ptr, idx = csr(train.groupby("user_id")["pos_items"].first())
    np.testing.assert_array_equal(ptr, g["train_pos_ptr"])
    np.testing.assert_array_equal(idx, g["train_pos_idx"])


def test_dataset_replays_reference_negative_stream(g):
    """MFDataset + torch DataLoader(shuffle=True) after set_seed() yields the reference's recorded
    first-epoch triplet stream (same torch permutation, same NumPy rejection draws)."""
    import torch
    from torch.utils.data import DataLoader
    from yelprecommendation_amd.data.datasets.mf_data_pipeline import MFDataPipeline
    from yelprecommendation_amd.data.datasets.mf_dataset import MFDataset
    from yelprecommendation_amd.models.mf import MatrixFactorization
    from yelprecommendation_amd.utils import make_config, set_seed
    cfgv = dict(zip(g["cfg_names"].tolist(), g["cfg_values"].tolist()))
    df = pd.DataFrame({"user_id": g["tsv_user"].astype(np.int64), "business_id": g["tsv_item"].astype(np.int64),
                       "rating": g["tsv_rating"].astype(np.int64)})
    cfg = make_config("MF", seed=int(cfgv["seed"]), device="cpu", embed_size=int(cfgv["embed_size"]),
                      batch_size=int(cfgv["batch_size"]))
    pipe = MFDataPipeline(cfg)
    pipe._set_num_items_and_num_users(df)
    train, valid, _, _ = pipe.split(df)
    set_seed(cfg.seed)                                              # reference train.py:57
    dl = DataLoader(MFDataset(train, num_items=pipe.num_items), batch_size=cfg.batch_size, shuffle=True)
    MatrixFactorization(cfg, pipe.num_users, pipe.num_items)        # model init consumes torch RNG first (train.py:88)
    n0 = int(g["train_batch_sizes"][:int(g["train_steps"][0])].sum())
    got = {k: [] for k in ("user_id", "pos_item", "neg_item")}
    sizes = []
    for batch in dl:
        sizes.append(len(batch["user_id"]))
        for k in got:
            got[k].append(batch[k].numpy())
            assert batch[k].dtype == torch.int64
    np.testing.assert_array_equal(sizes, g["train_batch_sizes"][:int(g["train_steps"][0])])
    np.testing.assert_array_equal(np.concatenate(got["user_id"]), g["train_u"][:n0])
    np.testing.assert_array_equal(np.concatenate(got["pos_item"]), g["train_p"][:n0])
    np.testing.assert_array_equal(np.concatenate(got["neg_item"]), g["train_n"][:n0])


def test_ngcf_pipeline_builds_reference_laplacian(golden_dir, tmp_path):
    from yelprecommendation_amd.data.datasets.ngcf_data_pipeline import NGCFDataPipeline
    from yelprecommendation_amd.utils import make_config
    g = np.load(os.path.join(golden_dir, "ngcf_tiny.npz"))
    df = pd.DataFrame({"user_id": g["tsv_user"], "business_id": g["tsv_item"], "rating": g["tsv_rating"]})
    df.to_csv(os.path.join(str(tmp_path), "yelp_interactions.tsv"), sep="\t", index=False)
    pipe = NGCFDataPipeline(make_config("NGCF", device="cpu", data_dir=str(tmp_path)))
    out = pipe.preprocess()
    assert len(out) == len(df) and (pipe.num_users, pipe.num_items) == (int(g["num_users"]), int(g["num_items"]))
    L = pipe.laplacian_matrix.coalesce()
    np.testing.assert_array_equal(L.indices().numpy(), np.stack([g["lap_row"], g["lap_col"]]))
    np.testing.assert_allclose(L.values().numpy(), g["lap_val"], rtol=1e-6, atol=1e-8)


def test_cdae_pipeline_reproduces_reference_split(golden_dir):
    """CDAEDataPipeline (pivot + per-user shuffle + 60/20/20 cut) against the masks the REFERENCE
    pipeline produced for the same frame under the same NumPy seed (tests/golden/cdae_small.npz,
    written by make_golden.py: make_frame(96, 200, 12.0, seed=99), np.random.seed(1))."""
    from yelprecommendation_amd.data.datasets.cdae_data_pipeline import CDAEDataPipeline
    from yelprecommendation_amd.data.datasets.cdae_dataset import CDAEDataset
    from yelprecommendation_amd.data.synthetic import make_frame
    from yelprecommendation_amd.utils import make_config
    g = np.load(os.path.join(golden_dir, "cdae_small.npz"))
    pipe = CDAEDataPipeline(make_config("CDAE", device="cpu", model_dir="/tmp/yr_cdae_pipe"))
    frame = pipe._transform_into_training_set(make_frame(96, 200, 12.0, seed=99))
    assert frame.shape == (int(g["num_users"]), int(g["num_items"]) + 1)
    np.random.seed(1)
    train_data, valid_data, test_data = pipe.split(frame)
    users = sorted(train_data)
    assert users == list(range(int(g["num_users"])))
    np.testing.assert_array_equal(np.stack([train_data[u]["input_mask"] for u in users]), g["train_input"])
    np.testing.assert_array_equal(np.stack([valid_data[u]["valid_mask"] for u in users]), g["valid_mask"])
    np.testing.assert_array_equal(np.stack([valid_data[u]["input_mask"] for u in users]), g["train_input"])
    np.testing.assert_array_equal(np.stack([test_data[u]["input_mask"] for u in users]), g["test_input"])
    np.testing.assert_array_equal(np.stack([test_data[u]["test_mask"] for u in users]), g["test_mask"])
    # the dataset's negatives: right count, never a positive, one np.random.choice per fetched user
    ds = CDAEDataset(valid_data, "valid", neg_times=2)
    np.random.seed(3)
    sample = ds[5]
    pos = sample["input_mask"] + sample["valid_mask"]
    assert sample["negative_mask"].sum() == 2 * pos.sum() and float((sample["negative_mask"] * pos).sum()) == 0.0
    np.random.seed(3)
    want = np.random.choice(np.flatnonzero(1 - pos), int(pos.sum()) * 2, replace=False)
    np.testing.assert_array_equal(np.flatnonzero(sample["negative_mask"]), np.sort(want))
    assert set(CDAEDataset(test_data, "test")[5]) == {"user_id", "input_mask", "test_mask"}
