"""BPR-MF at the HEADLINE configuration (BASELINE.json configs[1]: Yelp2018 shape, 31,668 users x 38,048 items,
dim 64) against a run of the REFERENCE ITSELF: tests/golden/mf_full.npz was written by
`tests/golden/make_golden.py mf_full`, which drives the reference's own MFDataPipeline.split, MFDataset +
DataLoader, MFTrainer.run / load_best_model / evaluate on CPU (2 epochs of 230 steps at batch 4,096, lr 5e-3,
seed 42; ~20 min) on the seeded synthetic frame.  The fixture is small: hashes of the frame, the split and every
epoch's triplet stream, losses, metrics, sampled table rows and the reference's own top-10 lists.

CPU test: the repo's host mirrors (synthetic frame -> MFDataPipeline.split -> DataLoader(MFDataset)) regenerate the
frame, the split and both epochs' train / valid streams BIT FOR BIT (SURVEY 8 rows a10 / a11 at full size).
GPU test: the HIP engine trains on that regenerated stream through MFTrainer.run and must give the reference's
per-step and per-epoch losses (rtol 1e-4), sampled table rows (rtol 1e-3), Recall@10 / NDCG@10 / P@10 / MAP@10 of
every validation and of the test evaluation within 1e-3, and the reference's top-10 lists up to near-ties."""
import functools
import hashlib
import os

import numpy as np
import pytest
import torch

from replay import assert_topk_equal_up_to_near_ties

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mf_full.npz")


def _sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(np.asarray(a, dtype=np.int64)).tobytes())
    return h.hexdigest()


def _csr(col):
    ptr = np.zeros(len(col) + 1, np.int64)
    ptr[1:] = np.cumsum([len(x) for x in col])
    return ptr, np.concatenate([np.asarray(x, dtype=np.int64) for x in col])


@functools.lru_cache(maxsize=1)
def _data():
    """(golden, cfg values, frame, pipeline, the four frames of split()) — ~40 s, once per session."""
    from yelprecommendation_amd.data.datasets.mf_data_pipeline import MFDataPipeline
    from yelprecommendation_amd.data.synthetic import make_frame
    from yelprecommendation_amd.utils import make_config
    g = np.load(GOLDEN)
    kw = dict(zip(g["frame_names"].tolist(), g["frame_values"].tolist()))
    df = make_frame(int(kw["num_users"]), int(kw["num_items"]), kw["mean_items"], seed=int(kw["seed"]),
                    min_item_degree=int(kw["min_item_degree"]))
    c = dict(zip(g["cfg_names"].tolist(), g["cfg_values"].tolist()))
    pipe = MFDataPipeline(make_config("MF", seed=int(c["seed"]), device="cpu", loss_name="bpr"))
    pipe._set_num_items_and_num_users(df)
    return g, c, df, pipe, pipe.split(df)


class _Recorder:
    """Iterates a DataLoader, keeps the epoch's triplets for the stream hash, and calls ``after_batch`` with the
    number of batches handed out so far before it fetches the next one (to read the previous step's loss)."""

    def __init__(self, dl, after_batch=None):
        self.dl, self.after_batch, self.epochs = dl, after_batch, []

    def __len__(self):
        return len(self.dl)

    def __iter__(self):
        rec = {k: [] for k in ("user_id", "pos_item", "neg_item")}
        self.epochs.append(rec)
        done = 0
        for batch in self.dl:
            if self.after_batch and done:
                self.after_batch()
            for k in rec:
                assert batch[k].dtype == torch.int64
                rec[k].append(batch[k].numpy().copy())
            done += 1
            yield batch
        if self.after_batch and done:
            self.after_batch()

    def sha(self, epoch):
        rec = self.epochs[epoch]
        return _sha(*(np.concatenate(rec[k]) for k in ("user_id", "pos_item", "neg_item")))


def _loaders(c, pipe, train, valid, after_train_batch=None):
    from torch.utils.data import DataLoader
    from yelprecommendation_amd.data.datasets.mf_dataset import MFDataset
    bs = int(c["batch_size"])
    return (_Recorder(DataLoader(MFDataset(train, num_items=pipe.num_items), batch_size=bs, shuffle=True), after_train_batch),
            _Recorder(DataLoader(MFDataset(valid, num_items=pipe.num_items), batch_size=bs, shuffle=True)))


def test_host_mirrors_reproduce_the_reference_split_and_stream_at_full_size():
    from yelprecommendation_amd.models.mf import MatrixFactorization
    from yelprecommendation_amd.utils import make_config, set_seed
    g, c, df, pipe, (train, valid, valid_eval, test_eval) = _data()
    assert (pipe.num_users, pipe.num_items, len(df)) == (int(g["num_users"]), int(g["num_items"]), int(g["num_rows"]))
    assert _sha(df.user_id.values, df.business_id.values, df.rating.values) == str(g["tsv_sha"])
    # split(): reference data/datasets/mf_data_pipeline.py:18-52 (per-user sklearn train_test_split x 2)
    np.testing.assert_array_equal([len(train), len(valid), len(valid_eval), len(test_eval)], g["split_rows"])
    assert _sha(train["index"].values, train.user_id.values, train.business_id.values) == g["split_sha"][0]
    assert _sha(valid["index"].values, valid.user_id.values, valid.business_id.values) == g["split_sha"][1]
    for frame, want in ((valid_eval, g["split_sha"][2]), (test_eval, g["split_sha"][3])):
        assert _sha(frame.index.values, *_csr(frame["pos_items"]), *_csr(frame["mask_items"])) == want
    # the stream: set_seed -> model init (consumes the torch generator first, train.py:88) -> per epoch the train
    # loader's permutation + one np.random.randint rejection sequence per row, then the valid loader's
    # (reference train.py:57,76-77, data/datasets/mf_dataset.py:18-32)
    set_seed(int(c["seed"]))
    tdl, vdl = _loaders(c, pipe, train, valid)
    cfg = make_config("MF", seed=int(c["seed"]), device="cpu", embed_size=int(c["embed_size"]))
    model = MatrixFactorization(cfg, pipe.num_users, pipe.num_items)
    np.testing.assert_array_equal(model.user_embedding.weight.detach().numpy()[g["sample_users"]], g["U0_rows"])
    np.testing.assert_array_equal(model.item_embedding.weight.detach().numpy()[g["sample_items"]], g["I0_rows"])
    for e in range(int(c["epochs"])):
        sizes = [len(b["user_id"]) for b in tdl]
        assert len(sizes) == int(g["train_steps"][e]) and sizes[-1] == g["train_last_batch"].shape[1]
        if e == 0:
            first = np.stack([tdl.epochs[0][k][0] for k in ("user_id", "pos_item", "neg_item")])
            np.testing.assert_array_equal(first, g["train_first_batch"])
        assert tdl.sha(e) == g["train_stream_sha"][e]
        assert sum(1 for _ in vdl) == int(g["valid_steps"][e])
        assert vdl.sha(e) == g["valid_stream_sha"][e]


@pytest.mark.gpu
def test_full_size_training_run_matches_the_reference_run(device, tmp_path):
    from yelprecommendation_amd.trainers import MFTrainer
    from yelprecommendation_amd.utils import make_config, set_seed
    g, c, df, pipe, (train, valid, valid_eval, test_eval) = _data()
    cfg = make_config("MF", embed_size=int(c["embed_size"]), lr=c["lr"], batch_size=int(c["batch_size"]),
                      epochs=int(c["epochs"]), seed=int(c["seed"]), top_n=int(c["top_n"]), device="cuda",
                      model_dir=str(tmp_path), best_metric="loss", patience=5)
    set_seed(cfg.seed)                                              # reference train.py:57
    step_losses = []
    holder = {}

    def after_batch():                                              # the loss of the step that has just been enqueued
        if holder["t"]._step is not None:
            step_losses.append(float(holder["t"]._step.loss.item()))

    tdl, vdl = _loaders(c, pipe, train, valid, after_batch)
    t = holder["t"] = MFTrainer(cfg, pipe.num_items, pipe.num_users)        # train.py:88
    su, si = g["sample_users"], g["sample_items"]
    U, I = t.model.user_embedding.weight, t.model.item_embedding.weight
    np.testing.assert_array_equal(U.detach().cpu().numpy()[su], g["U0_rows"])
    np.testing.assert_array_equal(I.detach().cpu().numpy()[si], g["I0_rows"])
    np.testing.assert_allclose([U.detach().double().sum().item(), I.detach().double().sum().item()], g["init_sum"], rtol=1e-9)

    log = {"train": [], "valid": [], "metrics": [], "Us": [], "Is": []}
    o_train, o_valid, o_eval = t.train, t.validate, t.evaluate

    def rec_train(dl):
        v = o_train(dl)
        log["train"].append(v)
        log["Us"].append(t.model.user_embedding.weight.detach().cpu().numpy()[su])
        log["Is"].append(t.model.item_embedding.weight.detach().cpu().numpy()[si])
        return v

    t.train = rec_train
    t.validate = lambda dl: (log["valid"].append(o_valid(dl)), log["valid"][-1])[1]
    t.evaluate = lambda data, mode="valid": (log["metrics"].append(o_eval(data, mode)), log["metrics"][-1])[1]
    t.run(tdl, vdl, valid_eval)                                     # train.py:89
    epochs = int(c["epochs"])
    # the engine trained on the reference's stream (same permutations, same rejection-sampled negatives)
    for e in range(epochs):
        assert tdl.sha(e) == g["train_stream_sha"][e] and vdl.sha(e) == g["valid_stream_sha"][e]
    np.testing.assert_allclose(step_losses, g["train_step_loss"], rtol=1e-4)
    np.testing.assert_allclose(log["train"], g["train_epoch_loss"], rtol=1e-4)
    np.testing.assert_allclose(log["valid"], g["valid_epoch_loss"], rtol=1e-4)
    # Recall@10 / NDCG@10 (and P@10, MAP@10) of every epoch's validation: +-1e-3 of the reference's CPU run
    np.testing.assert_allclose(np.asarray(log["metrics"]), g["valid_metrics"], atol=1e-3, rtol=0)
    assert g["valid_metrics"][-1][1] > 0.1                          # (a model that has learnt: Recall@10 = 0.125)
    for e in range(epochs):
        np.testing.assert_allclose(log["Us"][e], g["U_rows_epoch"][e], rtol=1e-3, atol=1e-4)
        np.testing.assert_allclose(log["Is"][e], g["I_rows_epoch"][e], rtol=1e-3, atol=1e-4)
    st = t.optimizer.state[t.model.user_embedding.weight]
    assert int(st["step"]) == int(g["adam_step"])
    np.testing.assert_allclose(st["exp_avg"].cpu().numpy()[su], g["mU_rows"], rtol=2e-3, atol=1e-7)
    np.testing.assert_allclose(st["exp_avg_sq"].cpu().numpy()[su], g["vU_rows"], rtol=2e-3, atol=1e-11)
    sti = t.optimizer.state[t.model.item_embedding.weight]
    np.testing.assert_allclose(sti["exp_avg"].cpu().numpy()[si], g["mI_rows"], rtol=2e-3, atol=1e-7)
    np.testing.assert_allclose(sti["exp_avg_sq"].cpu().numpy()[si], g["vI_rows"], rtol=2e-3, atol=1e-11)

    # best model (lowest validation loss, base_trainer.py:117-141) -> test evaluation (train.py:90-91)
    assert int(np.argmin(log["valid"])) == int(g["best_epoch"])
    t.load_best_model()
    Ub = t.model.user_embedding.weight.detach().cpu().numpy()
    Ib = t.model.item_embedding.weight.detach().cpu().numpy()
    np.testing.assert_allclose(Ub[su], g["U_rows_best"], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(Ib[si], g["I_rows_best"], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose([Ub.astype(np.float64).sum(), Ib.astype(np.float64).sum()], g["best_sum"], rtol=1e-3)
    test_metrics = o_eval(test_eval, "test")
    np.testing.assert_allclose(test_metrics, g["test_metrics"], atol=1e-3, rtol=0)

    # the reference's own top-10 lists (its per-user loop, mf_trainer.py:134-178): every test user and the sampled
    # users of the last validation.  The two runs' tables agree to ~1e-3, so two items whose scores are closer
    # than that may swap places: every differing row is checked to be such a near-tie (no agreement quota).
    differing = []
    for frame, want, rows in ((test_eval, g["top10_test"], None), (valid_eval, g["top10_valid_last"], g["top10_valid_last_rows"])):
        _, users, mask_ptr, mask_idx = t._eval_arrays(frame)
        top = t.recommend(users, mask_ptr, mask_idx).cpu().numpy()
        users = users.cpu().numpy()
        masks = [np.asarray(m) for m in frame["mask_items"]]
        if rows is not None:
            top, users, masks = top[rows], users[rows], [masks[r] for r in rows]
        ndiff = assert_topk_equal_up_to_near_ties(top, want, Ub, Ib, users, masks, rel=3e-3)
        assert ndiff <= len(want) // 5, f"{ndiff} of {len(want)} lists differ"
        differing.append((ndiff, len(want)))
    if os.environ.get("YR_PARITY_REPORT"):                          # the measured deltas, for profiles/ (not a check)
        import json
        rel = lambda a, b: float(np.max(np.abs(np.asarray(a) - np.asarray(b)) / np.abs(np.asarray(b))))
        json.dump({"train_step_loss_max_rel": rel(step_losses, g["train_step_loss"]),
                   "train_epoch_loss": log["train"], "reference_train_epoch_loss": g["train_epoch_loss"].tolist(),
                   "valid_epoch_loss": log["valid"], "reference_valid_epoch_loss": g["valid_epoch_loss"].tolist(),
                   "valid_metrics_P_R_MAP_NDCG": [list(m) for m in log["metrics"]],
                   "reference_valid_metrics": g["valid_metrics"].tolist(),
                   "test_metrics_P_R_MAP_NDCG": list(test_metrics), "reference_test_metrics": g["test_metrics"].tolist(),
                   "max_abs_metric_delta": float(max(np.max(np.abs(np.asarray(log["metrics"]) - g["valid_metrics"])),
                                                     np.max(np.abs(np.asarray(test_metrics) - g["test_metrics"])))),
                   "sampled_rows_max_abs_delta": float(max(np.max(np.abs(Ub[su] - g["U_rows_best"])), np.max(np.abs(Ib[si] - g["I_rows_best"])))),
                   "top10_lists_differing_of_total": {"test_all_users": differing[0], "valid_sampled_users": differing[1]}},
                  open(os.environ["YR_PARITY_REPORT"], "w"), indent=1)
