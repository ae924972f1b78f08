#!/usr/bin/env python3
"""Generate the committed golden vectors under tests/golden/ by RUNNING THE
REFERENCE (twndus/YelpRecommendation, mounted read-only at /root/reference) on
CPU in the build container.  This script is the provenance of every ``*.npz``
next to it; it is never imported by the product, the tests or the bench, and it
cannot run on the GPU box (no /root/reference there).

How the reference is imported: its hot-path modules import
``loguru`` / ``wandb`` / ``omegaconf`` for logging and type hints only, and none
of the three is installed here (no network).  Before the import this script
registers inert stand-ins for exactly those three names in ``sys.modules``
(a logger whose methods do nothing, ``wandb.run = None``, ``DictConfig = dict``).
They touch no arithmetic: every number in the fixtures comes out of the
reference's own Python driving torch / numpy / pandas / scikit-learn.

What is NOT executed from the reference: ``NGCFDataPipeline._set_laplacian_matrix``
(reference data/datasets/ngcf_data_pipeline.py:19-44) hard-codes ``.to('cuda')``
and cannot run without a GPU, so the Laplacian handed to the reference's NGCF
model is produced here by the same op sequence on CPU (function
``laplacian_like_reference`` below).  That one producer is therefore pinned by
construction, not by execution; everything downstream of it (NGCF forward,
backward, Adam) is the reference's own code.

Usage:  python tests/golden/make_golden.py            (writes tests/golden/*.npz, ~1 min)
        python tests/golden/make_golden.py mf_full    (the headline-size MF run, ~20 min -> mf_full.npz)
"""
from __future__ import annotations

import os
os.environ.setdefault("TQDM_DISABLE", "1")
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


# --------------------------------------------------------------------------- #
# logging / config stand-ins (no arithmetic)
# --------------------------------------------------------------------------- #
class _Quiet:
    def __getattr__(self, name):
        return lambda *a, **k: None


def _install_stubs():
    loguru = types.ModuleType("loguru")
    loguru.logger = _Quiet()
    sys.modules["loguru"] = loguru

    wandb = types.ModuleType("wandb")
    wandb.run = None
    for n in ("init", "log", "finish", "sweep", "agent"):
        setattr(wandb, n, lambda *a, **k: None)
    wandb.config = {}
    sys.modules["wandb"] = wandb

    omegaconf = types.ModuleType("omegaconf")
    dictconfig = types.ModuleType("omegaconf.dictconfig")

    class DictConfig(dict):
        __getattr__ = dict.__getitem__
        __setattr__ = dict.__setitem__

    omegaconf.DictConfig = DictConfig
    omegaconf.OmegaConf = type("OmegaConf", (), {})
    dictconfig.DictConfig = DictConfig
    omegaconf.dictconfig = dictconfig
    sys.modules["omegaconf"] = omegaconf
    sys.modules["omegaconf.dictconfig"] = dictconfig
    return DictConfig


DictConfig = _install_stubs()
sys.path.insert(0, REF)
sys.path.insert(0, REPO)

import pandas as pd  # noqa: E402
import torch  # noqa: E402
from torch.utils.data import DataLoader  # noqa: E402

# the reference (read-only mount)
import loss as ref_loss  # noqa: E402
import metric as ref_metric  # noqa: E402
import utils as ref_utils  # noqa: E402
from data.datasets.mf_data_pipeline import MFDataPipeline  # noqa: E402
from data.datasets.mf_dataset import MFDataset  # noqa: E402
from data.datasets.cdae_data_pipeline import CDAEDataPipeline  # noqa: E402
from data.datasets.cdae_dataset import CDAEDataset  # noqa: E402
from models.mf import MatrixFactorization  # noqa: E402
from models.ngcf import NGCF  # noqa: E402
from models.cdae import CDAE  # noqa: E402
from trainers.mf_trainer import MFTrainer  # noqa: E402
from trainers.ngcf_trainer import NGCFTrainer  # noqa: E402
from trainers.cdae_trainer import CDAETrainer  # noqa: E402

# the build's own synthetic-data generator (not reference code)
from yelprecommendation_amd.data.synthetic import make_frame  # noqa: E402

import sklearn  # noqa: E402

VERSIONS = np.array([f"torch={torch.__version__}", f"numpy={np.__version__}",
                     f"pandas={pd.__version__}", f"sklearn={sklearn.__version__}"])


def _csr(lists):
    """list of int lists -> (ptr int64[n+1], idx int64[nnz])"""
    ptr = np.zeros(len(lists) + 1, dtype=np.int64)
    for k, l in enumerate(lists):
        ptr[k + 1] = ptr[k] + len(l)
    idx = np.concatenate([np.asarray(l, dtype=np.int64) for l in lists]) if lists else np.zeros(0, np.int64)
    return ptr, idx


class RecordingLoader:
    """Iterates a DataLoader and keeps what it yielded (batch boundaries kept)."""

    def __init__(self, dl, keys):
        self.dl, self.keys, self.epochs = dl, keys, []

    def __iter__(self):
        rec = {k: [] for k in self.keys}
        rec["_sizes"] = []
        self.epochs.append(rec)
        for batch in self.dl:
            for k in self.keys:
                rec[k].append(batch[k].numpy().copy())
            rec["_sizes"].append(len(batch[self.keys[0]]))
            yield batch

    def __len__(self):
        return len(self.dl)


class RecordingLoss(torch.nn.Module):
    def __init__(self, inner):
        super().__init__()
        self.inner, self.values = inner, []

    def forward(self, *a):
        out = self.inner(*a)
        self.values.append(float(out.item()))
        return out


# --------------------------------------------------------------------------- #
# (i)+(ii)  BPR-MF, BASELINE.json configs[0]: 1k users x ~1k items, D = 32
# --------------------------------------------------------------------------- #
def golden_mf(out_path, num_users=1000, num_items=1000, mean_items=30.0, embed=32,
              lr=5e-3, batch=256, epochs=4, seed=42):
    cfg = DictConfig(seed=seed, shuffle=True, model_dir=tempfile.mkdtemp(), device="cpu",
                     epochs=epochs, batch_size=batch, lr=lr, optimizer="adam", loss_name="bpr",
                     patience=5, top_n=10, weight_decay=0, best_metric="loss", wandb=False,
                     model_name="MF", embed_size=embed, data_dir="unused")
    df = make_frame(num_users, num_items, mean_items, seed=1234)

    pipe = MFDataPipeline(cfg)
    pipe._set_num_items_and_num_users(df)                  # mf_data_pipeline.py:73-75
    train_data, valid_data, valid_eval, test_eval = pipe.split(df)   # train.py:161
    train_ds = MFDataset(train_data, num_items=pipe.num_items)       # train.py:162-163
    valid_ds = MFDataset(valid_data, num_items=pipe.num_items)

    ref_utils.set_seed(cfg.seed)                                     # train.py:57
    train_dl = RecordingLoader(DataLoader(train_ds, batch_size=cfg.batch_size, shuffle=cfg.shuffle),
                               ("user_id", "pos_item", "neg_item"))  # train.py:76-77
    valid_dl = RecordingLoader(DataLoader(valid_ds, batch_size=cfg.batch_size, shuffle=cfg.shuffle),
                               ("user_id", "pos_item", "neg_item"))
    trainer = MFTrainer(cfg, pipe.num_items, pipe.num_users)         # train.py:88
    U0 = trainer.model.user_embedding.weight.detach().numpy().copy()
    I0 = trainer.model.item_embedding.weight.detach().numpy().copy()
    trainer.loss = RecordingLoss(trainer.loss)

    epoch_log = {"train_loss": [], "valid_loss": [], "valid_metrics": [], "U": [], "I": []}
    orig_train, orig_valid, orig_eval = trainer.train, trainer.validate, trainer.evaluate
    eval_top = {}

    def rec_train(dl):
        v = orig_train(dl)
        epoch_log["train_loss"].append(v)
        epoch_log["U"].append(trainer.model.user_embedding.weight.detach().numpy().copy())
        epoch_log["I"].append(trainer.model.item_embedding.weight.detach().numpy().copy())
        return v

    def rec_valid(dl):
        v = orig_valid(dl)
        epoch_log["valid_loss"].append(v)
        return v

    def rec_eval(data, mode="valid"):
        m = orig_eval(data, mode)
        if mode == "valid":
            epoch_log["valid_metrics"].append(m)
        return m

    trainer.train, trainer.validate, trainer.evaluate = rec_train, rec_valid, rec_eval
    trainer.run(train_dl, valid_dl, valid_eval)                      # train.py:89
    trainer.load_best_model()                                        # train.py:90
    best_U = trainer.model.user_embedding.weight.detach().numpy().copy()
    best_I = trainer.model.item_embedding.weight.detach().numpy().copy()
    test_metrics = trainer.evaluate(test_eval, "test")               # train.py:91

    # per-user top-10 lists of the best model, produced by the reference's own
    # _generate_top_k_recommendation (mf_trainer.py:163-178)
    item_input = torch.arange(pipe.num_items)
    for name, frame in (("valid", valid_eval), ("test", test_eval)):
        tops = []
        for user_id, row in frame.iterrows():
            pred = trainer.model(torch.tensor([user_id] * pipe.num_items), item_input)
            tops.append(trainer._generate_top_k_recommendation(pred, row["mask_items"]))
        eval_top[name] = np.stack(tops).astype(np.int64)

    st = trainer.optimizer.state
    pU, pI = trainer.model.user_embedding.weight, trainer.model.item_embedding.weight
    n_train_steps = [len(e["_sizes"]) for e in train_dl.epochs]
    n_valid_steps = [len(e["_sizes"]) for e in valid_dl.epochs]
    # loss call order inside run(): per epoch all train steps then all valid steps
    losses = np.asarray(trainer.loss.values, dtype=np.float64)
    tl, vl, pos = [], [], 0
    for a, b in zip(n_train_steps, n_valid_steps):
        tl.append(losses[pos:pos + a]); pos += a
        vl.append(losses[pos:pos + b]); pos += b

    def cat(eps, k):
        return np.concatenate([np.concatenate(e[k]) for e in eps]).astype(np.int32)

    vp_ptr, vp_idx = _csr(list(valid_eval["pos_items"]))
    vm_ptr, vm_idx = _csr(list(valid_eval["mask_items"]))
    tp_ptr, tp_idx = _csr(list(test_eval["pos_items"]))
    tm_ptr, tm_idx = _csr(list(test_eval["mask_items"]))
    trp_ptr, trp_idx = _csr([list(r) for r in train_data.groupby("user_id")["pos_items"].first()])
    np.savez_compressed(
        out_path,
        versions=VERSIONS,
        cfg_names=np.array(["embed_size", "lr", "batch_size", "epochs", "seed", "top_n"]),
        cfg_values=np.array([embed, lr, batch, epochs, seed, 10], dtype=np.float64),
        num_users=np.int64(pipe.num_users), num_items=np.int64(pipe.num_items),
        tsv_user=df.user_id.values.astype(np.int32), tsv_item=df.business_id.values.astype(np.int32),
        tsv_rating=df.rating.values.astype(np.int32),
        # split() outputs, row order preserved (mf_data_pipeline.py:38-52)
        train_user=train_data.user_id.values.astype(np.int32), train_item=train_data.business_id.values.astype(np.int32),
        train_index=train_data["index"].values.astype(np.int32),
        valid_user=valid_data.user_id.values.astype(np.int32), valid_item=valid_data.business_id.values.astype(np.int32),
        valid_index=valid_data["index"].values.astype(np.int32),
        train_pos_ptr=trp_ptr, train_pos_idx=trp_idx.astype(np.int32),
        valid_eval_users=valid_eval.index.values.astype(np.int64), valid_pos_ptr=vp_ptr, valid_pos_idx=vp_idx.astype(np.int32),
        valid_mask_ptr=vm_ptr, valid_mask_idx=vm_idx.astype(np.int32),
        test_eval_users=test_eval.index.values.astype(np.int64), test_pos_ptr=tp_ptr, test_pos_idx=tp_idx.astype(np.int32),
        test_mask_ptr=tm_ptr, test_mask_idx=tm_idx.astype(np.int32),
        U0=U0, I0=I0,
        train_steps=np.asarray(n_train_steps), valid_steps=np.asarray(n_valid_steps),
        train_batch_sizes=np.concatenate([e["_sizes"] for e in train_dl.epochs]).astype(np.int32),
        valid_batch_sizes=np.concatenate([e["_sizes"] for e in valid_dl.epochs]).astype(np.int32),
        train_u=cat(train_dl.epochs, "user_id"), train_p=cat(train_dl.epochs, "pos_item"), train_n=cat(train_dl.epochs, "neg_item"),
        valid_u=cat(valid_dl.epochs, "user_id"), valid_p=cat(valid_dl.epochs, "pos_item"), valid_n=cat(valid_dl.epochs, "neg_item"),
        train_step_loss=np.concatenate(tl), valid_step_loss=np.concatenate(vl),
        train_epoch_loss=np.asarray(epoch_log["train_loss"]), valid_epoch_loss=np.asarray(epoch_log["valid_loss"]),
        valid_metrics=np.asarray(epoch_log["valid_metrics"], dtype=np.float64),
        test_metrics=np.asarray(test_metrics, dtype=np.float64),
        U_epoch0=epoch_log["U"][0], I_epoch0=epoch_log["I"][0],
        U_final=epoch_log["U"][-1], I_final=epoch_log["I"][-1],
        U_best=best_U, I_best=best_I,
        adam_step=np.int64(int(st[pU]["step"])),
        mU=st[pU]["exp_avg"].numpy(), vU=st[pU]["exp_avg_sq"].numpy(),
        mI=st[pI]["exp_avg"].numpy(), vI=st[pI]["exp_avg_sq"].numpy(),
        top10_valid=eval_top["valid"].astype(np.int32), top10_test=eval_top["test"].astype(np.int32),
    )
    print(f"[mf] U={pipe.num_users} I={pipe.num_items} train={len(train_data)} valid={len(valid_data)} "
          f"epochs={len(n_train_steps)} valid_metrics[-1]={epoch_log['valid_metrics'][-1]} test={test_metrics}")


# --------------------------------------------------------------------------- #
# (i')  BPR-MF at the HEADLINE configuration (BASELINE.json configs[1]): the Yelp2018-shaped
#       synthetic set (31,668 users x 38,048 items, D = 64), the reference's own split, DataLoader,
#       MFTrainer.run / load_best_model / evaluate on CPU.  ~20 min in the build container.
#       The fixture is SMALL: hashes of the split and of every epoch's triplet stream (the test
#       regenerates both with the repo's host mirrors and must hit the same hashes), per-step and
#       per-epoch losses, the four metrics per epoch and on test, and — for a fixed sample of
#       users / items — final table rows, Adam moments and the reference's own top-10 lists
#       (test lists for ALL users: the per-user loop of mf_trainer.py:134-161 is recorded as it runs).
# --------------------------------------------------------------------------- #
def _sha(*arrays):
    import hashlib
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(np.asarray(a, dtype=np.int64)).tobytes())
    return h.hexdigest()


FULL_FRAME = dict(num_users=31_668, num_items=38_048, mean_items=48.0, seed=1234, min_item_degree=5)


def golden_mf_full(out_path, embed=64, lr=5e-3, batch=4096, epochs=2, seed=42, n_sample=512):
    import time
    t0 = time.time()
    cfg = DictConfig(seed=seed, shuffle=True, model_dir=tempfile.mkdtemp(), device="cpu",
                     epochs=epochs, batch_size=batch, lr=lr, optimizer="adam", loss_name="bpr",
                     patience=5, top_n=10, weight_decay=0, best_metric="loss", wandb=False,
                     model_name="MF", embed_size=embed, data_dir="unused")
    df = make_frame(**FULL_FRAME)
    print(f"[mf_full] frame {len(df)} rows ({time.time() - t0:.0f} s)", flush=True)
    pipe = MFDataPipeline(cfg)
    pipe._set_num_items_and_num_users(df)
    train_data, valid_data, valid_eval, test_eval = pipe.split(df)   # train.py:161 (the reference's sklearn split)
    print(f"[mf_full] split: train {len(train_data)} valid {len(valid_data)} ({time.time() - t0:.0f} s)", flush=True)
    train_ds = MFDataset(train_data, num_items=pipe.num_items)
    valid_ds = MFDataset(valid_data, num_items=pipe.num_items)

    ref_utils.set_seed(cfg.seed)
    keys = ("user_id", "pos_item", "neg_item")
    train_dl = RecordingLoader(DataLoader(train_ds, batch_size=cfg.batch_size, shuffle=cfg.shuffle), keys)
    valid_dl = RecordingLoader(DataLoader(valid_ds, batch_size=cfg.batch_size, shuffle=cfg.shuffle), keys)
    trainer = MFTrainer(cfg, pipe.num_items, pipe.num_users)
    rs = np.random.RandomState(7)
    su = np.sort(rs.choice(pipe.num_users, n_sample, replace=False)).astype(np.int64)
    si = np.sort(rs.choice(pipe.num_items, n_sample, replace=False)).astype(np.int64)
    pU, pI = trainer.model.user_embedding.weight, trainer.model.item_embedding.weight
    U0s, I0s = pU.detach().numpy()[su].copy(), pI.detach().numpy()[si].copy()
    init_sum = np.array([pU.detach().double().sum().item(), pI.detach().double().sum().item()])
    trainer.loss = RecordingLoss(trainer.loss)

    log = {"train": [], "valid": [], "metrics": [], "Us": [], "Is": []}
    tops = {}                                                        # mode -> list of [k] arrays, call order = frame order
    o_train, o_valid, o_eval, o_top = trainer.train, trainer.validate, trainer.evaluate, trainer._generate_top_k_recommendation
    cur = {"mode": None}

    def rec_top(pred, mask_items):
        out = o_top(pred, mask_items)
        tops[cur["mode"]].append(np.asarray(out, dtype=np.int64).copy())
        return out

    def rec_train(dl):
        v = o_train(dl)
        log["train"].append(v)
        log["Us"].append(pU.detach().numpy()[su].copy())
        log["Is"].append(pI.detach().numpy()[si].copy())
        print(f"[mf_full] epoch {len(log['train']) - 1} train loss {v:.6f} ({time.time() - t0:.0f} s)", flush=True)
        return v

    def rec_valid(dl):
        v = o_valid(dl)
        log["valid"].append(v)
        return v

    def rec_eval(data, mode="valid"):
        key = f"valid{len(log['metrics'])}" if mode == "valid" else "test"
        cur["mode"] = key
        tops[key] = []
        m = o_eval(data, mode)
        if mode == "valid":
            log["metrics"].append(m)
        print(f"[mf_full] evaluate({mode}) = {m} ({time.time() - t0:.0f} s)", flush=True)
        return m

    trainer.train, trainer.validate, trainer.evaluate = rec_train, rec_valid, rec_eval
    trainer._generate_top_k_recommendation = rec_top
    trainer.run(train_dl, valid_dl, valid_eval)                      # train.py:89
    trainer.load_best_model()                                        # train.py:90
    best_Us, best_Is = pU.detach().numpy()[su].copy(), pI.detach().numpy()[si].copy()
    best_sum = np.array([pU.detach().double().sum().item(), pI.detach().double().sum().item()])
    test_metrics = trainer.evaluate(test_eval, "test")               # train.py:91

    st = trainer.optimizer.state
    n_train_steps = [len(e["_sizes"]) for e in train_dl.epochs]
    n_valid_steps = [len(e["_sizes"]) for e in valid_dl.epochs]
    losses = np.asarray(trainer.loss.values, dtype=np.float64)
    tl, vl, pos = [], [], 0
    for a, b in zip(n_train_steps, n_valid_steps):
        tl.append(losses[pos:pos + a]); pos += a
        vl.append(losses[pos:pos + b]); pos += b

    def stream_hashes(eps):
        return np.array([_sha(*(np.concatenate(e[k]) for k in keys)) for e in eps])

    best_epoch = int(np.argmin(log["valid"]))                        # best_metric == 'loss' (base_trainer.py:117-141)
    valid_users = valid_eval.index.values.astype(np.int64)
    test_users = test_eval.index.values.astype(np.int64)
    sel_v = np.flatnonzero(np.isin(valid_users, su))
    vp_ptr, vp_idx = _csr(list(valid_eval["pos_items"]))
    vm_ptr, vm_idx = _csr(list(valid_eval["mask_items"]))
    tp_ptr, tp_idx = _csr(list(test_eval["pos_items"]))
    tm_ptr, tm_idx = _csr(list(test_eval["mask_items"]))
    np.savez_compressed(
        out_path,
        versions=VERSIONS,
        frame_names=np.array(list(FULL_FRAME)), frame_values=np.array(list(FULL_FRAME.values()), dtype=np.float64),
        cfg_names=np.array(["embed_size", "lr", "batch_size", "epochs", "seed", "top_n"]),
        cfg_values=np.array([embed, lr, batch, epochs, seed, 10], dtype=np.float64),
        num_users=np.int64(pipe.num_users), num_items=np.int64(pipe.num_items), num_rows=np.int64(len(df)),
        tsv_sha=np.array(_sha(df.user_id.values, df.business_id.values, df.rating.values)),
        # split(): row order and list contents (mf_data_pipeline.py:18-52)
        split_sha=np.array([
            _sha(train_data["index"].values, train_data.user_id.values, train_data.business_id.values),
            _sha(valid_data["index"].values, valid_data.user_id.values, valid_data.business_id.values),
            _sha(valid_users, vp_ptr, vp_idx, vm_ptr, vm_idx),
            _sha(test_users, tp_ptr, tp_idx, tm_ptr, tm_idx)]),
        split_rows=np.array([len(train_data), len(valid_data), len(valid_eval), len(test_eval)], dtype=np.int64),
        train_steps=np.asarray(n_train_steps), valid_steps=np.asarray(n_valid_steps),
        train_stream_sha=stream_hashes(train_dl.epochs), valid_stream_sha=stream_hashes(valid_dl.epochs),
        # the first and last batch of the first epoch in clear, to localise a stream mismatch
        train_first_batch=np.stack([train_dl.epochs[0][k][0] for k in keys]).astype(np.int32),
        train_last_batch=np.stack([train_dl.epochs[0][k][-1] for k in keys]).astype(np.int32),
        train_step_loss=np.concatenate(tl), valid_step_loss=np.concatenate(vl),
        train_epoch_loss=np.asarray(log["train"]), valid_epoch_loss=np.asarray(log["valid"]),
        valid_metrics=np.asarray(log["metrics"], dtype=np.float64), test_metrics=np.asarray(test_metrics, dtype=np.float64),
        best_epoch=np.int64(best_epoch),
        sample_users=su, sample_items=si, init_sum=init_sum, best_sum=best_sum,
        U0_rows=U0s, I0_rows=I0s,
        U_rows_epoch=np.stack(log["Us"]), I_rows_epoch=np.stack(log["Is"]),
        U_rows_best=best_Us, I_rows_best=best_Is,
        adam_step=np.int64(int(st[pU]["step"])),
        mU_rows=st[pU]["exp_avg"].numpy()[su], vU_rows=st[pU]["exp_avg_sq"].numpy()[su],
        mI_rows=st[pI]["exp_avg"].numpy()[si], vI_rows=st[pI]["exp_avg_sq"].numpy()[si],
        # the reference's own lists: every test user (best model), the sampled users of the last validation
        top10_test=np.stack(tops["test"]).astype(np.int32),
        top10_valid_last_rows=sel_v.astype(np.int64),
        top10_valid_last=np.stack(tops[f"valid{len(log['metrics']) - 1}"])[sel_v].astype(np.int32),
    )
    print(f"[mf_full] U={pipe.num_users} I={pipe.num_items} steps={n_train_steps} valid={log['metrics']} "
          f"test={test_metrics} best_epoch={best_epoch} ({time.time() - t0:.0f} s)")


# --------------------------------------------------------------------------- #
# (iii)  NGCF tiny graph
# --------------------------------------------------------------------------- #
def laplacian_like_reference(df, num_users, num_items):
    """CPU re-run of the op sequence of ngcf_data_pipeline.py:23-42 (which itself
    hard-codes .to('cuda')): pivot(mean rating) -> A = [[0,R],[R^T,0]] ->
    D^-1/2 A D^-1/2 via two torch.sparse.mm calls."""
    uii = df.pivot_table(index="user_id", columns=["business_id"], values=["rating"])
    uii = uii.droplevel(0, 1).fillna(0)
    n = num_users + num_items
    adj = np.zeros((n, n), dtype=np.float32)
    adj[:num_users, num_users:] = uii
    adj[num_users:, :num_users] = uii.T
    deg = np.diag(1 / np.sqrt(adj.sum(axis=0))).astype(np.float32)
    deg = torch.from_numpy(deg).to_sparse()
    adj = torch.from_numpy(adj).to_sparse()
    lap = torch.sparse.mm(deg, adj)
    lap = torch.sparse.mm(lap, deg)
    return lap.coalesce()


def golden_ngcf(out_path, num_users=120, num_items=90, mean_items=9.0, embed=16, orders=2,
                lr=1e-2, batch=64, epochs=2, seed=42):
    cfg = DictConfig(seed=seed, shuffle=True, model_dir=tempfile.mkdtemp(), device="cpu",
                     epochs=epochs, batch_size=batch, lr=lr, optimizer="adam", loss_name="bpr",
                     patience=5, top_n=10, weight_decay=0, best_metric="loss", wandb=False,
                     model_name="NGCF", embed_size=embed, num_orders=orders, data_dir="unused")
    df = make_frame(num_users, num_items, mean_items, seed=77)
    pipe = MFDataPipeline(cfg)          # NGCFDataPipeline == MFDataPipeline + laplacian (ngcf_data_pipeline.py:13-49)
    pipe._set_num_items_and_num_users(df)
    U, I = pipe.num_users, pipe.num_items
    lap = laplacian_like_reference(df, U, I)
    train_data, valid_data, valid_eval, test_eval = pipe.split(df)
    train_ds = MFDataset(train_data, num_items=I)      # NGCFDataset is MFDataset (ngcf_dataset.py:6)
    valid_ds = MFDataset(valid_data, num_items=I)

    ref_utils.set_seed(cfg.seed)
    train_dl = RecordingLoader(DataLoader(train_ds, batch_size=batch, shuffle=True), ("user_id", "pos_item", "neg_item"))
    valid_dl = RecordingLoader(DataLoader(valid_ds, batch_size=batch, shuffle=True), ("user_id", "pos_item", "neg_item"))
    trainer = NGCFTrainer(cfg, I, U, lap)
    model: NGCF = trainer.model
    names = [n for n, _ in model.named_parameters()]
    P0 = {n: p.detach().numpy().copy() for n, p in model.named_parameters()}

    # ---- single-step probe on a fixed batch (does not disturb the RNG stream used below:
    #      it consumes none) -------------------------------------------------------------
    rs = np.random.RandomState(5)
    bu = torch.from_numpy(rs.randint(0, U, size=48).astype(np.int64))
    bp = torch.from_numpy(rs.randint(0, I, size=48).astype(np.int64))
    bn = torch.from_numpy(rs.randint(0, I, size=48).astype(np.int64))
    pos, neg = model.bpr_forward(bu, bp, bn, lap)                 # ngcf.py:30-45
    l = ref_loss.BPRLoss()(pos, neg)
    model.zero_grad()
    l.backward()
    G = {n: p.grad.detach().numpy().copy() for n, p in model.named_parameters()}
    with torch.no_grad():
        fwd_scores = model(bu, bp, lap).numpy().copy()            # ngcf.py:47-58
        e1 = model.embedding_propagation(model.embedding.weight, model.W1[0], model.W2[0], lap).numpy().copy()
    model.zero_grad()

    # ---- the trainer's own loop ---------------------------------------------------------
    trainer.loss = RecordingLoss(trainer.loss)
    log = {"train": [], "valid": [], "metrics": [], "eval_idx": []}
    o_train, o_valid, o_eval = trainer.train, trainer.validate, trainer.evaluate
    rand_orig = np.random.randint

    def rec_eval(data, mode="valid"):
        # evaluate() draws its 100 user positions from the global NumPy RNG
        # (ngcf_trainer.py:140); record them by wrapping the call it makes.
        def spy(*a, **k):
            out = rand_orig(*a, **k)
            if k.get("size", None) == 100:
                log["eval_idx"].append(np.asarray(out).copy())
            return out
        np.random.randint = spy
        try:
            m = o_eval(data, mode)
        finally:
            np.random.randint = rand_orig
        log["metrics"].append(m)
        return m

    trainer.train = lambda dl: (log["train"].append(o_train(dl)), log["train"][-1])[1]
    trainer.validate = lambda dl: (log["valid"].append(o_valid(dl)), log["valid"][-1])[1]
    trainer.evaluate = rec_eval
    trainer.run(train_dl, valid_dl, valid_eval)
    P1 = {n: p.detach().numpy().copy() for n, p in model.named_parameters()}

    def cat(eps, k):
        return np.concatenate([np.concatenate(e[k]) for e in eps]).astype(np.int32)

    n_train_steps = [len(e["_sizes"]) for e in train_dl.epochs]
    n_valid_steps = [len(e["_sizes"]) for e in valid_dl.epochs]
    losses = np.asarray(trainer.loss.values, dtype=np.float64)
    tl, vl, posn = [], [], 0
    for a, b in zip(n_train_steps, n_valid_steps):
        tl.append(losses[posn:posn + a]); posn += a
        vl.append(losses[posn:posn + b]); posn += b
    vp_ptr, vp_idx = _csr(list(valid_eval["pos_items"]))
    vm_ptr, vm_idx = _csr(list(valid_eval["mask_items"]))
    out = dict(
        versions=VERSIONS, num_users=np.int64(U), num_items=np.int64(I), num_orders=np.int64(orders),
        embed_size=np.int64(embed), lr=np.float64(lr), batch_size=np.int64(batch),
        tsv_user=df.user_id.values.astype(np.int32), tsv_item=df.business_id.values.astype(np.int32),
        tsv_rating=df.rating.values.astype(np.int32),
        lap_row=lap.indices()[0].numpy().astype(np.int32), lap_col=lap.indices()[1].numpy().astype(np.int32),
        lap_val=lap.values().numpy(),
        param_names=np.array(names),
        probe_u=bu.numpy(), probe_p=bp.numpy(), probe_n=bn.numpy(),
        probe_pos=pos.detach().numpy(), probe_neg=neg.detach().numpy(), probe_loss=np.float64(l.item()),
        probe_forward=fwd_scores, probe_layer1=e1,
        train_steps=np.asarray(n_train_steps), valid_steps=np.asarray(n_valid_steps),
        train_batch_sizes=np.concatenate([e["_sizes"] for e in train_dl.epochs]).astype(np.int32),
        valid_batch_sizes=np.concatenate([e["_sizes"] for e in valid_dl.epochs]).astype(np.int32),
        train_u=cat(train_dl.epochs, "user_id"), train_p=cat(train_dl.epochs, "pos_item"), train_n=cat(train_dl.epochs, "neg_item"),
        valid_u=cat(valid_dl.epochs, "user_id"), valid_p=cat(valid_dl.epochs, "pos_item"), valid_n=cat(valid_dl.epochs, "neg_item"),
        train_step_loss=np.concatenate(tl), valid_step_loss=np.concatenate(vl),
        train_epoch_loss=np.asarray(log["train"]), valid_epoch_loss=np.asarray(log["valid"]),
        eval_metrics=np.asarray(log["metrics"], dtype=np.float64),
        eval_positions=np.stack(log["eval_idx"]).astype(np.int64),
        valid_eval_users=valid_eval.index.values.astype(np.int64),
        valid_pos_ptr=vp_ptr, valid_pos_idx=vp_idx.astype(np.int32),
        valid_mask_ptr=vm_ptr, valid_mask_idx=vm_idx.astype(np.int32),
    )
    for n in names:
        key = n.replace(".", "__")
        out["init__" + key] = P0[n]
        out["grad__" + key] = G[n]
        out["final__" + key] = P1[n]
    np.savez_compressed(out_path, **out)
    print(f"[ngcf] U={U} I={I} nnz(L)={lap._nnz()} steps={n_train_steps} metrics={log['metrics']}")


# --------------------------------------------------------------------------- #
# (iv)  CDAE small
# --------------------------------------------------------------------------- #
def golden_cdae(out_path, num_users=96, num_items=200, mean_items=12.0, hidden=16,
                lr=1e-2, batch=32, epochs=2, seed=42):
    cfg = DictConfig(seed=seed, shuffle=True, model_dir=tempfile.mkdtemp(), device="cpu",
                     epochs=epochs, batch_size=batch, lr=lr, optimizer="adam", loss_name="bce",
                     patience=5, top_n=10, weight_decay=0, best_metric="loss", wandb=False,
                     model_name="CDAE", negative_sampling=True, neg_times=5, hidden_size=hidden,
                     corruption_level=0.6, hidden_activation="sigmoid", output_activation="sigmoid",
                     data_dir="unused")
    df = make_frame(num_users, num_items, mean_items, seed=99)
    pipe = CDAEDataPipeline(cfg)
    training_set = pipe._transform_into_training_set(df)       # cdae_data_pipeline.py:78-90
    np.random.seed(1)                                           # split() is unseeded in the reference
    train_data, valid_data, test_data = pipe.split(training_set)   # (cdae_data_pipeline.py:30); fixed here so the fixture is reproducible
    num_items_ = len(training_set.columns) - 1                  # train.py:159
    num_users_ = len(train_data)
    train_ds = CDAEDataset(train_data, "train", neg_times=cfg.neg_times)
    valid_ds = CDAEDataset(valid_data, "valid", neg_times=cfg.neg_times)
    test_ds = CDAEDataset(test_data, "test")

    ref_utils.set_seed(cfg.seed)
    keys_t = ("user_id", "input_mask", "negative_mask")
    keys_v = ("user_id", "input_mask", "valid_mask", "negative_mask")
    train_dl = RecordingLoader(DataLoader(train_ds, batch_size=batch, shuffle=True), keys_t)
    valid_dl = RecordingLoader(DataLoader(valid_ds, batch_size=batch, shuffle=True), keys_v)
    test_dl = DataLoader(test_ds, batch_size=batch)
    trainer = CDAETrainer(cfg, num_items_, num_users_)
    model: CDAE = trainer.model
    names = [n for n, _ in model.named_parameters()]
    P0 = {n: p.detach().numpy().copy() for n, p in model.named_parameters()}

    # record every corrupted input the dropout layer produces (cdae.py:43-44)
    corrupted = []
    model.dropout_layer.register_forward_hook(lambda m, i, o: corrupted.append(o.detach().numpy().copy()))

    # ---- single-batch probe in eval mode (dropout off => deterministic) ----
    model.eval()
    rs = np.random.RandomState(3)
    pu = torch.arange(0, 24, dtype=torch.int64)
    px = torch.from_numpy(np.stack([train_data[int(u)]["input_mask"].astype("float32") for u in pu]))
    pneg = torch.from_numpy((rs.rand(*px.shape) < 0.2).astype("float32")) * (1 - px)
    pred = model(pu, px)                                        # cdae.py:46-52
    l = ref_loss.NSBCELoss()(pred, px, pneg)                    # loss.py:12-16
    model.zero_grad()
    l.backward()
    G = {n: p.grad.detach().numpy().copy() for n, p in model.named_parameters()}
    model.zero_grad()
    corrupted.clear()

    trainer.loss = RecordingLoss(trainer.loss)
    log = {"train": [], "valid": []}
    o_train, o_valid = trainer.train, trainer.validate
    trainer.train = lambda dl: (log["train"].append(o_train(dl)), log["train"][-1])[1]
    trainer.validate = lambda dl: (log["valid"].append(o_valid(dl)), log["valid"][-1])[1]
    trainer.run(train_dl, valid_dl)                             # train.py:84
    P1 = {n: p.detach().numpy().copy() for n, p in model.named_parameters()}
    test_metrics = CDAETrainer.evaluate.__wrapped__(trainer, test_dl)   # log_metric only adds wandb logging (utils.py:27-41)

    n_train_steps = [len(e["_sizes"]) for e in train_dl.epochs]
    n_valid_steps = [len(e["_sizes"]) for e in valid_dl.epochs]
    losses = np.asarray(trainer.loss.values, dtype=np.float64)
    tl, vl, posn = [], [], 0
    for a, b in zip(n_train_steps, n_valid_steps):
        tl.append(losses[posn:posn + a]); posn += a
        vl.append(losses[posn:posn + b]); posn += b

    def cat(eps, k, dt):
        return np.concatenate([np.concatenate(e[k]) for e in eps]).astype(dt)

    # the dropout hook fires for train AND valid forwards (eval mode => identity); keep train ones
    # order of forwards inside run(): per epoch train steps then valid steps, then test
    cor_train, posn = [], 0
    for a, b in zip(n_train_steps, n_valid_steps):
        cor_train.extend(corrupted[posn:posn + a]); posn += a + b
    keep = np.concatenate(cor_train) != 0                        # dropout keep-mask where x != 0 ...
    x_train = cat(train_dl.epochs, "input_mask", np.uint8)
    # ... but a dropped 0 and a kept 0 are both 0: store the corrupted tensor's support instead
    out = dict(
        versions=VERSIONS, num_users=np.int64(num_users_), num_items=np.int64(num_items_),
        hidden_size=np.int64(hidden), lr=np.float64(lr), batch_size=np.int64(batch),
        corruption_level=np.float64(0.6), neg_times=np.int64(5),
        train_input=np.stack([train_data[u]["input_mask"] for u in range(num_users_)]).astype(np.uint8),
        valid_mask=np.stack([valid_data[u]["valid_mask"] for u in range(num_users_)]).astype(np.uint8),
        test_input=np.stack([test_data[u]["input_mask"] for u in range(num_users_)]).astype(np.uint8),
        test_mask=np.stack([test_data[u]["test_mask"] for u in range(num_users_)]).astype(np.uint8),
        param_names=np.array(names),
        probe_user=pu.numpy(), probe_x=px.numpy().astype(np.uint8), probe_neg=pneg.numpy().astype(np.uint8),
        probe_pred=pred.detach().numpy(), probe_loss=np.float64(l.item()),
        train_steps=np.asarray(n_train_steps), valid_steps=np.asarray(n_valid_steps),
        train_batch_sizes=np.concatenate([e["_sizes"] for e in train_dl.epochs]).astype(np.int32),
        valid_batch_sizes=np.concatenate([e["_sizes"] for e in valid_dl.epochs]).astype(np.int32),
        train_user=cat(train_dl.epochs, "user_id", np.int32),
        train_neg=cat(train_dl.epochs, "negative_mask", np.uint8),
        train_keep=keep.astype(np.uint8),
        valid_user=cat(valid_dl.epochs, "user_id", np.int32),
        valid_neg=cat(valid_dl.epochs, "negative_mask", np.uint8),
        train_step_loss=np.concatenate(tl), valid_step_loss=np.concatenate(vl),
        train_epoch=np.asarray([t for t in log["train"]], dtype=np.float64),
        valid_epoch=np.asarray([list(v) for v in log["valid"]], dtype=np.float64),
        test_metrics=np.asarray(test_metrics, dtype=np.float64),
    )
    assert (x_train[keep == 0] >= 0).all()
    for n in names:
        key = n.replace(".", "__")
        out["init__" + key] = P0[n]
        out["grad__" + key] = G[n]
        out["final__" + key] = P1[n]
    np.savez_compressed(out_path, **out)
    print(f"[cdae] U={num_users_} I={num_items_} steps={n_train_steps} valid={log['valid'][-1]} test={test_metrics}")


# --------------------------------------------------------------------------- #
# (v)  metric.py on random ragged lists (beyond the 4 KATs of test/test_metric.py)
# --------------------------------------------------------------------------- #
def golden_metric(out_path):
    rs = np.random.RandomState(0)
    cases = []
    for _ in range(40):
        nu = rs.randint(1, 8)
        actual = [list(rs.choice(30, size=rs.randint(0, 9), replace=False)) for _ in range(nu)]
        if all(len(a) == 0 for a in actual):
            actual[0] = [1, 2]
        predicted = [list(rs.choice(30, size=10, replace=False)) for _ in range(nu)]
        k = int(rs.choice([1, 3, 5, 10]))
        vals = []
        for fn in (ref_metric.precision_at_k, ref_metric.recall_at_k, ref_metric.map_at_k, ref_metric.ndcg_at_k):
            try:
                vals.append(float(fn(actual, predicted, k)))
            except ZeroDivisionError:
                vals.append(np.nan)
        cases.append((actual, predicted, k, vals))
    a_ptr, a_idx = _csr([a for c in cases for a in c[0]])
    np.savez_compressed(
        out_path, versions=VERSIONS,
        case_users=np.asarray([len(c[0]) for c in cases]), k=np.asarray([c[2] for c in cases]),
        actual_ptr=a_ptr, actual_idx=a_idx,
        predicted=np.asarray([p for c in cases for p in c[1]], dtype=np.int64),
        values=np.asarray([c[3] for c in cases], dtype=np.float64))
    print(f"[metric] {len(cases)} cases")


if __name__ == "__main__":
    which = sys.argv[1:] or ["mf", "ngcf", "cdae", "metric"]
    torch.set_num_threads(8)
    if "mf" in which:
        golden_mf(os.path.join(HERE, "mf_small.npz"))
    if "mf_full" in which:                       # ~20 min: only when named (python make_golden.py mf_full)
        golden_mf_full(os.path.join(HERE, "mf_full.npz"))
    if "ngcf" in which:
        golden_ngcf(os.path.join(HERE, "ngcf_tiny.npz"))
        # BASELINE configs[3]'s hyper-parameters (D = 64, K = 3 layers) on a graph the reference can still propagate
        # on CPU (it builds eye(N, N) per layer per batch)
        golden_ngcf(os.path.join(HERE, "ngcf_mid.npz"), num_users=600, num_items=500, mean_items=14.0, embed=64,
                    orders=3, lr=2e-3, batch=256, epochs=2, seed=42)
    if "cdae" in which:
        golden_cdae(os.path.join(HERE, "cdae_small.npz"))
        # BASELINE configs[4]'s hidden size (H = 128) and the reference's default batch (32) on a ~500-item catalogue
        golden_cdae(os.path.join(HERE, "cdae_mid.npz"), num_users=256, num_items=700, mean_items=16.0, hidden=128,
                    lr=2e-3, batch=32, epochs=2, seed=42)
    if "metric" in which:
        golden_metric(os.path.join(HERE, "metric_cases.npz"))
