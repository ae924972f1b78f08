"""GPU: the multi-GPU shape of the BPR-MF step on the one GPU of the test box.
(1) split item update (item pass emits the dense gradient, separate dense Adam) == fused step;
(2) two ranks sharing cuda:0 over gloo: user-sharded step == single-process oracle.  RCCL itself
needs one GPU per rank, so the collective here is gloo; the rest of the path is the product's."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import bpr_mf as obpr

pytestmark = pytest.mark.gpu


def _problem():
    rs = np.random.RandomState(31)
    nu, ni, d, B, steps = 301, 257, 64, 6000, 3
    U = (rs.standard_normal((nu, d)) * 0.2).astype(np.float32)
    I = (rs.standard_normal((ni, d)) * 0.2).astype(np.float32)
    batches = [(rs.randint(0, nu, B).astype(np.int64), rs.randint(0, ni, B).astype(np.int64),
                rs.randint(0, ni, B).astype(np.int64)) for _ in range(steps)]
    return nu, ni, d, B, U, I, batches


def test_split_item_update_equals_fused(device):
    from yelprecommendation_amd.bpr_step import BPRMFStep
    nu, ni, d, B, U, I, batches = _problem()
    a = BPRMFStep(torch.from_numpy(U).to(device), torch.from_numpy(I).to(device), lr=5e-3)
    b = BPRMFStep(torch.from_numpy(U).to(device), torch.from_numpy(I).to(device), lr=5e-3, split_item_update=True)
    for (u, p, n) in batches:
        t = [torch.from_numpy(x).to(device) for x in (u, p, n)]
        a.step(*t)
        b.step(*t)
    assert abs(a.epoch_loss() - b.epoch_loss()) < 1e-6
    torch.testing.assert_close(a.I, b.I, rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(a.U, b.U, rtol=1e-4, atol=1e-6)


def test_lookahead_index_equals_plain_steps(device):
    """step(..., next_batch=...) builds the next batch's index early (the multi-GPU overlap path);
    results must not change, also when the announced batch is NOT the one that arrives."""
    from yelprecommendation_amd.bpr_step import BPRMFStep
    nu, ni, d, B, U, I, batches = _problem()
    a = BPRMFStep(torch.from_numpy(U).to(device), torch.from_numpy(I).to(device), lr=5e-3, split_item_update=True)
    b = BPRMFStep(torch.from_numpy(U).to(device), torch.from_numpy(I).to(device), lr=5e-3, split_item_update=True)
    dev_batches = [tuple(torch.from_numpy(x).to(device) for x in bt) for bt in batches]
    for k, t in enumerate(dev_batches):
        a.step(*t)
        nxt = dev_batches[k + 1] if k + 1 < len(dev_batches) else None
        if k == 1:
            nxt = dev_batches[0]                      # a wrong announcement: must be ignored, not used
        b.step(*t, next_batch=nxt)
    assert abs(a.epoch_loss() - b.epoch_loss()) < 1e-6
    torch.testing.assert_close(a.I, b.I, rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(a.U, b.U, rtol=1e-4, atol=1e-6)
    a.check(); b.check()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir, item_exchange="all_reduce"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from yelprecommendation_amd.bpr_step import BPRMFStep
    from yelprecommendation_amd.user_shard import UserShard
    dev = torch.device("cuda:0")
    nu, ni, d, B, U, I, batches = _problem()
    shard = UserShard(nu, world, rank)
    impl = "auto"
    if item_exchange == "auto_skew":
        # impl="auto" with local batch sizes on BOTH sides of the switch: rank 0 owns ~85 % of every batch.
        # The form is chosen from the global batch (the two forms issue different collectives), so the
        # 400-triplet batches take the pull form and the 250-triplet one the atomic form on every rank.
        from yelprecommendation_amd import bpr_step
        bpr_step.AUTO_PULL_MIN_BATCH = 150                     # x world: switch at a global batch of 300
        item_exchange = "all_reduce"
        batches = _skewed_batches(nu, ni)
    step = BPRMFStep(torch.from_numpy(U[shard.lo:shard.hi].copy()).to(dev), torch.from_numpy(I).to(dev), lr=5e-3,
                     world_size=world, process_group=dist.group.WORLD, item_exchange=item_exchange, rank=rank, impl=impl)
    local = [tuple(torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in shard.select(u, p, n))
             for (u, p, n) in batches]
    used = []
    for k, t in enumerate(local):
        step.step(*t, global_batch=len(batches[k][0]), next_batch=local[k + 1] if k + 1 < len(local) else None)
        used.append(step.impl.split(":")[0])
    loss = step.epoch_loss()
    step.check()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), U=step.U.cpu().numpy(), I=step.I.cpu().numpy(),
             lo=shard.lo, hi=shard.hi, loss=loss, used=np.array(used), counts=np.array([t[0].numel() for t in local]))
    dist.destroy_process_group()


def _skewed_batches(nu, ni):
    rs = np.random.RandomState(77)
    out = []
    for B in (400, 250, 400):
        u = np.where(rs.rand(B) < 0.85, rs.randint(0, nu // 2, B), rs.randint(nu // 2, nu, B)).astype(np.int64)
        out.append((u, rs.randint(0, ni, B).astype(np.int64), rs.randint(0, ni, B).astype(np.int64)))
    return out


@pytest.mark.timeout(600)
def test_auto_form_is_the_same_on_every_rank(tmp_path, device):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), "auto_skew"), nprocs=world, join=True)
    nu, ni, d, B, U, I, _ = _problem()
    batches = _skewed_batches(nu, ni)
    ref = obpr.MFState(U, I, "adam", lr=5e-3)
    total = sum(float(ref.train_step(u, p, n)) for (u, p, n) in batches)
    outs = [np.load(os.path.join(str(tmp_path), f"rank{r}.npz")) for r in range(world)]
    assert outs[0]["used"].tolist() == outs[1]["used"].tolist() == ["pull", "atomic", "pull"]
    assert outs[0]["counts"][0] > 300 > outs[1]["counts"][0]       # local sizes straddle the switch
    for o in outs:
        np.testing.assert_allclose(o["U"], ref.U[int(o["lo"]):int(o["hi"])], rtol=1e-3, atol=1e-5)
        np.testing.assert_allclose(o["I"], ref.I, rtol=1e-3, atol=1e-5)
        np.testing.assert_allclose(float(o["loss"]), total, rtol=1e-5)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("item_exchange", ["all_reduce", "reduce_scatter"])
def test_two_ranks_on_one_gpu_match_oracle(tmp_path, device, item_exchange):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), item_exchange), nprocs=world, join=True)
    nu, ni, d, B, U, I, batches = _problem()
    ref = obpr.MFState(U, I, "adam", lr=5e-3)
    total = sum(float(ref.train_step(u, p, n)) for (u, p, n) in batches)
    for r in range(world):
        o = np.load(os.path.join(str(tmp_path), f"rank{r}.npz"))
        np.testing.assert_allclose(o["U"], ref.U[int(o["lo"]):int(o["hi"])], rtol=1e-3, atol=1e-5)
        np.testing.assert_allclose(o["I"], ref.I, rtol=1e-3, atol=1e-5)
        np.testing.assert_allclose(float(o["loss"]), total, rtol=1e-5)


def _nccl_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)      # RCCL
    from yelprecommendation_amd.bpr_step import BPRMFStep
    from yelprecommendation_amd.user_shard import sharded_item_exchange
    nu, ni, d, B, U, I, batches = _problem()
    step = BPRMFStep(torch.from_numpy(U).to(dev), torch.from_numpy(I).to(dev), lr=5e-3, split_item_update=True,
                     process_group=dist.group.WORLD)
    ref = BPRMFStep(torch.from_numpy(U).to(dev), torch.from_numpy(I).to(dev), lr=5e-3)
    # drive the product's exchange protocol through a real RCCL collective (a 1-rank group: SUM over
    # one rank is the identity, so the result must equal the fused single-GPU step)
    step.world_size = 2                       # take the collective branch ...
    local = [tuple(torch.from_numpy(x).to(dev) for x in bt) for bt in batches]
    for k, t in enumerate(local):
        step.step(*t, global_batch=B, next_batch=local[k + 1] if k + 1 < len(local) else None)
        ref.step(*t)
    step.world_size = 1
    ok = torch.allclose(step.I, ref.I, rtol=1e-4, atol=1e-6) and torch.allclose(step.U, ref.U, rtol=1e-4, atol=1e-6)
    open(os.path.join(out_dir, "ok"), "w").write(str(bool(ok)))
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_exchange_over_rccl_single_rank(tmp_path, device):
    mp.spawn(_nccl_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    assert open(os.path.join(str(tmp_path), "ok")).read() == "True"


# ------------------------------------------------------------------------------------------------
# The user-sharded MFTrainer (2 ranks on cuda:0 over gloo) against the single-process trainer
def _trainer_problem(tmp, optimizer="adam"):
    import pandas as pd
    from yelprecommendation_amd.data.synthetic import make_frame
    from yelprecommendation_amd.data.datasets.mf_data_pipeline import MFDataPipeline
    from yelprecommendation_amd.data.datasets.mf_dataset import MFDataset
    from yelprecommendation_amd.utils import make_config
    cfg = make_config("MF", embed_size=32, lr=5e-3, batch_size=512, epochs=3, device="cuda", model_dir=tmp,
                      seed=42, top_n=10, patience=5, best_metric="recall")
    if optimizer == "sgd":                     # plain SGD moves slowly: a large step and some weight decay
        cfg.update(optimizer="sgd", lr=2.0, weight_decay=1e-3)
    pipe = MFDataPipeline(cfg)
    df = make_frame(150, 120, 14.0)
    pipe._load_df = lambda: df
    df = pipe.preprocess()
    train_data, valid_data, valid_eval, test_eval = pipe.split(df)
    return cfg, pipe, MFDataset(train_data, num_items=pipe.num_items), MFDataset(valid_data, num_items=pipe.num_items), \
        valid_eval, test_eval


def _run_trainer(tmp, optimizer="adam"):
    from torch.utils.data import DataLoader
    from yelprecommendation_amd.trainers.mf_trainer import MFTrainer
    from yelprecommendation_amd.utils import set_seed
    cfg, pipe, train_ds, valid_ds, valid_eval, test_eval = _trainer_problem(tmp, optimizer)
    set_seed(cfg.seed)
    trainer = MFTrainer(cfg, pipe.num_items, pipe.num_users)
    trainer.run(DataLoader(train_ds, batch_size=cfg.batch_size, shuffle=True),
                DataLoader(valid_ds, batch_size=cfg.batch_size, shuffle=True), valid_eval)
    trainer.load_best_model()
    return trainer, trainer.evaluate(test_eval, 'test')


def _trainer_worker(rank, world, port, out_dir, optimizer="adam"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    trainer, metrics = _run_trainer(os.path.join(out_dir, "sharded"), optimizer)
    assert trainer.world_size == world and trainer.shard.rank == rank
    np.savez(os.path.join(out_dir, f"trainer_rank{rank}.npz"), metrics=np.asarray(metrics),
             U=trainer.model.user_embedding.weight.detach().cpu().numpy(),
             I=trainer.model.item_embedding.weight.detach().cpu().numpy())
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("optimizer", ["adam", "sgd"])
def test_sharded_trainer_matches_single_process(tmp_path, device, optimizer):
    """MFTrainer.run + load_best_model + evaluate under a 2-rank group == the same calls in one
    process: tables to float rounding (summation order of the item gradient), metrics identical
    on every rank and within 1e-3 of the single-process ones — with Adam (the fused step + item-gradient exchange)
    and with SGD + weight decay (reference trainers/base_trainer.py:39-40: scatter kernel, all-reduce, dense update
    of the item table and of the rank's own user rows)."""
    world = 2
    os.makedirs(os.path.join(str(tmp_path), "sharded"), exist_ok=True)
    mp.spawn(_trainer_worker, args=(world, _free_port(), str(tmp_path), optimizer), nprocs=world, join=True)
    single, want = _run_trainer(os.path.join(str(tmp_path), "single"), optimizer)
    outs = [np.load(os.path.join(str(tmp_path), f"trainer_rank{r}.npz")) for r in range(world)]
    np.testing.assert_array_equal(outs[0]["metrics"], outs[1]["metrics"])
    np.testing.assert_allclose(outs[0]["metrics"], np.asarray(want), atol=1e-3)
    for o in outs:
        np.testing.assert_allclose(o["U"], single.model.user_embedding.weight.detach().cpu().numpy(),
                                   rtol=2e-3, atol=2e-5)
        np.testing.assert_allclose(o["I"], single.model.item_embedding.weight.detach().cpu().numpy(),
                                   rtol=2e-3, atol=2e-5)


def test_eight_gpu_rank_shape_pull_equals_atomic(device):
    """The per-rank problem of the 8-GPU run (3,958 users x 38,048 items, 2^20 triplets with
    global_batch = 8 x 2^20: ~265 triplets per user, about half the user rows on the heavy path), in
    the multi-GPU form of the step (dense item gradient in 2 chunks, separate item Adam): the pull form
    equals the atomic form."""
    from yelprecommendation_amd.bpr_step import BPRMFStep
    g = torch.Generator(device=device).manual_seed(0)
    nu, ni, d, B = 3958, 38048, 64, 1 << 20
    U0 = torch.randn(nu, d, device=device, generator=g) * 0.05
    I0 = torch.randn(ni, d, device=device, generator=g) * 0.05
    u = torch.randint(0, nu, (B,), device=device, generator=g)
    p = torch.randint(0, ni, (B,), device=device, generator=g)
    n = torch.randint(0, ni, (B,), device=device, generator=g)
    res = {}
    for impl in ("pull", "atomic"):
        st = BPRMFStep(U0.clone(), I0.clone(), lr=1e-3, impl=impl, split_item_update=(impl == "pull"), item_chunks=2)
        for _ in range(2):
            st.step(u, p, n, global_batch=8 * B)
        st.check()
        res[impl] = (st.U.clone(), st.I.clone(), st.epoch_loss())
    torch.testing.assert_close(res["pull"][0], res["atomic"][0], rtol=1e-4, atol=2e-6)
    torch.testing.assert_close(res["pull"][1], res["atomic"][1], rtol=1e-4, atol=2e-6)
    assert abs(res["pull"][2] - res["atomic"][2]) < 1e-5 * abs(res["atomic"][2])
