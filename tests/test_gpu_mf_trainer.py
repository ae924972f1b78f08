"""GPU parity, trainer level: MFTrainer on the recorded triplet stream of the golden
reference run (tests/golden/mf_small.npz, BASELINE.json configs[0]) must reproduce the
reference's per-epoch losses, weights, Recall@10/NDCG@10 and top-10 lists."""
import os

import numpy as np
import pandas as pd
import pytest
import torch

from replay import ReplayLoader, assert_topk_equal_up_to_near_ties, epoch_slices

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "mf_small.npz"))


def _cfg(g, tmp_path):
    from yelprecommendation_amd.utils import make_config
    c = dict(zip(g["cfg_names"].tolist(), g["cfg_values"].tolist()))
    return make_config("MF", embed_size=int(c["embed_size"]), lr=c["lr"], batch_size=int(c["batch_size"]),
                       epochs=int(c["epochs"]), seed=int(c["seed"]), top_n=int(c["top_n"]),
                       device="cuda", model_dir=str(tmp_path))


def _eval_frame(users, pos_ptr, pos_idx, mask_ptr, mask_idx):
    return pd.DataFrame({
        "pos_items": [pos_idx[pos_ptr[r]:pos_ptr[r + 1]].tolist() for r in range(len(users))],
        "mask_items": [mask_idx[mask_ptr[r]:mask_ptr[r + 1]].tolist() for r in range(len(users))],
    }, index=pd.Index(users, name="user_id"))


def test_seeded_init_matches_reference(g, tmp_path, device):
    from yelprecommendation_amd.trainers import MFTrainer
    from yelprecommendation_amd.utils import set_seed
    cfg = _cfg(g, tmp_path)
    set_seed(cfg.seed)                      # reference train.py:57 then MFTrainer(...) at :88
    t = MFTrainer(cfg, int(g["num_items"]), int(g["num_users"]))
    np.testing.assert_array_equal(t.model.user_embedding.weight.detach().cpu().numpy(), g["U0"])
    np.testing.assert_array_equal(t.model.item_embedding.weight.detach().cpu().numpy(), g["I0"])
    assert sorted(t.model.state_dict().keys()) == ["item_embedding.weight", "user_embedding.weight"]


def test_training_run_matches_reference(g, tmp_path, device):
    from yelprecommendation_amd.trainers import MFTrainer
    cfg = _cfg(g, tmp_path)
    t = MFTrainer(cfg, int(g["num_items"]), int(g["num_users"]))
    with torch.no_grad():
        t.model.user_embedding.weight.copy_(torch.from_numpy(g["U0"]))
        t.model.item_embedding.weight.copy_(torch.from_numpy(g["I0"]))
    valid_eval = _eval_frame(g["valid_eval_users"], g["valid_pos_ptr"], g["valid_pos_idx"],
                             g["valid_mask_ptr"], g["valid_mask_idx"])
    tb, vb = g["train_batch_sizes"], g["valid_batch_sizes"]
    tsl, vsl = epoch_slices(g["train_steps"], tb), epoch_slices(g["valid_steps"], vb)
    for e, ((tb0, tb1, tr0, tr1), (vb0, vb1, vr0, vr1)) in enumerate(zip(tsl, vsl)):
        train_loss = t.train(ReplayLoader(g["train_u"][tr0:tr1], g["train_p"][tr0:tr1], g["train_n"][tr0:tr1], tb[tb0:tb1]))
        valid_loss = t.validate(ReplayLoader(g["valid_u"][vr0:vr1], g["valid_p"][vr0:vr1], g["valid_n"][vr0:vr1], vb[vb0:vb1]))
        metrics = t.evaluate(valid_eval, "valid")
        # per-epoch SUM of batch-mean losses (mf_trainer.py:116), rtol 1e-4
        np.testing.assert_allclose(train_loss, g["train_epoch_loss"][e], rtol=1e-4)
        np.testing.assert_allclose(valid_loss, g["valid_epoch_loss"][e], rtol=1e-4)
        # Recall@10 / NDCG@10 (and P, MAP) within +-1e-3 of the reference CPU run
        np.testing.assert_allclose(metrics, g["valid_metrics"][e], atol=1e-3, rtol=0)
        if e == 0:
            np.testing.assert_allclose(t.model.user_embedding.weight.detach().cpu().numpy(), g["U_epoch0"], rtol=1e-3, atol=2e-5)
    U = t.model.user_embedding.weight.detach().cpu().numpy()
    I = t.model.item_embedding.weight.detach().cpu().numpy()
    np.testing.assert_allclose(U, g["U_final"], rtol=1e-3, atol=5e-5)
    np.testing.assert_allclose(I, g["I_final"], rtol=1e-3, atol=5e-5)
    st = t.optimizer.state[t.model.user_embedding.weight]
    assert st["step"] == int(g["adam_step"])
    np.testing.assert_allclose(st["exp_avg"].cpu().numpy(), g["mU"], rtol=1e-3, atol=1e-7)
    np.testing.assert_allclose(st["exp_avg_sq"].cpu().numpy(), g["vU"], rtol=1e-3, atol=1e-10)


def test_evaluate_and_top10_match_reference(g, tmp_path, device):
    from yelprecommendation_amd.trainers import MFTrainer
    cfg = _cfg(g, tmp_path)
    t = MFTrainer(cfg, int(g["num_items"]), int(g["num_users"]))
    # state_dict round trip with the reference's keys (base_trainer.py:110,155-157)
    torch.save({"user_embedding.weight": torch.from_numpy(g["U_best"]),
                "item_embedding.weight": torch.from_numpy(g["I_best"])}, os.path.join(str(tmp_path), "best_model.pt"))
    t.load_best_model()
    test_eval = _eval_frame(g["test_eval_users"], g["test_pos_ptr"], g["test_pos_idx"],
                            g["test_mask_ptr"], g["test_mask_idx"])
    metrics = t.evaluate(test_eval, "test")
    np.testing.assert_allclose(metrics, g["test_metrics"], atol=1e-3, rtol=0)
    _, users, mask_ptr, mask_idx = t._eval_arrays(test_eval)
    top = t.recommend(users, mask_ptr, mask_idx).cpu().numpy()
    ref = g["top10_test"]
    # identical lists, except rows where two exact scores are closer than float32 rounding: every differing
    # row is examined (no agreement quota), and such rows are rare
    masks = [g["test_mask_idx"][g["test_mask_ptr"][r]:g["test_mask_ptr"][r + 1]] for r in range(len(ref))]
    ndiff = assert_topk_equal_up_to_near_ties(top, ref, g["U_best"], g["I_best"], users.cpu().numpy(), masks)
    assert ndiff <= len(ref) // 100
    # single-user entry point of the reference surface
    pred = t.model(torch.full((int(g["num_items"]),), int(users[3]), dtype=torch.int64, device=device),
                   torch.arange(int(g["num_items"]), device=device))
    one = t._generate_top_k_recommendation(pred, test_eval.iloc[3]["mask_items"])
    assert one.tolist() == top[3].tolist()


def test_reference_style_loop_matches_fused(g, tmp_path, device):
    """The reference's own loop shape — model(u,p), model(u,n), zero_grad, loss, backward,
    step (mf_trainer.py:106-112) — runs on the HIP autograd ops and agrees with the fused op."""
    from yelprecommendation_amd.trainers import MFTrainer
    cfg = _cfg(g, tmp_path)
    a = MFTrainer(cfg, int(g["num_items"]), int(g["num_users"]))
    b = MFTrainer(cfg, int(g["num_items"]), int(g["num_users"]))
    for t in (a, b):
        with torch.no_grad():
            t.model.user_embedding.weight.copy_(torch.from_numpy(g["U0"]))
            t.model.item_embedding.weight.copy_(torch.from_numpy(g["I0"]))
    sizes = g["train_batch_sizes"][:20]
    n = int(sizes.sum())
    loader = ReplayLoader(g["train_u"][:n], g["train_p"][:n], g["train_n"][:n], sizes)
    losses = []
    for data in loader:
        u, p, q = (data[k].to(device) for k in ("user_id", "pos_item", "neg_item"))
        pos_pred = a.model(u, p)
        neg_pred = a.model(u, q)
        a.optimizer.zero_grad()
        loss = a.loss(pos_pred, neg_pred)
        loss.backward()
        a.optimizer.step()
        losses.append(loss.item())
    total = b.train(loader)
    np.testing.assert_allclose(sum(losses), total, rtol=1e-5)
    np.testing.assert_allclose(losses, g["train_step_loss"][:20], rtol=1e-4)
    np.testing.assert_allclose(a.model.item_embedding.weight.detach().cpu().numpy(),
                               b.model.item_embedding.weight.detach().cpu().numpy(), rtol=1e-4, atol=1e-6)


def test_device_metrics_match_reference_definition(device, golden_dir):
    """yr_rank_metrics vs the host functions (the reference's definitions, pinned by its KATs and by
    metric_cases.npz): ragged lists, empty `actual` rows, duplicates inside `actual`, all k."""
    from yelprecommendation_amd import engine, metric
    c = np.load(os.path.join(golden_dir, "metric_cases.npz"))
    u0 = 0
    for case, (nu, k) in enumerate(zip(c["case_users"], c["k"])):
        actual = [c["actual_idx"][c["actual_ptr"][u]:c["actual_ptr"][u + 1]].tolist() for u in range(u0, u0 + nu)]
        predicted = c["predicted"][u0:u0 + nu]
        u0 += nu
        ptr = np.zeros(nu + 1, np.int64); ptr[1:] = np.cumsum([len(a) for a in actual])
        idx = np.asarray([x for a in actual for x in a], dtype=np.int64)
        out = engine.rank_metrics(torch.from_numpy(np.ascontiguousarray(predicted[:, :int(k)])).to(device),
                                  torch.from_numpy(ptr).to(device), torch.from_numpy(idx).to(device)).cpu().numpy()
        np.testing.assert_allclose(out[:4], c["values"][case], rtol=1e-12, atol=1e-15)
    # the reference's own known-answer test (test/test_metric.py:9-47) and a case with duplicates
    actual = [[1, 2, 3, 4, 5], [6, 7, 8, 9, 10], [], [3, 3, 9, 3]]
    predicted = np.array([[1, 6, 7, 11, 12], [6, 7, 14, 16, 20], [1, 2, 3, 4, 5], [9, 3, 1, 2, 4]], dtype=np.int64)
    ptr = np.zeros(5, np.int64); ptr[1:] = np.cumsum([len(a) for a in actual])
    idx = np.asarray([x for a in actual for x in a], dtype=np.int64)
    for k in (1, 2, 3, 5):
        out = engine.rank_metrics(torch.from_numpy(np.ascontiguousarray(predicted[:, :k])).to(device),
                                  torch.from_numpy(ptr).to(device), torch.from_numpy(idx).to(device)).cpu().numpy()
        want = metric.ranking_metrics(actual, predicted.tolist(), k)
        np.testing.assert_allclose(out[:4], want, rtol=1e-12)
        assert out[4] == 3


def test_checkpoint_resume_continues_the_run(g, tmp_path, device):
    """save_checkpoint / load_checkpoint (weights + Adam moments + step counts): a fresh trainer that
    loads the file and replays the next epoch's batches ends with the same tables as the trainer
    that kept running — through the fused step (which shares the optimizer's state).  Equal to float
    rounding, not bitwise: the scatter-add order of a step is not fixed."""
    from yelprecommendation_amd.trainers import MFTrainer
    cfg = _cfg(g, tmp_path)
    rs = np.random.RandomState(4)
    nu, ni = int(g["num_users"]), int(g["num_items"])
    def batches(k):
        return [{"user_id": torch.from_numpy(rs.randint(0, nu, 300)), "pos_item": torch.from_numpy(rs.randint(0, ni, 300)),
                 "neg_item": torch.from_numpy(rs.randint(0, ni, 300))} for _ in range(k)]
    first, second = batches(5), batches(5)
    a = MFTrainer(cfg, ni, nu)
    a.train(first)
    path = os.path.join(str(tmp_path), "ckpt.pt")
    a.save_checkpoint(path, epoch=1, best=0.25)
    la = a.train(second)
    b = MFTrainer(cfg, ni, nu)                                   # different random init, overwritten by the load
    extra = b.load_checkpoint(path)
    assert extra == {"epoch": 1, "best": 0.25}
    lb = b.train(second)
    assert abs(la - lb) <= 1e-6 * abs(la)
    for (na, pa), (nb, pb) in zip(a.model.named_parameters(), b.model.named_parameters()):
        assert na == nb
        torch.testing.assert_close(pa, pb, rtol=1e-5, atol=1e-7)
    sa, sb = a.optimizer.state_dict()["state"], b.optimizer.state_dict()["state"]
    for k in sa:
        assert sa[k]["step"] == sb[k]["step"] == 10
        torch.testing.assert_close(sa[k]["exp_avg"], sb[k]["exp_avg"], rtol=1e-4, atol=1e-9)
        torch.testing.assert_close(sa[k]["exp_avg_sq"], sb[k]["exp_avg_sq"], rtol=1e-4, atol=1e-12)
    # without the optimizer state the continuation is a different run
    c = MFTrainer(cfg, ni, nu)
    c.model.load_state_dict(a.model.state_dict())
    assert c.optimizer.state_dict()["state"] == {}


def test_whole_epoch_validate_equals_per_batch_loop(device, tmp_path):
    """MFTrainer.validate over the device-side EpochLoader: the sum over batches of the batch-mean loss in two
    launches (full batches weighted 1 / batch_size, the short last batch 1 / its length) == the per-batch loop of
    the reference's shape (trainers/mf_trainer.py:118-132) on the same sampled epoch; batch sizes that divide the
    epoch, leave a remainder, and exceed it."""
    from yelprecommendation_amd.data.synthetic import make_interactions_torch
    from yelprecommendation_amd.data.triplets import EpochLoader, TripletSampler
    from yelprecommendation_amd.trainers import MFTrainer
    from yelprecommendation_amd.utils import make_config
    nu, ni = 300, 500
    u, i = make_interactions_torch(nu, ni, 12.0, seed=3, device=device)
    for bs in (32, 100, u.numel(), 3 * u.numel()):
        got = {}
        for whole in (True, False):
            torch.manual_seed(9)
            t = MFTrainer(make_config("MF", device="cuda", model_dir=str(tmp_path), embed_size=32, batch_size=bs,
                                      whole_epoch_validate=whole), ni, nu)
            got[whole] = t.validate(EpochLoader(TripletSampler(u, i, nu, ni, seed=5), bs))
        np.testing.assert_allclose(got[True], got[False], rtol=2e-6)


def test_evaluation_hints_do_not_change_metrics_or_lists(g, tmp_path, device):
    """MFTrainer hands the top-n lists of one evaluation to the next evaluation of the same eval set as hint lists
    (cfg.eval_hints, default on).  Same metrics and same lists as without, evaluation after evaluation while the
    model moves (here: towards the golden trained tables), also when the set of eval users changes under one key."""
    from yelprecommendation_amd.trainers import MFTrainer
    test_eval = _eval_frame(g["test_eval_users"], g["test_pos_ptr"], g["test_pos_idx"], g["test_mask_ptr"], g["test_mask_idx"])
    got = {}
    for hints in (True, False):
        cfg = _cfg(g, tmp_path)
        cfg.eval_hints = hints
        torch.manual_seed(4)
        t = MFTrainer(cfg, int(g["num_items"]), int(g["num_users"]))
        U0, I0 = t.model.user_embedding.weight.data.clone(), t.model.item_embedding.weight.data.clone()
        U1, I1 = torch.from_numpy(g["U_best"]).to(device), torch.from_numpy(g["I_best"]).to(device)
        runs = []
        for w in (0.0, 0.3, 0.35, 1.0, 1.0):
            t.model.user_embedding.weight.data.copy_((1 - w) * U0 + w * U1)
            t.model.item_embedding.weight.data.copy_((1 - w) * I0 + w * I1)
            metrics = t.evaluate(test_eval, "valid")
            _, users, mask_ptr, mask_idx = t._eval_arrays(test_eval)
            runs.append((metrics, t.recommend(users, mask_ptr, mask_idx, hint_key=id(test_eval)).cpu().numpy()))
        assert (len(t._eval_hints) == 1) == hints
        if hints:
            t._eval_hints[id(test_eval)] = t._eval_hints[id(test_eval)][:5]                      # a hint of another shape
        runs.append((t.evaluate(test_eval, "valid"), None))
        got[hints] = runs
    for (ma, la), (mb, lb) in zip(got[True], got[False]):
        assert ma == mb
        assert la is None or np.array_equal(la, lb)
