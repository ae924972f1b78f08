"""Whole steps at BASELINE's FULL sizes against the NumPy oracle (the other GPU tests check the same kernels at
sizes the oracle walks in milliseconds; the host-side tile / split-K / bucket choices differ at full size):

* configs[1]  BPR-MF 31,668 x 38,048, D = 64, B = 2^18, pull form: loss, and a few hundred user and item rows
  with all four Adam moments over two steps, against oracle/bpr_mf.py + oracle/adam.py restricted to the
  triplets that touch those rows;
* configs[3]  NGCF K = 3, D = 64 on the 69,716-node graph (ragged last 32-row tile: 69,716 = 2178 x 32 + 20):
  scores, loss, the embedding gradient of EVERY node and all six weight gradients against oracle/ngcf.py;
* configs[4]  CDAE I = 38,048, H = 128, B = 256: prediction, NS-BCE loss and one Adam step on all five
  parameters against oracle/cdae.py (both routes: autograd launch by launch, and the fused step of
  cdae_step.py).
"""
import numpy as np
import pytest
import torch

from oracle import adam as oadam
from oracle import bpr_mf as obpr
from oracle import cdae as ocdae
from oracle import ngcf as ongcf

pytestmark = pytest.mark.gpu

NU, NI, D = 31668, 38048, 64


def test_bpr_pull_step_full_size_sampled_rows_match_oracle(device):
    from yelprecommendation_amd.bpr_step import BPRMFStep
    rs = np.random.RandomState(11)
    B, lr = 1 << 18, 5e-3
    U = (rs.standard_normal((NU, D)) * 0.1).astype(np.float32)
    I = (rs.standard_normal((NI, D)) * 0.1).astype(np.float32)
    step = BPRMFStep(torch.from_numpy(U).to(device), torch.from_numpy(I).to(device), lr=lr, impl="pull")
    su = rs.choice(NU, 300, replace=False)
    si = np.r_[np.arange(8), rs.choice(NI, 292, replace=False)]        # incl. the most popular items
    mU, vU = np.zeros((300, D), np.float32), np.zeros((300, D), np.float32)
    mI, vI = np.zeros((300, D), np.float32), np.zeros((300, D), np.float32)
    total = 0.0
    for t in (1, 2):
        u = rs.randint(0, NU, B).astype(np.int64)
        p = np.minimum((rs.pareto(1.5, B) * 40).astype(np.int64), NI - 1)   # popularity-skewed positives
        n = rs.randint(0, NI, B).astype(np.int64)
        Uc, Ic = step.U.cpu().numpy(), step.I.cpu().numpy()
        total += float(obpr.bpr_loss(obpr.forward(Uc, Ic, u, p), obpr.forward(Uc, Ic, u, n)))
        # gradient rows of the sample from the triplets that touch them (the oracle's mean is over its
        # sub-batch: rescale to 1 / B)
        tu = np.isin(u, su)
        _, gU, _ = obpr.loss_and_grads(Uc, Ic, u[tu], p[tu], n[tu])
        ti = np.isin(p, si) | np.isin(n, si)
        _, _, gI = obpr.loss_and_grads(Uc, Ic, u[ti], p[ti], n[ti])
        gU, gI = gU[su] * np.float32(tu.sum() / B), gI[si] * np.float32(ti.sum() / B)
        wantU, wantI = Uc[su].copy(), Ic[si].copy()
        oadam.adam_update(wantU, gU.astype(np.float32), mU, vU, t, lr)
        oadam.adam_update(wantI, gI.astype(np.float32), mI, vI, t, lr)
        step.step(*(torch.from_numpy(a).to(device) for a in (u, p, n)))
        assert step.impl.startswith("pull")
        np.testing.assert_allclose(step.U.cpu().numpy()[su], wantU, rtol=1e-3, atol=1e-5)
        np.testing.assert_allclose(step.I.cpu().numpy()[si], wantI, rtol=1e-3, atol=1e-5)
        np.testing.assert_allclose(step.mU.cpu().numpy()[su], mU, rtol=1e-3, atol=1e-9)
        np.testing.assert_allclose(step.vU.cpu().numpy()[su], vU, rtol=1e-3, atol=1e-13)
        np.testing.assert_allclose(step.mI.cpu().numpy()[si], mI, rtol=1e-3, atol=1e-9)
        np.testing.assert_allclose(step.vI.cpu().numpy()[si], vI, rtol=1e-3, atol=1e-13)
    np.testing.assert_allclose(step.epoch_loss(), total, rtol=2e-5)
    step.check()


def test_ngcf_full_size_step_matches_oracle(device, tmp_path):
    from yelprecommendation_amd.data.synthetic import make_interactions_torch
    from yelprecommendation_amd.graph import LaplacianCSR
    from yelprecommendation_amd.loss import BPRLoss
    from yelprecommendation_amd.models.ngcf import NGCF
    from yelprecommendation_amd.utils import make_config
    u, i = make_interactions_torch(NU, NI, 47.0, device=device)
    r = torch.randint(1, 6, u.shape, device=device, generator=torch.Generator(device=device).manual_seed(3))
    un, inn, rn = u.cpu().numpy(), i.cpu().numpy(), r.cpu().numpy()
    graph = LaplacianCSR.from_interactions(un, inn, rn, NU, NI, device)
    L = ongcf.laplacian_csr(un, inn, rn, NU, NI)
    assert graph.n == NU + NI == 69716 and graph.n % 32 == 20
    torch.manual_seed(5)
    model = NGCF(make_config("NGCF", embed_size=D, num_orders=3, device="cuda", model_dir=str(tmp_path)), NU, NI).to(device)
    with torch.no_grad():
        model.embedding.weight.mul_(0.1)                        # N(0, 1) rows make |scores| ~ 30: saturated sigmoid
    names = [k for k, _ in model.named_parameters()]
    params = {k: v.detach().cpu().numpy().copy() for k, v in model.named_parameters()}
    W1s = [params[f"W1.{k}.weight"] for k in range(3)]
    W2s = [params[f"W2.{k}.weight"] for k in range(3)]
    rs = np.random.RandomState(9)
    B = 4096
    bu, bp, bn = rs.randint(0, NU, B), rs.randint(0, NI, B), rs.randint(0, NI, B)
    bu[:4], bp[:4] = NU - 1, NI - 1                              # the last rows of both halves (ragged tile)
    want_pos, want_neg = ongcf.bpr_forward(params["embedding.weight"], W1s, W2s, L, NU, bu, bp, bn)
    want_loss, dE, dW1, dW2 = ongcf.loss_and_grads(params["embedding.weight"], W1s, W2s, L, NU, bu, bp, bn)
    t = lambda a: torch.from_numpy(a.astype(np.int64)).to(device)
    pos, neg = model.bpr_forward(t(bu), t(bp), t(bn), graph)
    np.testing.assert_allclose(pos.detach().cpu().numpy(), want_pos, rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(neg.detach().cpu().numpy(), want_neg, rtol=1e-3, atol=1e-4)
    loss = BPRLoss()(pos, neg)
    np.testing.assert_allclose(loss.item(), float(want_loss), rtol=1e-4)
    loss.backward()
    want = {"embedding.weight": dE}                             # parameter names of models/ngcf.py:15-23
    want.update({f"W1.{k}.weight": dW1[k] for k in range(3)})
    want.update({f"W2.{k}.weight": dW2[k] for k in range(3)})
    assert sorted(want) == sorted(names)
    for name, prm in model.named_parameters():
        w = want[name]
        np.testing.assert_allclose(prm.grad.cpu().numpy(), w, rtol=2e-3, atol=1e-7 + 2e-4 * np.abs(w).max(), err_msg=name)
    # the last 20 rows (the ragged 32-row tile) and the first rows, explicitly
    g = model.embedding.weight.grad.cpu().numpy()
    assert np.abs(g[-20:]).max() > 0 and np.abs(dE[-20:] - g[-20:]).max() <= 2e-4 * np.abs(dE).max() + 1e-7
    model.check_indices()


def test_cdae_full_size_step_matches_oracle(device, tmp_path):
    from yelprecommendation_amd.loss import NSBCELoss
    from yelprecommendation_amd.models.cdae import CDAE
    from yelprecommendation_amd.optim import Adam
    from yelprecommendation_amd.utils import make_config
    rs = np.random.RandomState(4)
    H, B = 128, 256
    cfg = make_config("CDAE", hidden_size=H, device="cuda", model_dir=str(tmp_path), lr=1e-3)
    torch.manual_seed(2)
    model = CDAE(cfg, NI, NU)
    params = [p.detach().cpu().numpy().copy() for p in model.parameters()]
    ref = ocdae.CDAEState(params, lr=1e-3)
    u = rs.choice(NU, size=B, replace=False).astype(np.int64)
    x = (rs.rand(B, NI) < 0.0013).astype(np.float32)             # ~49 positives per user
    keep = (rs.rand(B, NI) >= 0.6).astype(np.float32)
    neg = ((rs.rand(B, NI) < 0.0065) * (1 - x)).astype(np.float32)   # neg_times = 5
    xin = x * keep * np.float32(2.5)
    want_pred = ref.predict(u, xin)
    want = float(ref.train_step(u, xin, x, neg))
    opt = Adam(model.parameters(), lr=1e-3)
    t = lambda a: torch.from_numpy(a).to(device)
    pred = model.encode_decode(t(u), t(xin))
    cols = np.r_[0:64, NI - 40:NI, rs.choice(NI, 400, replace=False)]   # first / last (ragged) tiles + a sample
    np.testing.assert_allclose(pred.detach().cpu().numpy()[:, cols], want_pred[:, cols], rtol=1e-4, atol=1e-6)
    loss = NSBCELoss()(pred, t(x), t(neg))
    opt.zero_grad()
    loss.backward()
    opt.step()
    np.testing.assert_allclose(loss.item(), want, rtol=1e-5)
    for (name, p), r in zip(model.named_parameters(), ref.params):
        np.testing.assert_allclose(p.detach().cpu().numpy(), r, rtol=1e-3, atol=2e-6, err_msg=name)
    # the fused step (cdae_step.py) from the same init on the same batch: same loss, same parameters
    from yelprecommendation_amd.cdae_step import CDAEStep
    for decoder in ("sampled", "dense"):
        torch.manual_seed(2)
        model2 = CDAE(cfg, NI, NU)
        step = CDAEStep(model2, Adam(model2.parameters(), lr=1e-3), decoder=decoder, transposed_wh=decoder == "sampled")
        step.step(t(u), t(x), t(neg), x_in=t(xin))
        step.release()
        np.testing.assert_allclose(float(step.last_loss()), want, rtol=1e-5)
        for (name, p), r in zip(model2.named_parameters(), ref.params):
            np.testing.assert_allclose(p.detach().cpu().numpy(), r, rtol=1e-3, atol=2e-6, err_msg=f"{decoder} {name}")
        step.check()


def test_evaluation_full_size_every_form_gives_the_same_lists(device):
    """BASELINE configs[1] size (31,668 users x 38,048 items, ~47 masked items per user, top-10): the fused
    evaluation in every form — f32 matrix instruction / three-term bf16 splits, prescan on / off, hint lists (own
    result and the result of perturbed tables), catalogue slices on / off, the two-role form of the sweep — must give the lists of the f32
    instruction without prescan; rows may differ between the two precisions only at float near-ties (examined one by
    one against float64 scores).  A sample of 512 users is checked against the oracle's per-user loop as well, and
    no list may hold a masked item."""
    from oracle import mf_eval
    from replay import assert_topk_equal_up_to_near_ties
    from yelprecommendation_amd import engine
    g = torch.Generator(device=device).manual_seed(17)
    U = torch.randn(NU, D, device=device, generator=g) * 0.1
    I = torch.randn(NI, D, device=device, generator=g) * 0.1
    users = torch.arange(NU, device=device)
    cnt = torch.randint(10, 85, (NU,), device=device, generator=g)
    ptr = torch.zeros(NU + 1, dtype=torch.int64, device=device)
    ptr[1:] = torch.cumsum(cnt, 0)
    idx = engine.sort_mask_rows(ptr, torch.randint(0, NI, (int(ptr[-1]),), device=device, generator=g))
    k = 10
    base = engine.mf_eval_topk(U, I, users, ptr, idx, k, precision="f32", prescan=False)
    for prescan in (True, False):
        assert torch.equal(engine.mf_eval_topk(U, I, users, ptr, idx, k, precision="f32", prescan=prescan), base)
    split = engine.mf_eval_topk(U, I, users, ptr, idx, k, precision="bf16x3", prescan=False)
    for kw in (dict(prescan=True), dict(hint=split), dict(hint=base), dict(sliced=False),
               dict(hint=engine.mf_eval_topk(U * 1.02 + 0.003, I, users, ptr, idx, k)),
               dict(form="two_roles", prescan=False), dict(form="two_roles", prescan=True),
               dict(form="two_roles", hint=base), dict(form="four_waves", hint=base)):
        assert torch.equal(engine.mf_eval_topk(U, I, users, ptr, idx, k, precision="bf16x3", **kw), split), kw
    Un, In, pn, xn = U.cpu().numpy(), I.cpu().numpy(), ptr.cpu().numpy(), idx.cpu().numpy()
    lists = [xn[pn[r]:pn[r + 1]] for r in range(NU)]
    ndiff = assert_topk_equal_up_to_near_ties(split.cpu().numpy(), base.cpu().numpy(), Un, In, np.arange(NU), lists, rel=2e-6)
    assert ndiff <= NU // 1000                                                 # near-ties are rare
    rows = np.random.RandomState(3).choice(NU, 512, replace=False)
    sub_ptr = np.zeros(len(rows) + 1, np.int64)
    sub_ptr[1:] = np.cumsum([len(lists[r]) for r in rows])
    want = mf_eval.recommend(Un, In, rows.astype(np.int64), sub_ptr, np.concatenate([lists[r] for r in rows]).astype(np.int64), k)
    assert_topk_equal_up_to_near_ties(split.cpu().numpy()[rows], want, Un, In, rows, [lists[r] for r in rows])
    got = split.cpu().numpy()
    for r in rows:
        assert not set(got[r].tolist()) & set(lists[r].tolist())
