"""GPU parity for CDAE: the f32 MFMA GEMM in all layouts, the model's forward / NS-BCE loss / every
parameter gradient on the golden probe batch, and a CDAETrainer run replaying the reference's
recorded batches (users, negative masks, dropout outcomes) — tests/golden/cdae_small.npz."""
import os

import numpy as np
import pytest
import torch

from oracle import cdae as ocdae

pytestmark = pytest.mark.gpu


# two captures of the reference: the small one (H = 16) and one with BASELINE configs[4]'s hidden size (H = 128) at the
# reference's default batch 32 (tests/golden/make_golden.py cdae)
@pytest.fixture(scope="module", params=["cdae_small.npz", "cdae_mid.npz"])
def g(golden_dir, request):
    return np.load(os.path.join(golden_dir, request.param))


@pytest.mark.parametrize("tA", [False, True])
@pytest.mark.parametrize("tB", [False, True])
@pytest.mark.parametrize("M,N,K,split", [(1, 1, 1, 1), (32, 128, 167, 1), (70, 167, 33, 1), (24, 16, 1000, 4),
                                         (167, 16, 24, 1), (130, 65, 64, 2)])
def test_gemm_f32_all_layouts(device, tA, tB, M, N, K, split):
    from yelprecommendation_amd import engine
    rs = np.random.RandomState(M + 3 * N + 7 * K)
    A = rs.standard_normal((K, M) if tA else (M, K)).astype(np.float32)      # asymmetric operands
    B = rs.standard_normal((N, K) if tB else (K, N)).astype(np.float32)
    want = (A.T if tA else A) @ (B.T if tB else B)
    dA, dB = torch.from_numpy(A).to(device), torch.from_numpy(B).to(device)
    got = engine.gemm_f32(dA, dB, transA=tA, transB=tB, split_k=split)
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-4, atol=1e-4)
    if split == 1:
        bias = rs.standard_normal(N).astype(np.float32)
        got = engine.gemm_f32(dA, dB, transA=tA, transB=tB, bias=torch.from_numpy(bias).to(device),
                              act=engine.ACT_SIGMOID)
        np.testing.assert_allclose(got.cpu().numpy(), 1 / (1 + np.exp(-(want + bias))), rtol=1e-4, atol=1e-5)
    acc0 = rs.standard_normal((M, N)).astype(np.float32)
    acc = torch.from_numpy(acc0.copy()).to(device)
    engine.gemm_f32(dA, dB, transA=tA, transB=tB, out=acc, accumulate=True, split_k=split)
    np.testing.assert_allclose(acc.cpu().numpy(), acc0 + want, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("tA", [False, True])
@pytest.mark.parametrize("tB", [False, True])
@pytest.mark.parametrize("M,N,K,split", [(6144, 6144, 40, 1),       # 128 x 128 workgroup tiles
                                         (8200, 2048, 72, 1),       # 64 x 128 (+ ragged last row tile)
                                         (2048, 8201, 72, 1),       # 128 x 64 (+ ragged last column tile)
                                         (300, 260, 136, 1),        # 64 x 64, K a multiple of neither 16 nor 32
                                         (256, 128, 5000, 8),       # split-K, atomic epilogue
                                         (129, 131, 36, 1)])        # unaligned leading dimensions: generic kernel
def test_gemm_f32_tiled_variants(device, tA, tB, M, N, K, split):
    """Every workgroup-tile variant of the double-buffered kernel (the host picks the tile from the
    problem size) in all four layouts, against float64 NumPy; strided views exercise lda > K."""
    from yelprecommendation_amd import engine
    rs = np.random.RandomState(M + 3 * N + 7 * K + 2 * tA + tB)
    pad = 8                                                       # operands are views into wider buffers
    Af = rs.standard_normal((K, M + pad) if tA else (M, K + pad)).astype(np.float32)
    Bf = rs.standard_normal((N, K + pad) if tB else (K, N + pad)).astype(np.float32)
    dAf, dBf = torch.from_numpy(Af).to(device), torch.from_numpy(Bf).to(device)
    dA = dAf[:, :M] if tA else dAf[:, :K]
    dB = dBf[:, :K] if tB else dBf[:, :N]
    A, B = dA.cpu().numpy().astype(np.float64), dB.cpu().numpy().astype(np.float64)
    want = (A.T if tA else A) @ (B.T if tB else B)
    tol = dict(rtol=1e-4, atol=2e-5 * np.sqrt(K) * 4)
    lib_gemm = lambda **kw: _gemm_views(engine, dA, dB, tA, tB, M, N, K, **kw)
    got = lib_gemm(split_k=split)
    np.testing.assert_allclose(got.cpu().numpy(), want, **tol)
    if split == 1:
        bias = rs.standard_normal(N).astype(np.float32)
        got = lib_gemm(bias=torch.from_numpy(bias).to(device), act=engine.ACT_SIGMOID)
        np.testing.assert_allclose(got.cpu().numpy(), 1 / (1 + np.exp(-(want + bias))), rtol=1e-4, atol=1e-5)


def _gemm_views(engine, dA, dB, tA, tB, M, N, K, bias=None, act=0, split_k=1):
    """engine.gemm_f32 insists on contiguous tensors; strided row-major views go through the C ABI
    directly (the ABI takes leading dimensions)."""
    from yelprecommendation_amd import _lib
    lib = _lib.load()
    out = (torch.zeros if split_k > 1 else torch.empty)((M, N), dtype=torch.float32, device=dA.device)
    rc = lib.yr_gemm_f32(int(tA), int(tB), M, N, K, dA.data_ptr(), dA.stride(0), dB.data_ptr(), dB.stride(0),
                         out.data_ptr(), out.stride(0), bias.data_ptr() if bias is not None else None, int(act), 0,
                         int(split_k), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    return out


def _cfg(g, tmp_path, **kw):
    from yelprecommendation_amd.utils import make_config
    c = make_config("CDAE", hidden_size=int(g["hidden_size"]), lr=float(g["lr"]), batch_size=int(g["batch_size"]),
                    corruption_level=float(g["corruption_level"]), neg_times=int(g["neg_times"]), loss_name="bce",
                    device="cuda", model_dir=str(tmp_path))
    c.update(kw)
    return c


def _load(model, g, prefix):
    with torch.no_grad():
        for name, p in model.named_parameters():
            p.copy_(torch.from_numpy(g[f"{prefix}__{name.replace('.', '__')}"]))


def test_probe_forward_loss_grads_match_reference(g, tmp_path, device):
    from yelprecommendation_amd.loss import NSBCELoss
    from yelprecommendation_amd.models.cdae import CDAE
    model = CDAE(_cfg(g, tmp_path), int(g["num_items"]), int(g["num_users"]))
    assert [n for n, _ in model.named_parameters()] == g["param_names"].tolist()
    _load(model, g, "init")
    model.eval()                                             # dropout off, as in the golden probe
    u = torch.from_numpy(g["probe_user"]).to(device)
    x = torch.from_numpy(g["probe_x"].astype(np.float32)).to(device)
    neg = torch.from_numpy(g["probe_neg"].astype(np.float32)).to(device)
    pred = model(u, x)
    np.testing.assert_allclose(pred.detach().cpu().numpy(), g["probe_pred"], rtol=1e-5, atol=1e-6)
    loss = NSBCELoss()(pred, x, neg)
    np.testing.assert_allclose(loss.item(), float(g["probe_loss"]), rtol=1e-5)
    loss.backward()
    for name, p in model.named_parameters():
        want = g[f"grad__{name.replace('.', '__')}"]
        np.testing.assert_allclose(p.grad.cpu().numpy(), want, rtol=1e-3, atol=1e-7 + 1e-4 * np.abs(want).max())
    model.check_indices()


def test_dropout_statistics_and_eval_identity(g, tmp_path, device):
    from yelprecommendation_amd.models.cdae import CDAE
    model = CDAE(_cfg(g, tmp_path), int(g["num_items"]), int(g["num_users"]))
    x = torch.ones(200, int(g["num_items"]), device=device)
    model.train()
    y = model.add_noise(x)
    kept = (y != 0).float().mean().item()
    assert abs(kept - 0.4) < 0.02 and set(torch.unique(y).tolist()) <= {0.0, 2.5}     # p = 0.6, scale 1/(1-p)
    model.eval()
    assert model.add_noise(x) is x


def test_seeded_dropout_is_a_fair_reproducible_mask(device):
    """yr_dropout_seeded (Philox in the kernel): same (seed, shape) -> same mask, another seed ->
    another mask, keep rate 1 - p in every row and column block, no correlation between neighbouring
    elements, ragged sizes (scalar tail), and the explicit-uniform kernel's semantics (x / (1 - p))."""
    from yelprecommendation_amd import engine
    from yelprecommendation_amd.utils import set_seed
    x = torch.full((512, 4099), 3.0, device=device)                      # 4099: tail of 3 elements per ... flat index
    a = engine.dropout_seeded(x, 1234, 0.6)
    b = engine.dropout_seeded(x, 1234, 0.6)
    c = engine.dropout_seeded(x, 1235, 0.6)
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert set(torch.unique(a).tolist()) <= {0.0, 7.5}
    keep = (a != 0).float()
    assert abs(keep.mean().item() - 0.4) < 0.005
    assert (keep.mean(1) - 0.4).abs().max().item() < 0.05 and (keep.mean(0) - 0.4).abs().max().item() < 0.1
    k = keep.flatten()
    for lag in (1, 2, 3, 4, 4099):
        corr = ((k[:-lag] - 0.4) * (k[lag:] - 0.4)).mean().item() / 0.24
        assert abs(corr) < 0.01, (lag, corr)
    small = torch.ones(7, device=device)                                   # n < 4 groups, only tails
    assert set(torch.unique(engine.dropout_seeded(small, 5, 0.5)).tolist()) <= {0.0, 2.0}
    # the model draws its seed from torch's generator: set_seed pins the masks
    from yelprecommendation_amd.models.cdae import CDAE
    from yelprecommendation_amd.utils import make_config
    model = CDAE(make_config("CDAE", hidden_size=8, device="cuda", model_dir="/tmp/yr_cdae_do"), 4099, 10)
    model.train()
    set_seed(7); m1 = model.add_noise(x)
    set_seed(7); m2 = model.add_noise(x)
    m3 = model.add_noise(x)
    assert torch.equal(m1, m2) and not torch.equal(m1, m3)


@pytest.mark.parametrize("fused_step", ["sampled", "dense", False])
def test_trainer_run_matches_reference(g, tmp_path, device, fused_step):
    """The reference's recorded run, once through the fused step (cdae_step.py, the default) and once launch by
    launch through autograd (model node + loss module + optimizer.step)."""
    from yelprecommendation_amd.trainers import CDAETrainer
    t = CDAETrainer(_cfg(g, tmp_path, negative_sampling=True, fused_step=bool(fused_step),
                         train_decoder=fused_step or "auto"), int(g["num_items"]), int(g["num_users"]))
    assert (t._fused_step().decoder if fused_step else t._fused_step()) == (fused_step or None)
    _load(t.model, g, "init")
    X = g["train_input"].astype(np.float32)
    VM = g["valid_mask"].astype(np.float32)
    tpos = vpos = tb = vb = 0
    for e, (ns, nv) in enumerate(zip(g["train_steps"], g["valid_steps"])):
        batches, corrupted = [], []
        for b in g["train_batch_sizes"][tb:tb + ns]:
            s = slice(tpos, tpos + int(b)); tpos += int(b)
            u = g["train_user"][s].astype(np.int64)
            batches.append({"user_id": torch.from_numpy(u), "input_mask": torch.from_numpy(X[u]),
                            "negative_mask": torch.from_numpy(g["train_neg"][s].astype(np.float32))})
            corrupted.append(torch.from_numpy(g["train_keep"][s].astype(np.float32) * np.float32(2.5)))
        tb += ns
        it = iter(corrupted)
        t.model.add_noise = lambda x: next(it).to(x.device)              # the reference's recorded dropout outcomes
        train_loss = t.train(batches)
        del t.model.add_noise
        np.testing.assert_allclose(train_loss, g["train_epoch"][e], rtol=1e-4)
        vbatches = []
        for b in g["valid_batch_sizes"][vb:vb + nv]:
            s = slice(vpos, vpos + int(b)); vpos += int(b)
            u = g["valid_user"][s].astype(np.int64)
            vbatches.append({"user_id": torch.from_numpy(u), "input_mask": torch.from_numpy(X[u]),
                             "valid_mask": torch.from_numpy(VM[u]),
                             "negative_mask": torch.from_numpy(g["valid_neg"][s].astype(np.float32))})
        vb += nv
        out = t.validate(vbatches)
        np.testing.assert_allclose(out[0], g["valid_epoch"][e][0], rtol=1e-4)
        np.testing.assert_allclose(out[1:], g["valid_epoch"][e][1:], atol=1e-3, rtol=0)
    for name, p in t.model.named_parameters():
        np.testing.assert_allclose(p.detach().cpu().numpy(), g[f"final__{name.replace('.', '__')}"], rtol=1e-3, atol=1e-5)
    # test split: evaluate() with the reference's masks
    users = np.arange(int(g["num_users"]), dtype=np.int64)
    tbatches = [{"user_id": torch.from_numpy(users[i:i + 32]),
                 "input_mask": torch.from_numpy(g["test_input"][i:i + 32].astype(np.float32)),
                 "test_mask": torch.from_numpy(g["test_mask"][i:i + 32].astype(np.float32))}
                for i in range(0, len(users), 32)]
    dev_metrics = t.evaluate(tbatches)
    np.testing.assert_allclose(dev_metrics, g["test_metrics"], atol=1e-3, rtol=0)
    # the device-side metric sums against the reference's host route (np.nonzero + metric.py loops)
    t.cfg.host_metrics = True
    np.testing.assert_allclose(t.evaluate(tbatches), dev_metrics, rtol=1e-12)
    np.testing.assert_allclose(t.validate(vbatches)[1:], out[1:], rtol=1e-12)


def test_config5_shape_step_matches_oracle(device, tmp_path):
    """BASELINE configs[4] shape in miniature (hidden 128, ragged catalogue, batch 40): one Adam step
    against the NumPy oracle."""
    from yelprecommendation_amd.loss import NSBCELoss
    from yelprecommendation_amd.models.cdae import CDAE
    from yelprecommendation_amd.optim import Adam
    from yelprecommendation_amd.utils import make_config
    rs = np.random.RandomState(2)
    nu, ni, H, B = 90, 1501, 128, 40
    cfg = make_config("CDAE", hidden_size=H, device="cuda", model_dir=str(tmp_path), lr=1e-3)
    model = CDAE(cfg, ni, nu)
    params = [p.detach().cpu().numpy().copy() for p in model.parameters()]
    ref = ocdae.CDAEState(params, lr=1e-3)
    u = rs.choice(nu, size=B, replace=False).astype(np.int64)
    x = (rs.rand(B, ni) < 0.02).astype(np.float32)
    keep = (rs.rand(B, ni) >= 0.6).astype(np.float32)
    neg = ((rs.rand(B, ni) < 0.1) * (1 - x)).astype(np.float32)
    want = float(ref.train_step(u, x * keep * np.float32(2.5), x, neg))
    opt = Adam(model.parameters(), lr=1e-3)
    t = lambda a: torch.from_numpy(a).to(device)
    pred = model.encode_decode(t(u), t(x * keep * np.float32(2.5)))
    loss = NSBCELoss()(pred, t(x), t(neg))
    opt.zero_grad()
    loss.backward()
    opt.step()
    np.testing.assert_allclose(loss.item(), want, rtol=1e-5)
    for p, r in zip(model.parameters(), ref.params):
        np.testing.assert_allclose(p.detach().cpu().numpy(), r, rtol=1e-3, atol=2e-6)


def test_sparse_encoder_equals_dense_encoder(device, tmp_path):
    """The sparse-input form of the encoder (row compaction with the dropout mask applied on the fly,
    gather encoder, scatter dW_h) against the dense GEMM form on the same inputs: compacted rows ==
    yr_dropout_seeded's output exactly, outputs and ALL parameter gradients equal to rounding; ragged
    catalogue width, an all-zero row, a dense-ish row, duplicate users."""
    from yelprecommendation_amd import engine
    from yelprecommendation_amd.loss import NSBCELoss
    from yelprecommendation_amd.models.cdae import CDAE
    from yelprecommendation_amd.utils import make_config
    rs = np.random.RandomState(6)
    nu, ni, H, B = 70, 1503, 128, 37
    x = (rs.rand(B, ni) < 0.03).astype(np.float32)
    x[3] = 0.0
    x[5] = (rs.rand(ni) < 0.6).astype(np.float32)
    neg = ((rs.rand(B, ni) < 0.1) * (1 - x)).astype(np.float32)
    users = rs.randint(0, nu, B).astype(np.int64); users[7] = users[2]
    t = lambda a: torch.from_numpy(a).to(device)
    for p, seed in ((0.0, 0), (0.6, 123456789012345)):
        rows = engine.SparseRows(t(x), seed, p)
        want = engine.dropout_seeded(t(x), seed, p) if p > 0 else t(x)
        assert torch.equal(rows.to_dense(), want)                                  # same mask, same scale
        cols = rows.row_columns(5)
        assert bool((cols[1:] > cols[:-1]).all()) and rows.row_columns(3).numel() == 0   # ascending, empty row
    grads = {}
    for sparse in (True, False):
        torch.manual_seed(11)
        model = CDAE(make_config("CDAE", hidden_size=H, device="cuda", model_dir=str(tmp_path), sparse_encoder=sparse), ni, nu)
        assert model.sparse_encoder == sparse
        model.train()
        torch.manual_seed(5)                                                       # same dropout seed draw
        pred = model(t(users), t(x))
        NSBCELoss()(pred, t(x), t(neg)).backward()
        grads[sparse] = (pred.detach(), [q.grad.clone() for q in model.parameters()])
        model.eval()
        grads[sparse] += (model(t(users), t(x)).detach(),)
    torch.testing.assert_close(grads[True][0], grads[False][0], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(grads[True][2], grads[False][2], rtol=1e-5, atol=1e-6)
    for a, b in zip(grads[True][1], grads[False][1]):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-7 + 1e-5 * float(b.abs().max()))


def test_device_batches_match_host_definitions(device):
    """yr_csr_rows_to_dense / yr_negative_mask behind data/cdae_batches.py: dense rows identical to
    the torch (CPU) construction; negative masks with the reference's law — exact count, never a
    positive, uniform over the non-positives, reproducible per seed — on ragged catalogue sizes."""
    from yelprecommendation_amd import engine
    from yelprecommendation_amd.data.cdae_batches import CDAEBatchLoader, CDAEInteractions
    rs = np.random.RandomState(0)
    nu, ni = 300, 4099
    u = np.repeat(np.arange(nu), 25); i = rs.randint(0, ni, size=u.shape[0])
    cpu = CDAEInteractions.from_interactions(torch.from_numpy(u), torch.from_numpy(i), nu, ni, seed=3, device="cpu")
    gpu = CDAEInteractions(nu, ni, {k: cpu._csr[k] for k in CDAEInteractions.PARTS}, device)
    users = torch.from_numpy(rs.permutation(nu)[:77].astype(np.int64))
    from oracle import cdae_batches as ocb
    for part in ("train", "valid", "test", "train_valid"):
        want = sum(ocb.dense_rows(*cpu.csr(p), users, ni) for p in (("train", "valid") if part == "train_valid" else (part,)))
        torch.testing.assert_close(gpu.dense(part, users.to(device)).cpu(), want.clamp(max=1.0), rtol=0, atol=0)
    for mode in ("train", "valid"):
        batches = list(CDAEBatchLoader(gpu, mode, batch_size=64, neg_times=5, seed=11))
        again = list(CDAEBatchLoader(gpu, mode, batch_size=64, neg_times=5, seed=11))
        other = list(CDAEBatchLoader(gpu, mode, batch_size=64, neg_times=5, seed=12))
        assert sum(b["user_id"].numel() for b in batches) == nu
        for b, a, o in zip(batches, again, other):
            pos = b["input_mask"] + (b["valid_mask"] if mode == "valid" else 0)
            neg = b["negative_mask"]
            assert set(neg.unique().tolist()) <= {0.0, 1.0}
            assert float((neg * pos).sum().item()) == 0.0
            torch.testing.assert_close(neg.sum(1), 5 * pos.sum(1))
            assert torch.equal(neg, a["negative_mask"]) and not torch.equal(neg, o["negative_mask"])
    # uniformity over the non-positives (one row, many seeds) and the two edge cases
    pos = torch.zeros(1, 133, device=device); pos[0, ::19] = 1                    # 7 positives, 126 non-positives
    hits = sum(engine.negative_mask(pos, 3, seed) for seed in range(3000))[0]
    assert float((hits * pos[0]).sum().item()) == 0.0
    freq = hits[pos[0] == 0] / 3000.0                                              # expected 21 / 126
    assert float((freq - 21 / 126).abs().max().item()) < 0.035
    assert float(engine.negative_mask(torch.zeros(2, 50, device=device), 4, 1).sum().item()) == 0.0
    flag = engine.new_error_flag(device)
    crowded = torch.ones(1, 10, device=device); crowded[0, 0] = 0
    out = engine.negative_mask(crowded, 2, 1, err_flag=flag)
    assert int(flag.item()) == engine.FLAG_BAD_ITEM and float(out.sum().item()) == 1.0
    loader = CDAEBatchLoader(CDAEInteractions(1, 10, {"train": (torch.tensor([0, 9]), torch.arange(1, 10)),
                                                       "valid": (torch.tensor([0, 0]), torch.zeros(0, dtype=torch.int64)),
                                                       "test": (torch.tensor([0, 0]), torch.zeros(0, dtype=torch.int64))},
                                              device), "train", 4, neg_times=2)
    with pytest.raises(ValueError):
        list(loader)


def test_user_indexed_item_lists_equal_mask_derived_lists(device, tmp_path, g):
    """The CDAE trainer's metric sums through the loader's per-user CSR (yr_topk_masked / yr_rank_metrics
    with row indirection) equal the ones derived from the dense masks with nonzero(), in valid and test
    mode; and the plain kernels with an explicit row map equal the re-packed CSR."""
    from yelprecommendation_amd import engine
    from yelprecommendation_amd.data.cdae_batches import CDAEBatchLoader, CDAEInteractions
    from yelprecommendation_amd.trainers import CDAETrainer
    rs = np.random.RandomState(2)
    nu, ni = 90, 700
    u = np.repeat(np.arange(nu), 20); i = rs.randint(0, ni, size=u.shape[0])
    data = CDAEInteractions.from_interactions(torch.from_numpy(u), torch.from_numpy(i), nu, ni, seed=1, device=device)
    t = CDAETrainer(_cfg(g, tmp_path, negative_sampling=True), ni, nu)
    for mode, key in (("valid", "valid_mask"), ("test", "test_mask")):
        for batch in CDAEBatchLoader(data, mode, batch_size=32, neg_times=1, shuffle=True, seed=3):
            pred = torch.rand(batch["user_id"].numel(), ni, device=device) + 0.01
            a = t._metric_sums(pred, batch[key], batch["input_mask"], batch["item_lists"])
            b = t._metric_sums(pred, batch[key], batch["input_mask"], None)
            torch.testing.assert_close(a, b, rtol=1e-12, atol=0)
    # row map on the raw kernels: rows listed twice / out of order
    ptr, idx = data.csr("train")
    rows = torch.tensor([5, 5, 0, 89, 17], device=device)
    scores = torch.rand(5, ni, device=device)
    got = engine.topk_masked(scores, ptr, idx, 10, mask_rows=rows)
    cnt = (ptr[1:] - ptr[:-1])[rows]
    p2 = torch.zeros(6, dtype=torch.int64, device=device); p2[1:] = torch.cumsum(cnt, 0)
    i2 = torch.cat([idx[ptr[r]:ptr[r + 1]] for r in rows.tolist()])
    assert torch.equal(got, engine.topk_masked(scores, p2, i2, 10))
    torch.testing.assert_close(engine.rank_metrics(got, ptr, idx, pos_rows=rows), engine.rank_metrics(got, p2, i2), rtol=1e-12, atol=0)


@pytest.mark.parametrize("tA", [False, True])
@pytest.mark.parametrize("tB", [False, True])
@pytest.mark.parametrize("M,N,K,split", [(300, 128, 136, 1), (2048, 8200, 72, 1), (256, 128, 5000, 8)])
def test_gemm_ex_count_scale_and_row_sums(device, tA, tB, M, N, K, split):
    """yr_gemm_f32_ex: the product scaled by 1 / (a count that lives on the device) and, without a K split,
    the row sums of op(A) from the same pass; a zero count scales by 0 (the empty-loss convention)."""
    from yelprecommendation_amd import engine
    rs = np.random.RandomState(5 * M + N + K + 2 * tA + tB)
    A = rs.standard_normal((K, M) if tA else (M, K)).astype(np.float32)
    B = rs.standard_normal((N, K) if tB else (K, N)).astype(np.float32)
    opA = (A.T if tA else A).astype(np.float64)
    want = opA @ (B.T if tB else B).astype(np.float64)
    dA, dB = torch.from_numpy(A).to(device), torch.from_numpy(B).to(device)
    for c in (7, 0):
        count = engine.spread_count(c, device)
        scale = 1.0 / c if c else 0.0
        rowsum = torch.full((M,), -1.0, dtype=torch.float32, device=device) if split == 1 else None
        out = torch.zeros(M, N, dtype=torch.float32, device=device)
        engine.gemm_f32(dA, dB, transA=tA, transB=tB, out=out, accumulate=split > 1, split_k=split,
                        alpha_count=count, rowsum=rowsum)
        np.testing.assert_allclose(out.cpu().numpy(), want * scale, rtol=1e-4, atol=2e-5 * np.sqrt(K) * 4)
        if rowsum is not None:
            np.testing.assert_allclose(rowsum.cpu().numpy(), opA.sum(1) * scale, rtol=1e-4, atol=1e-4)
    with pytest.raises(engine.EngineError):                       # row sums need the whole of K in one pass
        engine.gemm_f32(dA, dB, transA=tA, transB=tB, out=out, accumulate=True, split_k=4,
                        alpha_count=count, rowsum=torch.zeros(M, device=device))


def test_adam_flat_equals_adam_dense(device):
    """One launch over five tensors of any size == five yr_adam_dense launches, bit for bit; marked rows: only
    their gradient is read (stale values elsewhere are ignored), and it is cleared with the mark."""
    from yelprecommendation_amd import engine
    rs = np.random.RandomState(8)
    shapes = [(128, 1503), (128,), (300, 128), (1503, 128), (1503,), (7,), (3, 5)]
    mk = lambda s, scale=1.0: torch.from_numpy((rs.standard_normal(s) * scale).astype(np.float32)).to(device)
    for decoupled, wd in ((False, 0.0), (False, 0.01), (True, 0.01)):
        P, G = [mk(s) for s in shapes], [mk(s, 0.1) for s in shapes]
        M, V = [mk(s, 0.01) for s in shapes], [mk(s, 0.01).abs() for s in shapes]
        marks = torch.zeros(300, dtype=torch.uint8, device=device)
        hit = torch.from_numpy(rs.choice(300, 40, replace=False)).to(device)
        marks[hit] = 1
        Gd = [g.clone() for g in G]
        Gd[2] = torch.zeros_like(G[2]); Gd[2][hit] = G[2][hit]          # what the marked form stands for
        want = [(p.clone(), m.clone(), v.clone()) for p, m, v in zip(P, M, V)]
        for (p, m, v), g in zip(want, Gd):
            flat = [t.reshape(-1) for t in (p, g.clone(), m, v)]
            if flat[0].numel() % 4 == 0:
                engine.adam_dense(*flat, 3, 1e-2, 0.9, 0.999, 1e-8, wd, decoupled=decoupled)
            else:
                engine.adam_dense_multi([tuple(flat)], 3, 1e-2, 0.9, 0.999, 1e-8, wd, decoupled=decoupled)
        tensors = [(p, g, m, v, marks if k == 2 else None, {0: 1, 1: 2}.get(k, 0))
                   for k, (p, g, m, v) in enumerate(zip(P, G, M, V))]
        engine.adam_dense_flat(tensors, 3, 1e-2, 0.9, 0.999, 1e-8, wd, decoupled=decoupled)
        for k, ((p, m, v), gp, gm, gv) in enumerate(zip(want, P, M, V)):
            assert torch.equal(p, gp) and torch.equal(m, gm) and torch.equal(v, gv), k
        assert int(marks.sum()) == 0 and float(G[2][hit].abs().sum()) == 0.0        # consumed: cleared + unmarked
        assert float(G[0].abs().sum()) == 0.0 and float(G[1].abs().sum()) == 0.0 \
            and float(G[3].abs().sum()) > 0.0                                        # clear modes 1, 2 / left alone


@pytest.mark.parametrize("ni,negative_sampling,decoder,H", [(1501, True, "dense", 128), (1504, True, "sampled", 128),
                                                            (1501, True, "sampled", 128), (1503, False, "dense", 128),
                                                            (1001, True, "sampled", 32), (1001, True, "sampled", 256),
                                                            (1001, True, "dense", 100)])
def test_fused_step_equals_autograd_route(device, tmp_path, ni, negative_sampling, decoder, H):
    """cdae_step.CDAEStep — dense decoder (loss in the GEMM epilogue, count-scaled gradient products) and sampled
    decoder (forward, loss and decoder gradients on the loss positions only), one Adam launch —
    against the autograd route (model + loss module + optimizer.step) over four steps from the same init, with
    the same dropout seeds: losses, all parameters, all Adam moments.  Odd catalogue widths run with the transposed
    working copy of W_h (released and re-acquired in the middle of the run).  Ragged catalogue widths, duplicate users,
    an all-zero row; NS-BCE and plain BCE; then both against the NumPy oracle for the first step."""
    from yelprecommendation_amd.cdae_step import CDAEStep
    from yelprecommendation_amd.loss import BCELoss, NSBCELoss
    from yelprecommendation_amd.models.cdae import CDAE
    from yelprecommendation_amd.optim import Adam
    from yelprecommendation_amd.utils import make_config
    rs = np.random.RandomState(ni)
    nu, B, steps = 90, 40, 4
    t = lambda a: torch.from_numpy(a).to(device)
    batches = []
    for _ in range(steps):
        u = rs.randint(0, nu, B).astype(np.int64); u[7] = u[2]
        x = (rs.rand(B, ni) < 0.02).astype(np.float32); x[3] = 0.0
        neg = ((rs.rand(B, ni) < 0.1) * (1 - x)).astype(np.float32)
        batches.append((u, x, neg, int(rs.randint(1, 1 << 40))))
    out = {}
    for fused in (False, True):
        torch.manual_seed(3)
        model = CDAE(make_config("CDAE", hidden_size=H, device="cuda", model_dir=str(tmp_path), lr=1e-3), ni, nu)
        model.train()
        opt = Adam(model.parameters(), lr=1e-3)
        losses = []
        if fused:
            step = CDAEStep(model, opt, negative_sampling, decoder=decoder, transposed_wh=ni % 2 == 1)
            assert step.decoder == decoder
            for k, (u, x, neg, seed) in enumerate(batches):
                step.step(t(u), t(x), t(neg) if negative_sampling else None, seed=seed, p=model.corruption_level)
                losses.append(float(step.last_loss()))
                if k == 1:
                    step.release()                              # back to the module mid-run, re-acquired by the next step
            step.release()
            assert abs(step.epoch_loss() - sum(losses)) < 1e-5 and float(step.dV.abs().sum()) == 0.0 \
                and (step.dWh is None or float(step.dWh.abs().sum()) == 0.0)
            if decoder == "sampled":                                  # consumed gradients are cleared, marks reset
                assert float(step.dWo.abs().sum()) == 0.0 and float(step.dbo.abs().sum()) == 0.0
            step.check()
        else:
            lossf = NSBCELoss() if negative_sampling else BCELoss()
            for u, x, neg, seed in batches:
                xin = engine_dropout(t(x), seed, model.corruption_level)
                pred = model.encode_decode(t(u), xin)
                loss = lossf(pred, t(x), t(neg)) if negative_sampling else lossf(pred, t(x))
                opt.zero_grad(); loss.backward(); opt.step()
                losses.append(float(loss.detach()))
        out[fused] = (losses, [p.detach().clone() for p in model.parameters()],
                      [opt.state[p]["exp_avg"].clone() for p in model.parameters()],
                      [opt.state[p]["exp_avg_sq"].clone() for p in model.parameters()])
        assert all(opt.state[p]["step"] == steps for p in model.parameters())
    np.testing.assert_allclose(out[True][0], out[False][0], rtol=2e-6)
    for k in (1, 2, 3):
        for a, b in zip(out[True][k], out[False][k]):
            torch.testing.assert_close(a, b, rtol=2e-4, atol=1e-7 + 2e-5 * float(b.abs().max()))
    # and the first step against the oracle (recomputed from the same init)
    torch.manual_seed(3)
    model = CDAE(make_config("CDAE", hidden_size=H, device="cuda", model_dir=str(tmp_path), lr=1e-3), ni, nu)
    ref = ocdae.CDAEState([p.detach().cpu().numpy().copy() for p in model.parameters()], lr=1e-3)
    u, x, neg, seed = batches[0]
    xin = engine_dropout(t(x), seed, model.corruption_level).cpu().numpy()
    want = float(ref.train_step(u, xin, x, neg if negative_sampling else np.ones_like(x)))
    step = CDAEStep(model, Adam(model.parameters(), lr=1e-3), negative_sampling, decoder=decoder, transposed_wh=ni % 2 == 1)
    step.step(t(u), t(x), t(neg) if negative_sampling else None, seed=seed, p=model.corruption_level)
    step.release()
    np.testing.assert_allclose(float(step.last_loss()), want, rtol=1e-5)
    for p, r in zip(model.parameters(), ref.params):
        np.testing.assert_allclose(p.detach().cpu().numpy(), r, rtol=1e-3, atol=2e-6)


def engine_dropout(x, seed, p):
    from yelprecommendation_amd import engine
    return engine.dropout_seeded(x, seed, p) if p > 0 else x


def _csr(rs, nu, ni, counts):
    ptr = np.zeros(nu + 1, np.int64)
    ptr[1:] = np.cumsum(counts)
    idx = np.concatenate([np.sort(rs.choice(ni, c, replace=False)) for c in counts] or [np.zeros(0)]).astype(np.int64)
    return ptr, idx


@pytest.mark.parametrize("ni", [1501, 600, 70001])          # 70,001: a part spans more than one 2,048-column wave step
def test_train_lists_from_csr_have_the_reference_law(device, ni):
    """yr_cdae_train_lists (a training batch as lists straight from the per-user CSR): the encoder list is exactly
    what the dense route compacts from dropout_p(dense row); the loss list holds every positive (target 1) and
    exactly neg_times x as many distinct non-positives (target 0), reproducible per seed, uniform over the
    non-positives; rows with no items, rows that want more than half of the non-positives (the excluded items are
    drawn instead), repeated users; flags for a row that wants more negatives than exist and for a bad user."""
    from yelprecommendation_amd import engine
    rs = np.random.RandomState(ni)
    nu, neg_times, p = 40, 5, 0.6
    counts = rs.randint(1, 30, nu); counts[3] = 0; counts[5] = ni // 8          # 5 x (ni / 8) > half of the rest
    ptr, idx = _csr(rs, nu, ni, counts)
    users = rs.permutation(nu)[:24].astype(np.int64); users[:3] = (3, 5, 7); users[9] = users[10]
    t = lambda a: torch.from_numpy(a).to(device)
    flag = engine.new_error_flag(device)
    L = engine.TrainLists(t(ptr), t(idx), t(users), nu, ni, neg_times, 11, 987654321, p, err_flag=flag)
    x = np.zeros((len(users), ni), np.float32)
    for b, u in enumerate(users):
        x[b, idx[ptr[u]:ptr[u + 1]]] = 1.0
    want = engine.SparseRows(t(x), 987654321, p)                                  # the dense route's lists
    n = want.count.long()
    assert torch.equal(L.rows.count, want.count) and torch.equal(L.rows.to_dense(), want.to_dense())
    for b in range(len(users)):
        assert torch.equal(L.rows.row_columns(b), want.row_columns(b))
    target, neg = (a.cpu().numpy() for a in L.loss_dense())
    np.testing.assert_array_equal(target, x)                                      # every positive, nothing else
    assert float((neg * x).sum()) == 0.0
    np.testing.assert_array_equal(neg.sum(1), neg_times * x.sum(1))
    assert int(flag.item()) == 0
    again = engine.TrainLists(t(ptr), t(idx), t(users), nu, ni, neg_times, 11, 5, 0.0).loss_dense()[1].cpu().numpy()
    np.testing.assert_array_equal(again, neg)                                     # the negatives depend on neg_seed only
    other = engine.TrainLists(t(ptr), t(idx), t(users), nu, ni, neg_times, 12, 5, 0.0).loss_dense()[1].cpu().numpy()
    assert (other != neg).any()
    assert (neg[9] != neg[10]).any()                                              # the same user twice: two draws
    # one batch of a pool is alive at a time: L's storage now holds a later batch, and L says so instead of
    # handing out another batch's lists; a batch from another pool (what every CDAEBatchLoader owns) is untouched
    with pytest.raises(engine.EngineError, match="reused by a later batch"):
        L.loss_dense()
    own = {}
    M = engine.TrainLists(t(ptr), t(idx), t(users), nu, ni, neg_times, 11, 5, 0.0, pool=own)
    engine.TrainLists(t(ptr), t(idx), t(users), nu, ni, neg_times, 13, 5, 0.0)    # class-level pool
    np.testing.assert_array_equal(M.alive().loss_dense()[1].cpu().numpy(), neg)
    # uniform over the non-positives: 400 seeds on a small catalogue
    small_ptr, small_idx = _csr(rs, 2, 97, [6, 12])
    hits = np.zeros((2, 97))
    for seed in range(400):
        hits += engine.TrainLists(t(small_ptr), t(small_idx), t(np.arange(2, dtype=np.int64)), 2, 97, 3, seed, 0,
                                  0.0).loss_dense()[1].cpu().numpy()
    for b, k in enumerate((6, 12)):
        free = np.setdiff1d(np.arange(97), small_idx[small_ptr[b]:small_ptr[b + 1]])
        expect = 400 * 3 * k / len(free)
        chi2 = float(((hits[b, free] - expect) ** 2 / expect).sum())
        assert chi2 < len(free) + 5 * np.sqrt(2 * len(free)), (b, chi2)
        assert hits[b].sum() == 400 * 3 * k
    # flags: more negatives wanted than non-positives exist (np.random.choice raises there); a user out of range
    crowded_ptr, crowded_idx = _csr(rs, 1, 40, [10])
    C = engine.TrainLists(t(crowded_ptr), t(crowded_idx), t(np.zeros(1, np.int64)), 1, 40, 5, 1, 0, 0.0, err_flag=flag)
    tc, nc = (a.cpu().numpy() for a in C.loss_dense())
    assert int(flag.item()) & 2 and nc.sum() == 30 and float((tc * nc).sum()) == 0.0
    flag.zero_()
    engine.TrainLists(t(ptr), t(idx), t(np.array([nu + 3], np.int64)), nu, ni, neg_times, 1, 0, 0.0, err_flag=flag)
    assert int(flag.item()) & 1


def test_list_batches_train_like_dense_batches(device, tmp_path):
    """CDAEStep.step_lists on engine.TrainLists == CDAEStep.step on the dense row / negative mask the lists stand
    for (same dropout seed): losses and all parameters over three steps; then a whole trainer epoch over
    CDAEBatchLoader(lists=True)."""
    from yelprecommendation_amd import engine
    from yelprecommendation_amd.cdae_step import CDAEStep
    from yelprecommendation_amd.data.cdae_batches import CDAEBatchLoader, CDAEInteractions
    from yelprecommendation_amd.models.cdae import CDAE
    from yelprecommendation_amd.optim import Adam
    from yelprecommendation_amd.trainers import CDAETrainer
    from yelprecommendation_amd.utils import make_config
    rs = np.random.RandomState(12)
    nu, ni, H, B = 120, 1503, 64, 32
    ptr, idx = _csr(rs, nu, ni, rs.randint(0, 25, nu))
    t = lambda a: torch.from_numpy(a).to(device)
    cfg = make_config("CDAE", hidden_size=H, device="cuda", model_dir=str(tmp_path), lr=1e-3, negative_sampling=True,
                      neg_times=3, loss_name="bce", batch_size=B)
    out = {}
    for form in ("lists", "dense"):
        torch.manual_seed(4)
        model = CDAE(cfg, ni, nu); model.train()
        step = CDAEStep(model, Adam(model.parameters(), lr=1e-3))
        assert step.decoder == "sampled"
        losses = []
        for k in range(3):
            users = t(np.random.RandomState(k).permutation(nu)[:B].astype(np.int64))
            L = engine.TrainLists(t(ptr), t(idx), users, nu, ni, 3, 100 + k, 200 + k, model.corruption_level)
            if form == "lists":
                step.step_lists(users, L)
            else:
                x, neg = L.loss_dense()
                step.step(users, x, neg, seed=200 + k, p=model.corruption_level)
            losses.append(float(step.last_loss()))
        step.check()
        out[form] = (losses, [q.detach().clone() for q in model.parameters()])
    np.testing.assert_allclose(out["lists"][0], out["dense"][0], rtol=1e-6)
    for a, b in zip(out["lists"][1], out["dense"][1]):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-7)
    # the loader + trainer
    data = CDAEInteractions(nu, ni, {"train": (torch.from_numpy(ptr), torch.from_numpy(idx)),
                                     "valid": (torch.zeros(nu + 1, dtype=torch.int64), torch.zeros(0, dtype=torch.int64)),
                                     "test": (torch.zeros(nu + 1, dtype=torch.int64), torch.zeros(0, dtype=torch.int64))}, device)
    trainer = CDAETrainer(cfg, ni, nu)
    loader = CDAEBatchLoader(data, "train", batch_size=B, neg_times=3, shuffle=True, seed=1, lists=True,
                             dropout=trainer.model.corruption_level)
    first = trainer.train(loader)
    for _ in range(5):
        last = trainer.train(loader)
    assert np.isfinite(first) and last < first and trainer._fused_step().decoder == "sampled"


def test_list_route_of_validate_and_evaluate_equals_dense_route(device, tmp_path):
    """validate() / evaluate() over list batches — encoder per batch, NS-BCE terms on the loss positions, then ALL
    users scored at once by the fused evaluation kernel with the decoder bias (yr_mf_eval_topk_bias) — against the
    reference-shaped route over the dense batches the same lists stand for: same loss, same four metrics; the
    top-10 lists behind them agree row by row up to float near-ties."""
    from yelprecommendation_amd import engine
    from yelprecommendation_amd.data.cdae_batches import CDAEBatchLoader, CDAEInteractions
    from yelprecommendation_amd.trainers import CDAETrainer
    from yelprecommendation_amd.utils import make_config
    from replay import assert_topk_equal_up_to_near_ties
    rs = np.random.RandomState(3)
    nu, ni, H, B = 300, 1503, 64, 64
    t = lambda a: torch.from_numpy(a).to(device)
    parts = {}
    taken = np.zeros((nu, ni), bool)
    for name, hi in (("train", 30), ("valid", 8), ("test", 8)):
        counts = rs.randint(0, hi, nu)
        ptr = np.zeros(nu + 1, np.int64); ptr[1:] = np.cumsum(counts)
        idx = []
        for u_, c in enumerate(counts):
            free = np.flatnonzero(~taken[u_])
            pick = np.sort(rs.choice(free, c, replace=False))
            taken[u_, pick] = True
            idx.append(pick)
        parts[name] = (torch.from_numpy(ptr), torch.from_numpy(np.concatenate(idx).astype(np.int64)))
    data = CDAEInteractions(nu, ni, parts, device)
    cfg = make_config("CDAE", hidden_size=H, device="cuda", model_dir=str(tmp_path), lr=1e-2, negative_sampling=True,
                      neg_times=3, loss_name="bce", batch_size=B, top_n=10)
    trainer = CDAETrainer(cfg, ni, nu)
    trainer.train(CDAEBatchLoader(data, "train", batch_size=B, neg_times=3, shuffle=True, seed=1, lists=True,
                                  dropout=trainer.model.corruption_level))          # a few steps off the init
    for mode in ("valid", "test"):
        as_lists = CDAEBatchLoader(data, mode, batch_size=B, neg_times=3, seed=7, lists=True)
        dense = []
        for batch in CDAEBatchLoader(data, mode, batch_size=B, neg_times=3, seed=7, lists=True):
            users = batch["user_id"]
            d = {"user_id": users, "item_lists": batch["item_lists"]}
            if mode == "valid":
                target, neg = batch["lists"].loss_dense()
                d.update(input_mask=data.dense("train", users), valid_mask=data.dense("valid", users), negative_mask=neg.clone())
                assert torch.equal(target, d["input_mask"] + d["valid_mask"])        # positives of the loss list
                assert float((neg * target).sum()) == 0.0 and torch.equal(neg.sum(1), 3 * target.sum(1))
            else:
                d.update(input_mask=data.dense("train_valid", users), test_mask=data.dense("test", users))
            assert torch.equal(batch["lists"].rows.to_dense(), d["input_mask"])      # the encoder's input
            dense.append(d)
        if mode == "valid":
            got, want = trainer.validate(as_lists), trainer.validate(dense)
            np.testing.assert_allclose(got[0], want[0], rtol=1e-5)
            # as_lists went through CDAEBatchLoader.super_batches (8 batches per launch, cfg.eval_batch_group); one
            # launch set per batch gives the same lists (same seeds per batch), the same per-batch losses and metrics
            for grp in (1, 3):
                trainer.cfg.eval_batch_group = grp
                again = trainer.validate(CDAEBatchLoader(data, mode, batch_size=B, neg_times=3, seed=7, lists=True))
                np.testing.assert_allclose(again[0], got[0], rtol=2e-6)
                assert tuple(again[1:]) == tuple(got[1:])
            trainer.cfg.eval_batch_group = 8
            np.testing.assert_allclose(got[1:], want[1:], atol=1e-3, rtol=0)     # the project's bar for the metrics:
        else:                                                                    # a float near-tie at rank 10 / 11
            got, want = trainer.evaluate(as_lists), trainer.evaluate(dense)      # may move one membership
            np.testing.assert_allclose(got, want, atol=1e-3, rtol=0)
            # the rows per evaluation batch do not enter the result (train.py builds the test loader with larger ones)
            for rows in (7, nu):
                again = trainer.evaluate(CDAEBatchLoader(data, mode, batch_size=rows, neg_times=3, seed=7, lists=True))
                assert tuple(again) == tuple(got)
    # the lists behind the metrics: fused all-user top-10 against the per-batch masked top-k of the dense prediction
    model = trainer.model.eval()
    users = torch.arange(nu, device=device)
    x = data.dense("train_valid", users)
    with torch.no_grad():
        pred = model(users, x)
        z = engine.cdae_sparse_encode(engine.SparseRows(x), *(q.data for q in model._params()[:3]), users, model._hidden_act)
    sp, si = data.csr("train_valid")
    top_dense = engine.topk_masked(pred.contiguous(), sp, si, 10, mask_value=0.0)
    top_fused = engine.mf_eval_topk(z, model.output_layer.weight.data, users, sp, si, 10,
                                    item_bias=model.output_layer.bias.data)
    # scores as dot products of [z, 1] with [W_o, b_o] for the near-tie examination (float64 on the host)
    Ua = np.c_[z.cpu().numpy(), np.ones(nu, np.float32)]
    Ia = np.c_[model.output_layer.weight.data.cpu().numpy(), model.output_layer.bias.data.cpu().numpy()]
    spn, sin = sp.cpu().numpy(), si.cpu().numpy()
    masked = [sin[spn[r]:spn[r + 1]] for r in range(nu)]
    assert_topk_equal_up_to_near_ties(top_fused.cpu().numpy(), top_dense.cpu().numpy(), Ua, Ia, np.arange(nu), masked=masked)


@pytest.mark.parametrize("list_batches", [True, False])
def test_train_entry_point_with_device_side_batches(device, tmp_path, list_batches):
    """python -m yelprecommendation_amd.train model_name=CDAE synthetic=... fast_loader=true: pipeline -> sparse
    device store -> list (or dense) batches -> run() (train / validate / best model) -> evaluate(test)."""
    from yelprecommendation_amd import train
    metrics = train.main(["model_name=CDAE", "synthetic=300x900x14", "fast_loader=true", "epochs=3", "batch_size=64",
                          "hidden_size=64", "device=cuda", f"model_dir={tmp_path}", "lr=0.01", "loss_name=bce",
                          f"list_batches={'true' if list_batches else 'false'}"])
    assert len(metrics) == 4 and all(np.isfinite(m) and 0.0 <= m <= 1.0 for m in metrics)
    assert os.path.exists(os.path.join(str(tmp_path), "best_model.pt"))


def test_train_lists_with_an_empty_item_index(device):
    """No user has an item: the CSR's index tensor is empty (no address); every list comes out empty."""
    from yelprecommendation_amd import engine
    t = lambda a: torch.from_numpy(a).to(device)
    L = engine.TrainLists(t(np.zeros(6, np.int64)), t(np.zeros(0, np.int64)), t(np.arange(5, dtype=np.int64)), 5, 300, 5, 1, 2, 0.5)
    assert float(L.rows.to_dense().abs().sum()) == 0.0
    target, neg = L.loss_dense()
    assert float(target.sum()) == 0.0 and float(neg.sum()) == 0.0
